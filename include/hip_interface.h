// hip_interface.h -- C++ adapter over the C ABI (include/viterbi_hip.h) with the duck type the reference harness
// expects from a third-party decoder: constructible from (poly, transmit_bits), then reset() / update(sym, n) /
// chainback(data, bits)  (what test_third_party<K,R,decoder_t> calls, src/main.cpp:239-282; the ka9q/spiral twins
// are src/ka9q_interface.h:12-55 and src/spiral_interface.h:13-56).
//
// Unlike those adapters this one is keyed on a code id rather than five function-pointer template arguments, owns
// its handle with unique_ptr semantics, and adds the batched/device entry points the GPU path needs.
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>

#include "viterbi_hip.h"

template <int CODE, std::size_t K_, std::size_t R_>
class hip_viterbi_decoder {
public:
    static constexpr std::size_t K = K_;
    static constexpr std::size_t R = R_;

    // one frame per handle: the reference's contract
    hip_viterbi_decoder(const int *poly, std::size_t transmit_bits) : hip_viterbi_decoder(poly, transmit_bits, 1) {}
    // nframes independent frames per handle, frame-major buffers
    hip_viterbi_decoder(const int *poly, std::size_t transmit_bits, int nframes)
        : m_handle(vhip_create(CODE, poly, static_cast<int>(transmit_bits), nframes)), m_nframes(nframes) {
        if (!m_handle) throw std::runtime_error(std::string("hip_viterbi_decoder: ") + vhip_last_error());
    }
    hip_viterbi_decoder(hip_viterbi_decoder &&o) noexcept : m_handle(std::exchange(o.m_handle, nullptr)), m_nframes(o.m_nframes) {}
    hip_viterbi_decoder &operator=(hip_viterbi_decoder &&o) noexcept {
        if (this != &o) {
            vhip_delete(m_handle);
            m_handle = std::exchange(o.m_handle, nullptr);
            m_nframes = o.m_nframes;
        }
        return *this;
    }
    hip_viterbi_decoder(const hip_viterbi_decoder &) = delete;
    hip_viterbi_decoder &operator=(const hip_viterbi_decoder &) = delete;
    ~hip_viterbi_decoder() { vhip_delete(m_handle); }

    // -- the three calls the harness times (src/main.cpp:264-278); host pointers, blocking
    void reset() { check(vhip_init(m_handle, 0), "init"); }
    void update(std::uint8_t *sym, std::size_t total_syms) {
        const std::size_t per_frame = total_syms / static_cast<std::size_t>(m_nframes);
        check(vhip_update(m_handle, sym, static_cast<int>(per_frame / R)), "update");
    }
    void chainback(std::uint8_t *data, std::size_t total_bits) {
        // ka9q615 returns a path metric here (viterbi615_sse2.cpp:90), any int: like the reference adapter we ignore the
        // value, but a failed GPU chainback must not hand back stale bytes silently -- the status is separate
        (void)vhip_chainback(m_handle, data, static_cast<unsigned>(total_bits), 0u);
        check(vhip_status(m_handle), "chainback");
    }

    // -- device-resident variants (asynchronous on the handle's stream)
    void set_stream(void *hip_stream) { check(vhip_set_stream(m_handle, hip_stream), "set_stream"); }
    void update_device(const std::uint8_t *d_sym, int steps) { check(vhip_update_dev(m_handle, d_sym, steps), "update_dev"); }
    void chainback_device(std::uint8_t *d_data, std::size_t total_bits) {
        check(vhip_chainback_dev(m_handle, d_data, static_cast<unsigned>(total_bits), 0u), "chainback_dev");
    }
    // double-buffered decodes: reset() starts a new decode on the next history buffer; join() orders the outputs on the stream
    void set_pipeline_depth(int depth) { check(vhip_set_pipeline_depth(m_handle, depth), "set_pipeline_depth"); }
    void reset_device() { check(vhip_init(m_handle, 0), "init"); }
    void join() { check(vhip_join(m_handle), "join"); }
    void sync() { check(vhip_sync(m_handle), "sync"); }
    int frames() const { return m_nframes; }
    vhip_decoder *handle() { return m_handle; }

private:
    static void check(int rc, const char *what) {
        if (rc < 0) throw std::runtime_error(std::string("hip_viterbi_decoder::") + what + ": " + vhip_last_error());
    }
    vhip_decoder *m_handle;
    int m_nframes;
};

using hip_viterbi27 = hip_viterbi_decoder<VHIP_KA9Q27, 7, 2>;
using hip_viterbi29 = hip_viterbi_decoder<VHIP_KA9Q29, 9, 2>;
using hip_viterbi615 = hip_viterbi_decoder<VHIP_KA9Q615, 15, 6>;
using hip_viterbi224 = hip_viterbi_decoder<VHIP_KA9Q224, 24, 2>;
using hip_spiral47 = hip_viterbi_decoder<VHIP_SPIRAL47, 7, 4>;
using hip_spiral49 = hip_viterbi_decoder<VHIP_SPIRAL49, 9, 4>;
// the remaining spiral arithmetic variants of src/main.cpp:370,388,408 (test_spiral<...> lines)
using hip_spiral27 = hip_viterbi_decoder<VHIP_SPIRAL27, 7, 2>;
using hip_spiral29 = hip_viterbi_decoder<VHIP_SPIRAL29, 9, 2>;
using hip_spiral615 = hip_viterbi_decoder<VHIP_SPIRAL615, 15, 6>;
