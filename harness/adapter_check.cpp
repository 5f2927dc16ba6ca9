// adapter_check.cpp -- drives the HIP backend exactly the way the reference's test_third_party<K,R,decoder_t>
// (src/main.cpp:239-282) drives a third-party decoder: decoder_t(poly, transmit_bits), then reset() / update(sym, n) /
// chainback(data, bits) on HOST buffers, one frame per handle.  Two bindings are exercised:
//   (1) include/hip_interface.h's hip_viterbiXX classes;
//   (2) the five C functions per code through a function-pointer adapter shaped like src/ka9q_interface.h:12-55,
//       which is what INTEGRATION.md §2a proposes for the reference tree.
// Exit code 0 = every code decoded a noise-free frame without bit errors through both bindings.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/hip_interface.h"

// (2) a stand-in for the reference's adapter template: same template parameter list, same three methods
template <size_t K_, size_t R_, typename vRK, vRK *(*vRK_create)(const int *, int), int (*vRK_init)(vRK *, int),
          void (*vRK_update)(vRK *, uint8_t *, int), int (*vRK_chainback)(vRK *, uint8_t *, uint32_t, uint32_t),
          void (*vRK_delete)(vRK *)>
class five_function_adapter {
public:
    five_function_adapter(const int *poly, size_t transmit_bits) : m(vRK_create(poly, int(transmit_bits))) {}
    ~five_function_adapter() { vRK_delete(m); }
    void reset() { vRK_init(m, 0); }
    void update(uint8_t *sym, size_t total_syms) { vRK_update(m, sym, int(total_syms / R_)); }
    void chainback(uint8_t *data, size_t total_bits) { vRK_chainback(m, data, uint32_t(total_bits), 0); }
    bool ok() const { return m != nullptr; }

private:
    vRK *m;
};
using ff_viterbi27 = five_function_adapter<7, 2, v27_hip, create_viterbi27_hip, init_viterbi27_hip, update_viterbi27_blk_hip, chainback_viterbi27_hip, delete_viterbi27_hip>;
using ff_viterbi615 = five_function_adapter<15, 6, v615_hip, create_viterbi615_hip, init_viterbi615_hip, update_viterbi615_blk_hip, chainback_viterbi615_hip, delete_viterbi615_hip>;
using ff_spiral47 = five_function_adapter<7, 4, spiral47_hip, create_spiral47_hip, init_spiral47_hip, update_spiral47_hip, chainback_spiral47_hip, delete_spiral47_hip>;

// the reference's call sequence (src/main.cpp:246-280), once
template <size_t K, size_t R, typename decoder_t>
static int run_third_party(const char *name, const int *poly, int payload_bytes, size_t extra_chainback_bits = 0) {
    const size_t decode_bits = size_t(payload_bytes) * 8, transmit_bits = decode_bits + K - 1, symbols = transmit_bits * R;
    std::vector<uint8_t> x_in(payload_bytes), y(symbols), x_out((decode_bits + extra_chainback_bits + 7) / 8);
    if (vhip_gen_frames_host(int(K), int(R), poly, 0x1234, 0, 1, payload_bytes, int(127.5 * 65536), 0, x_in.data(), y.data()) != 0) return 1;
    decoder_t decoder(poly, transmit_bits);
    for (int rep = 0; rep < 2; rep++) {  // the harness loops; make sure a handle can be reused
        std::memset(x_out.data(), 0, x_out.size());
        decoder.reset();
        decoder.update(y.data(), y.size());
        decoder.chainback(x_out.data(), decode_bits + extra_chainback_bits);
    }
    size_t errors = 0;
    for (int i = 0; i < payload_bytes; i++) errors += size_t(__builtin_popcount(unsigned(x_in[i] ^ x_out[i])));
    printf("%-28s K=%zu R=%zu  bit errors %zu / %zu\n", name, K, R, errors, decode_bits);
    return errors == 0 ? 0 : 1;
}

int main() {
    if (vhip_device_count() < 1) {
        fprintf(stderr, "no HIP device\n");
        return 2;
    }
    const int p27[] = {0x6d, 0x4f}, p47[] = {121, 117, 91, 111}, p29[] = {0x1af, 0x11d}, p49[] = {501, 441, 331, 315};
    const int p615[] = {042631, 047245, 056507, 073363, 077267, 064537}, p224[] = {062650457, 062650455};
    int bad = 0;
    bad += run_third_party<7, 2, hip_viterbi27>("hip_interface.h viterbi27", p27, 1024);
    bad += run_third_party<7, 4, hip_spiral47>("hip_interface.h spiral47", p47, 1024);
    bad += run_third_party<9, 2, hip_viterbi29>("hip_interface.h viterbi29", p29, 512);
    bad += run_third_party<9, 4, hip_spiral49>("hip_interface.h spiral49", p49, 512);
    bad += run_third_party<15, 6, hip_viterbi615>("hip_interface.h viterbi615", p615, 256);
    // K=24 decodes correctly only when chainback is asked for nbits+K-1
    bad += run_third_party<24, 2, hip_viterbi224>("hip_interface.h viterbi224", p224, 8, 23);
    bad += run_third_party<7, 2, ff_viterbi27>("five functions viterbi27", p27, 1024);
    bad += run_third_party<7, 4, ff_spiral47>("five functions spiral47", p47, 1024);
    bad += run_third_party<15, 6, ff_viterbi615>("five functions viterbi615", p615, 256);
    printf(bad ? "FAILED (%d)\n" : "all bindings ok\n", bad);
    return bad ? 1 : 0;
}
