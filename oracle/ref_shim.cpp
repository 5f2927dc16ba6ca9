// ref_shim.cpp -- ORACLE side (test infrastructure, NOT a product path).
//
// extern "C" wrappers around the reference's own decoder entry points so that Python (ctypes) can drive
// the *genuine* reference objects compiled by oracle/Makefile from the sources where they lie under
// /root/reference (nothing of the reference is copied into this repository; outputs go to oracle/_ref/).
//
// The reference headers forward-declare opaque structs (ka9q_libfec_port/viterbi27_sse2.h:3-8 and
// siblings; spiral/spiral47.h).  To let the parity tests compare decision rows and path metrics -- not
// just decoded bytes -- this file re-declares *layout mirrors* of those structs (field order and sizes as
// documented in SURVEY.md §8a: viterbi27_sse2.cpp:9-39, viterbi29_sse2.cpp:6-29, viterbi615_sse2.cpp:13-26,
// viterbi224_sse2.cpp:14-29, spiral47.cpp:20-51) and reads through them.  A wrong mirror shows up as a
// row/metric mismatch in tests/test_oracle_vs_reference.py, never as a silent pass.
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <chrono>
#include <vector>

#include "viterbi224_sse2.h"
#include "viterbi27_sse2.h"
#include "viterbi29_sse2.h"
#include "viterbi615_sse2.h"
#ifndef REF_NO_SPIRAL
#include "spiral27.h"
#include "spiral29.h"
#include "spiral47.h"
#include "spiral49.h"
#include "spiral615.h"
#endif

namespace {

// Layout mirror of a ka9q handle: two metric buffers of MBYTES each (16-byte aligned unions holding
// __m128i), then dp, old_metrics, new_metrics, decisions.
template <size_t MBYTES>
struct ka9q_peek {
    alignas(16) unsigned char m1[MBYTES];
    alignas(16) unsigned char m2[MBYTES];
    void *dp;
    void *old_metrics;
    void *new_metrics;
    void *decisions;
};
// Layout mirror of a spiral handle: two byte arrays, then old, new, decisions.
template <size_t MBYTES>
struct spiral_peek {
    unsigned char m1[MBYTES];
    unsigned char m2[MBYTES];
    void *old_metrics;
    void *new_metrics;
    void *decisions;
};

enum {
    C_KA9Q27 = 0, C_KA9Q29, C_KA9Q615, C_KA9Q224, C_SPIRAL47, C_SPIRAL49, C_SPIRAL27, C_SPIRAL29, C_SPIRAL615
};

}  // namespace

extern "C" {

// sizeof(unsigned long) the reference objects were built with: the K=15 chainback indexes decision words
// through `unsigned long w[512]` (viterbi615_sse2.cpp:13,86), correct only when this is 4.
int ref_sizeof_long(void) {
#ifdef REF_W32
    return 4;
#else
    return (int)sizeof(unsigned long);
#endif
}

void *ref_create(int code, const int *poly, int len) {
    switch (code) {
    case C_KA9Q27: return create_viterbi27_sse2(poly, len);
    case C_KA9Q29: return create_viterbi29_sse2(poly, len);
    case C_KA9Q615: return create_viterbi615_sse2(poly, len);
    case C_KA9Q224: return create_viterbi224_sse2(poly, len);
#ifndef REF_NO_SPIRAL
    case C_SPIRAL47: return create_spiral47(poly, len);
    case C_SPIRAL49: return create_spiral49(poly, len);
    case C_SPIRAL27: return create_spiral27(poly, len);
    case C_SPIRAL29: return create_spiral29(poly, len);
    case C_SPIRAL615: return create_spiral615(poly, len);
#endif
    }
    return nullptr;
}

int ref_init(int code, void *p, int s) {
    switch (code) {
    case C_KA9Q27: return init_viterbi27_sse2((v27 *)p, s);
    case C_KA9Q29: return init_viterbi29_sse2((v29 *)p, s);
    case C_KA9Q615: return init_viterbi615_sse2((v615 *)p, s);
    case C_KA9Q224: return init_viterbi224_sse2((v224 *)p, s);
#ifndef REF_NO_SPIRAL
    case C_SPIRAL47: return init_spiral47((spiral47 *)p, s);
    case C_SPIRAL49: return init_spiral49((spiral49 *)p, s);
    case C_SPIRAL27: return init_spiral27((spiral27 *)p, s);
    case C_SPIRAL29: return init_spiral29((spiral29 *)p, s);
    case C_SPIRAL615: return init_spiral615((spiral615 *)p, s);
#endif
    }
    return -1;
}

void ref_update(int code, void *p, unsigned char *syms, int nbits) {
    switch (code) {
    case C_KA9Q27: update_viterbi27_blk_sse2((v27 *)p, syms, nbits); break;
    case C_KA9Q29: update_viterbi29_blk_sse2((v29 *)p, syms, nbits); break;
    case C_KA9Q615: update_viterbi615_blk_sse2((v615 *)p, syms, nbits); break;
    case C_KA9Q224: update_viterbi224_blk_sse2((v224 *)p, syms, nbits); break;
#ifndef REF_NO_SPIRAL
    case C_SPIRAL47: update_spiral47((spiral47 *)p, syms, nbits); break;
    case C_SPIRAL49: update_spiral49((spiral49 *)p, syms, nbits); break;
    case C_SPIRAL27: update_spiral27((spiral27 *)p, syms, nbits); break;
    case C_SPIRAL29: update_spiral29((spiral29 *)p, syms, nbits); break;
    case C_SPIRAL615: update_spiral615((spiral615 *)p, syms, nbits); break;
#endif
    }
}

int ref_chainback(int code, void *p, unsigned char *data, unsigned int nbits, unsigned int endstate) {
    switch (code) {
    case C_KA9Q27: return chainback_viterbi27_sse2((v27 *)p, data, nbits, endstate);
    case C_KA9Q29: return chainback_viterbi29_sse2((v29 *)p, data, nbits, endstate);
    case C_KA9Q615: return chainback_viterbi615_sse2((v615 *)p, data, nbits, endstate);
    case C_KA9Q224: return chainback_viterbi224_sse2((v224 *)p, data, nbits, endstate);
#ifndef REF_NO_SPIRAL
    case C_SPIRAL47: return chainback_spiral47((spiral47 *)p, data, nbits, endstate);
    case C_SPIRAL49: return chainback_spiral49((spiral49 *)p, data, nbits, endstate);
    case C_SPIRAL27: return chainback_spiral27((spiral27 *)p, data, nbits, endstate);
    case C_SPIRAL29: return chainback_spiral29((spiral29 *)p, data, nbits, endstate);
    case C_SPIRAL615: return chainback_spiral615((spiral615 *)p, data, nbits, endstate);
#endif
    }
    return -1;
}

void ref_delete(int code, void *p) {
    if (!p) return;
    switch (code) {
    case C_KA9Q27: delete_viterbi27_sse2((v27 *)p); break;
    case C_KA9Q29: delete_viterbi29_sse2((v29 *)p); break;
    case C_KA9Q615: delete_viterbi615_sse2((v615 *)p); break;
    case C_KA9Q224: delete_viterbi224_sse2((v224 *)p); break;
#ifndef REF_NO_SPIRAL
    case C_SPIRAL47: delete_spiral47((spiral47 *)p); break;
    case C_SPIRAL49: delete_spiral49((spiral49 *)p); break;
    case C_SPIRAL27: delete_spiral27((spiral27 *)p); break;
    case C_SPIRAL29: delete_spiral29((spiral29 *)p); break;
    case C_SPIRAL615: delete_spiral615((spiral615 *)p); break;
#endif
    }
}

// Decision rows: base pointer of row 0, stride in bytes between rows (LP64 strides: SURVEY.md App. A.1),
// used bytes per row = N/8.
const unsigned char *ref_rows(int code, void *p, size_t *stride) {
    switch (code) {
    case C_KA9Q27: *stride = 2 * sizeof(unsigned long); return (const unsigned char *)((ka9q_peek<64> *)p)->decisions;
    case C_KA9Q29: *stride = 8 * sizeof(unsigned long); return (const unsigned char *)((ka9q_peek<256> *)p)->decisions;
    case C_KA9Q615:
#ifdef REF_W32
        *stride = 512 * 4;
#else
        *stride = 512 * sizeof(unsigned long);
#endif
        return (const unsigned char *)((ka9q_peek<32768> *)p)->decisions;
    case C_KA9Q224: *stride = (size_t)1 << 20; return (const unsigned char *)((ka9q_peek<(1u << 24)> *)p)->decisions;
    case C_SPIRAL47: case C_SPIRAL27: *stride = 8; return (const unsigned char *)((spiral_peek<64> *)p)->decisions;
    case C_SPIRAL49: case C_SPIRAL29: *stride = 32; return (const unsigned char *)((spiral_peek<256> *)p)->decisions;
    case C_SPIRAL615: *stride = 2048; return (const unsigned char *)((spiral_peek<16384> *)p)->decisions;
    }
    *stride = 0;
    return nullptr;
}

// Current ("old") path metrics widened to int32 in natural units.
void ref_metrics(int code, void *p, int32_t *out) {
    switch (code) {
    case C_KA9Q27: { auto *m = (const uint8_t *)((ka9q_peek<64> *)p)->old_metrics; for (int i = 0; i < 64; i++) out[i] = m[i]; break; }
    case C_KA9Q29: { auto *m = (const uint8_t *)((ka9q_peek<256> *)p)->old_metrics; for (int i = 0; i < 256; i++) out[i] = m[i]; break; }
    case C_KA9Q615: { auto *m = (const int16_t *)((ka9q_peek<32768> *)p)->old_metrics; for (int i = 0; i < 16384; i++) out[i] = m[i]; break; }
    case C_KA9Q224: { auto *m = (const int16_t *)((ka9q_peek<(1u << 24)> *)p)->old_metrics; for (int i = 0; i < (1 << 23); i++) out[i] = m[i]; break; }
    case C_SPIRAL47: case C_SPIRAL27: { auto *m = (const uint8_t *)((spiral_peek<64> *)p)->old_metrics; for (int i = 0; i < 64; i++) out[i] = m[i]; break; }
    case C_SPIRAL49: case C_SPIRAL29: { auto *m = (const uint8_t *)((spiral_peek<256> *)p)->old_metrics; for (int i = 0; i < 256; i++) out[i] = m[i]; break; }
    case C_SPIRAL615: { auto *m = (const uint8_t *)((spiral_peek<16384> *)p)->old_metrics; for (int i = 0; i < 16384; i++) out[i] = m[i]; break; }
    }
}

// Timed decode loop for bench.py's cpu_baseline leg: one frame per call, reset + update + chainback per frame
// exactly as src/main.cpp:257-280 does, cycling over `nsample` frames until `seconds` have elapsed.  Runs entirely
// in C so that Python threads (GIL released by ctypes) measure the decoder, not the interpreter.
long ref_bench_loop(int code, void *p, const unsigned char *syms, int nsample, long frame_stride, int steps, unsigned nbits,
                    double seconds, double *elapsed) {
    std::vector<unsigned char> out((nbits + 7) / 8 + 8), tmp((size_t)frame_stride);
    const auto t0 = std::chrono::steady_clock::now();
    long n = 0;
    double el = 0;
    for (;;) {
        memcpy(tmp.data(), syms + (size_t)(n % nsample) * (size_t)frame_stride, (size_t)frame_stride);
        ref_init(code, p, 0);
        ref_update(code, p, tmp.data(), steps);
        ref_chainback(code, p, out.data(), nbits, 0);
        n++;
        if ((n & 7) == 0 || steps > 100000) {
            el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (el >= seconds) break;
        }
    }
    *elapsed = el;
    return n;
}

// Decodes `nframes` frames one after the other with ONE decoder object (reset + update + chainback per frame, the harness's call
// sequence) and keeps every frame's bytes: the whole-batch parity tests compare a full GPU batch with this.
void ref_decode_batch(int code, void *p, const unsigned char *syms, long nframes, long frame_stride, int steps, unsigned nbits,
                      unsigned char *out, long out_stride) {
    std::vector<unsigned char> tmp((size_t)frame_stride);
    for (long f = 0; f < nframes; f++) {
        memcpy(tmp.data(), syms + (size_t)f * (size_t)frame_stride, (size_t)frame_stride);  // the reference takes unsigned char*, not const
        ref_init(code, p, 0);
        ref_update(code, p, tmp.data(), steps);
        memset(out + (size_t)f * (size_t)out_stride, 0, (size_t)out_stride);
        ref_chainback(code, p, out + (size_t)f * (size_t)out_stride, nbits, 0);
    }
}

}  // extern "C"
