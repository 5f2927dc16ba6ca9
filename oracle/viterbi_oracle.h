/*
 * viterbi_oracle.h -- CPU ORACLE (test infrastructure, NOT a product path).
 *
 * Scalar plain-C restatement of the Viterbi decoders on the north-star hot path of
 * williamyang98/ka9q_viterbi_comparison.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library; the shipped decode path is the HIP
 * library in ka9q_viterbi_comparison_amd/csrc and never calls into oracle/.
 *
 * Parity status: PINNED.  The reference ships no tests or golden vectors, so the
 * restatement is pinned by running the reference's own decoder sources, compiled where they lie
 * under /root/reference by oracle/Makefile into oracle/_ref/, on identical inputs
 * (tests/test_oracle_vs_reference.py) and by the fixtures under tests/golden/ that
 * oracle/make_golden.py generated from that compiled reference.
 *
 * Every function cites the reference file:line it follows (paths relative to the reference root).
 */
#ifndef VITERBI_ORACLE_H
#define VITERBI_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Decoder variants.  The first six are the GPU parity targets; the last three are
 * the remaining spiral arithmetic variants (CPU baseline column / "next" row n3). */
enum vo_code {
    VO_KA9Q27 = 0,    /* ka9q_libfec_port/viterbi27_sse2.cpp  K=7  r=1/2 u8 modular      */
    VO_KA9Q29 = 1,    /* ka9q_libfec_port/viterbi29_sse2.cpp  K=9  r=1/2 u8 modular      */
    VO_KA9Q615 = 2,   /* ka9q_libfec_port/viterbi615_sse2.cpp K=15 r=1/6 i16 saturating  */
    VO_KA9Q224 = 3,   /* ka9q_libfec_port/viterbi224_sse2.cpp K=24 r=1/2 i16 saturating  */
    VO_SPIRAL47 = 4,  /* spiral/spiral47.cpp                  K=7  r=1/4 u8 saturating   */
    VO_SPIRAL49 = 5,  /* spiral/spiral49.cpp                  K=9  r=1/4 u8 saturating   */
    VO_SPIRAL27 = 6,  /* spiral/spiral27.cpp                  K=7  r=1/2 u8 saturating   */
    VO_SPIRAL29 = 7,  /* spiral/spiral29.cpp                  K=9  r=1/2 u8 saturating   */
    VO_SPIRAL615 = 8, /* spiral/spiral615.cpp                 K=15 r=1/6 u8 saturating   */
    VO_NUM_CODES = 9
};

typedef struct vo_decoder vo_decoder;

int vo_code_K(int code);
int vo_code_R(int code);

/* The five reference entry points (ka9q_libfec_port/viterbi27_sse2.h:3-8 and siblings). */
vo_decoder *vo_create(int code, const int *poly, int len);
int vo_init(vo_decoder *p, int starting_state);
void vo_update_blk(vo_decoder *p, const unsigned char *syms, int nbits);
int vo_chainback(vo_decoder *p, unsigned char *data, unsigned int nbits, unsigned int endstate);
void vo_delete(vo_decoder *p);

/* Sliding-window traceback over the rows written so far (row n4 of SURVEY.md §8f; semantics defined in the .c file --
 * there is no reference function for it: parity unpinned). */
int vo_chainback_windowed(vo_decoder *p, unsigned char *data, unsigned int nbits, unsigned int depth, unsigned int block);

/* Introspection for parity tests: decision bitmap rows (little-endian, bit n = new state n, row
 * stride = 2^(K-1)/8 bytes), number of rows written, and the current ("old") path metrics widened to
 * int32 in natural units (u8 value or i16 value). */
const unsigned char *vo_decision_rows(const vo_decoder *p);
size_t vo_row_bytes(const vo_decoder *p);
int vo_rows_written(const vo_decoder *p);
int vo_num_states(const vo_decoder *p);
void vo_get_metrics(const vo_decoder *p, int32_t *out);
int vo_renorm_count(const vo_decoder *p);

/* Convolutional encoder used to make test input (convention of SURVEY.md App. A.1; the reference's own
 * encoder lives in the un-vendored williamyang98/ViterbiDecoderCpp submodule, src/util.h:14-62). Emits
 * R*(8*nbytes+K-1) hard coded bits (0/1), step-major. Returns the number of coded bits written. */
size_t vo_encode(int K, int R, const int *poly, const unsigned char *payload, size_t nbytes,
                 unsigned char *coded_bits);

/* bench.py cpu_baseline leg only: timed reset+update+chainback loop, returns frames decoded. */
long vo_bench_loop(vo_decoder *p, const unsigned char *syms, int nsample, long frame_stride, int steps, unsigned nbits,
                   double seconds, double *elapsed);

/* whole-batch parity tests: nframes frames decoded one after the other (reset + update + chainback each), all bytes kept */
void vo_decode_batch(vo_decoder *p, const unsigned char *syms, long nframes, long frame_stride, int steps, unsigned nbits,
                     unsigned char *out, long out_stride);

#ifdef __cplusplus
}
#endif
#endif
