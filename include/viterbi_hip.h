/*
 * viterbi_hip.h -- C ABI of the MI355X (gfx950) Viterbi decode backend.
 *
 * Drop-in boundary for the decoder path of williamyang98/ka9q_viterbi_comparison: the reference harness
 * drives every third-party decoder through five free functions over an opaque handle
 *      create / init / update_*_blk / chainback / delete
 * (ka9q_libfec_port/viterbi27_sse2.h:3-8, viterbi29_sse2.h, viterbi615_sse2.h, viterbi224_sse2.h;
 *  spiral/spiral47.h:5-9, spiral49.h) wrapped by src/ka9q_interface.h:12-55 / src/spiral_interface.h:13-56.
 * This header declares the same five functions per code for the HIP backend (one frame per handle, host
 * pointers, blocking -- exactly the reference contract), plus an additive batched / device-pointer API
 * (`vhip_*`) that decodes thousands of independent frames per launch.
 *
 * Plain C: pointers and sizes only.  The library is ka9q_viterbi_comparison_amd/csrc/libviterbi_hip.so.
 * Every function fails loudly (NULL / negative return, message in vhip_last_error()) when no gfx950 device
 * or kernel image is available; there is no CPU fallback.
 */
#ifndef VITERBI_HIP_H
#define VITERBI_HIP_H

#include <stddef.h>
#include <stdint.h>

/* The library is built with -fvisibility=hidden: exactly the functions declared here are exported. */
#if defined(__GNUC__)
#define VHIP_API __attribute__((visibility("default")))
#else
#define VHIP_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* Code identifiers (numbering shared with oracle/viterbi_oracle.h so tests can iterate both). */
enum vhip_code {
    VHIP_KA9Q27 = 0,   /* K=7  r=1/2, ka9q u8 modular arithmetic  (viterbi27_sse2.cpp)   */
    VHIP_KA9Q29 = 1,   /* K=9  r=1/2, ka9q u8 modular arithmetic  (viterbi29_sse2.cpp)   */
    VHIP_KA9Q615 = 2,  /* K=15 r=1/6, ka9q i16 saturating         (viterbi615_sse2.cpp)  */
    VHIP_KA9Q224 = 3,  /* K=24 r=1/2, ka9q i16 saturating         (viterbi224_sse2.cpp)  */
    VHIP_SPIRAL47 = 4, /* K=7  r=1/4, spiral u8 saturating        (spiral47.cpp)         */
    VHIP_SPIRAL49 = 5, /* K=9  r=1/4, spiral u8 saturating        (spiral49.cpp)         */
    /* the remaining spiral arithmetic variants (decoder x code matrix of src/main.cpp:363-419) */
    VHIP_SPIRAL27 = 6, /* K=7  r=1/2, spiral u8 saturating        (spiral27.cpp)         */
    VHIP_SPIRAL29 = 7, /* K=9  r=1/2, spiral u8 saturating        (spiral29.cpp)         */
    VHIP_SPIRAL615 = 8,/* K=15 r=1/6, spiral u8 saturating        (spiral615.cpp)        */
    VHIP_NUM_CODES = 9
};

typedef struct vhip_decoder vhip_decoder;

/* ------------------------------------------------------------------------------------------------
 * Batched API (additive; SURVEY.md §8b "Batch extension").
 *
 * Semantics follow the reference per frame:
 *  - poly: R ints, K-bit tap masks on sr=(state<<1)|bit; every polynomial must have bit K-1 and bit 0
 *    set (butterfly symmetry, viterbi27_sse2.cpp:149-152); otherwise create returns NULL.
 *  - len:  trellis steps per frame including the K-1 tail (src/main.cpp:247).
 *  - update: `syms` holds nframes frames, frame-major, each nbits*R bytes step-major (src/util.h:31-48),
 *    offset-binary u8 (0 = strong 0, 255 = strong 1).  ka9q codes are incremental (rows are appended at the
 *    handle's decision pointer, viterbi27_sse2.cpp:121,174); spiral codes restart at row 0 on every call and
 *    drop an odd last step (spiral47.cpp:536-538).
 *  - chainback: writes ceil(nbits/8) bytes per frame, frame-major, MSB-first (viterbi27_sse2.cpp:98-103).
 *    Return value: 0, or for VHIP_KA9Q615 with nframes==1 the end-state path metric
 *    (viterbi615_sse2.cpp:76,90); negative on error.
 * ------------------------------------------------------------------------------------------------ */
VHIP_API vhip_decoder *vhip_create(int code, const int *poly, int len, int nframes);
VHIP_API int vhip_init(vhip_decoder *p, int starting_state);
/* host-pointer, blocking (copies H2D / D2H around the kernels) */
VHIP_API int vhip_update(vhip_decoder *p, const unsigned char *syms, int nbits);
VHIP_API int vhip_chainback(vhip_decoder *p, unsigned char *data, unsigned int nbits, unsigned int endstate);
/* device-pointer, asynchronous on the handle's stream (inputs already resident in HBM).
 * Exception -- VHIP_KA9Q224: vhip_update_dev BLOCKS the calling thread until the update has been enqueued to its end.  The
 * K=24 renormalisation (viterbi224_sse2.cpp:226-246) is run speculatively, several multi-step passes ahead of the device,
 * and the host has to follow the passes' progress word to commit or replay them; with nframes > 1 three frames at a time
 * are in flight on internal streams, all driven by the calling thread (no helper threads).  vhip_init and
 * vhip_chainback_dev stay asynchronous. */
VHIP_API int vhip_update_dev(vhip_decoder *p, const unsigned char *d_syms, int nbits);
VHIP_API int vhip_chainback_dev(vhip_decoder *p, unsigned char *d_data, unsigned int nbits, unsigned int endstate);
VHIP_API void vhip_delete(vhip_decoder *p);

/* Error status of the last init / update / chainback on the handle: 0 = succeeded, -1 = failed (message in
 * vhip_last_error()).  The reference ABI cannot carry it: update_*_blk returns void, and chainback_viterbi615 returns
 * the end state's path metric (viterbi615_sse2.cpp:76,90), which may be any int, -1 included. */
VHIP_API int vhip_status(const vhip_decoder *p);

/* Pipelined decodes (additive).  depth = 2 (or 3) gives the handle that many {decision history, path metrics} sets and
 * internal streams; every vhip_init() starts a new decode on the next set, so init / update / chainback issued back to
 * back for decode i+1 overlap the (HBM-bound) chainback of decode i with the (VALU-bound) update of decode i+1.
 * Ordering: each device-pointer call is ordered BEHIND what the caller has enqueued on the handle's stream so far (its
 * symbols are ready); the decoded bytes are ordered on the handle's stream only by vhip_join() (device-side wait, the host
 * does not block) or vhip_sync().  Consecutive decodes must be given different output buffers.  depth can be set once,
 * before the first update; K=24 handles do not take it (they keep several frames in flight by themselves).
 * HBM cost: depth x the decision history (vhip_device_bytes). */
VHIP_API int vhip_set_pipeline_depth(vhip_decoder *p, int depth);
VHIP_API int vhip_get_pipeline_depth(const vhip_decoder *p);
VHIP_API int vhip_join(vhip_decoder *p);

/* Chainback of the K >= 15 codes with at most 64 frames per handle: one frame's traceback is a chain of dependent DRAM round
 * trips, so it is cut into segments of `seg_bits` decoded bits (a multiple of 8) that separate waves walk at the same time,
 * each from a guessed state `warmup_rows` rows above its segment; the guesses are then verified against the exact walk from
 * the caller's end state downwards and every segment whose guess had not merged is walked again, so the bytes are always
 * those of chainback_viterbi615_sse2 / chainback_viterbi224_sse2 (viterbi615_sse2.cpp:65-91, viterbi224_sse2.cpp:79-121).
 * seg_bits = 0: one walk per frame; negative values: the defaults (64 / 160).
 * vhip_chainback_rewalked: segments the last chainback had to walk twice (summed over frames; blocks until it is done);
 * *nseg = segments per frame of that chainback, 0 if it ran as one walk. */
VHIP_API int vhip_set_chainback_segments(vhip_decoder *p, int seg_bits, int warmup_rows);
VHIP_API int vhip_chainback_rewalked(vhip_decoder *p, int *nseg);

/* Live kernel timing.  When enabled, every vhip_update_dev / vhip_chainback_dev is bracketed by HIP events on the stream
 * its kernels run on (the handle's stream, or the internal stream of the current pipeline slot).  vhip_read_timing waits
 * for the handle to go idle and returns the summed durations (ms) and launch counts since the previous read. */
VHIP_API int vhip_enable_timing(vhip_decoder *p, int on);
VHIP_API int vhip_read_timing(vhip_decoder *p, double *update_ms_sum, int *n_update, double *chainback_ms_sum, int *n_chainback);

/* Fused sliding-window decode (additive): init + ACS update + traceback in ONE kernel for the K <= 9
 * and K = 15 codes with the harness polynomials.  The decisions of the last trellis steps live in a ring -- 64 rows in LDS
 * for K <= 9, 160 rows of a cache-resident global buffer per resident workgroup for K = 15 (320 MiB at most, whatever the
 * batch) --; as soon as the row vhip_window_depth() steps beyond a block of vhip_window_block() payload bits exists, the
 * block is traced back out of the ring (start state 0) and its bytes are stored -- no decision history and no path metrics
 * ever reach HBM, so the handle's history buffer is not used and the batch size is not bounded by it.  d_syms: nframes frames of (nbits + K - 1) * R
 * symbols from a freshly initialised decoder (start state 0); d_data: ceil(nbits/8) bytes per frame, MSB-first.
 * Results are those of a sliding-window Viterbi decoder -- NOT bit-identical to init/update/chainback unless the depth
 * covers the frame (the error rate converges on the exact path's as the depth grows: tests/test_windowed.py).  There is no
 * reference function with these semantics (the reference's decoders that take a traceback length live in its un-vendored
 * submodule, src/main.cpp:169); they are defined by oracle/viterbi_oracle.c vo_chainback_windowed. */
VHIP_API int vhip_decode_windowed_dev(vhip_decoder *p, const unsigned char *d_syms, unsigned int nbits, unsigned char *d_data);
VHIP_API int vhip_window_depth(const vhip_decoder *p); /* traceback depth in trellis steps, or -1 if the code has no fused decode */
VHIP_API int vhip_window_block(const vhip_decoder *p); /* payload bits decoded per traceback */

/* Stream / device plumbing.  `stream` is a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream);
 * NULL selects the device's default stream. */
VHIP_API int vhip_set_stream(vhip_decoder *p, void *stream);
VHIP_API int vhip_sync(vhip_decoder *p);
VHIP_API int vhip_device_count(void);
VHIP_API const char *vhip_last_error(void);

/* Kernel variant selection (0 = automatic).  Exposed so parity tests can pin every kernel family. */
enum vhip_variant {
    VHIP_VARIANT_AUTO = 0,
    VHIP_VARIANT_LDS = 1,   /* one workgroup per frame, metrics ping-pong in LDS, natural decision rows */
    VHIP_VARIANT_REGS = 2,  /* metrics packed in VGPRs: frames across lanes (K<=9), workgroup per frame (K=15) */
    VHIP_VARIANT_HBM = 3,   /* K=24: metrics tiled through HBM, one launch per trellis step */
    VHIP_VARIANT_HBM_FUSED = 4, /* K=24: 4 or 7 trellis steps per pass over the metric array (harness polynomials) */
    VHIP_VARIANT_HBM_TILED = 5, /* K=24: 9 or 14 steps per pass -- two passes per 23 steps, tiles regrouped through LDS */
    VHIP_VARIANT_WAVE = 6   /* K=7: one wavefront per frame, one state per lane -- the latency geometry for handles with few
                               frames (the reference's one-frame-per-handle methodology, src/main.cpp:257-280); any polynomials */
};
VHIP_API int vhip_set_variant(vhip_decoder *p, int variant);
/* Polynomials other than the reference harness's (src/main.cpp:367-415): the fast kernels are compiled for them when the
 * handle is created (hipcc --genco on the library's own kernel sources, a few seconds, cached under $VHIP_JIT_CACHE or
 * /tmp/vhip_jit_cache_<uid>) and run at the speed of the harness polynomials; 1 = this handle runs such a build.  Without
 * the sources or hipcc next to the library, or with VHIP_JIT=0, the handle uses the slower any-polynomial kernels
 * (VHIP_VARIANT_LDS / VHIP_VARIANT_HBM) and asking for the fast variant explicitly is an error. */
VHIP_API int vhip_is_runtime_specialised(const vhip_decoder *p);
/* Needs no device: 1 if the kernel sources next to the library are the ones it was built from (a fingerprint of their contents
 * is baked in at build time), 0 if they differ -- run-time builds are then refused, because a kernel with another decision
 * layout than the library's chainback kernels expect would decode silently wrong --, -1 if they cannot be read. */
VHIP_API int vhip_runtime_build_sources_ok(void);
VHIP_API int vhip_get_variant(const vhip_decoder *p);

/* Introspection for parity tests (blocking): natural decision bitmap rows (bit n of a row = new state n,
 * 2^(K-1)/8 bytes per row) and current path metrics in natural units widened to int32. */
VHIP_API int vhip_read_decision_rows(vhip_decoder *p, int frame, int row0, int nrows, unsigned char *out);
VHIP_API int vhip_read_metrics(vhip_decoder *p, int frame, int32_t *out);
VHIP_API int vhip_rows_written(const vhip_decoder *p);
VHIP_API int vhip_code_K(int code);
VHIP_API int vhip_code_R(int code);
/* Bytes of HBM the handle holds (decision history + metrics + staging). */
VHIP_API size_t vhip_device_bytes(const vhip_decoder *p);

/* ------------------------------------------------------------------------------------------------
 * Synthetic frames (analogue of src/util.h:8-62: random payload -> convolutional encoder -> soft symbols).
 * Integer-only and counter-based so that host and device generators are bit-identical:
 *   payload byte i of frame f  = splitmix64 stream keyed by (seed, f)
 *   symbol = clamp(round(127.5 + amp*(2c-1) + noise), 0, 255), noise = Irwin-Hall(8) scaled by noise_q12
 * amp_q16 = amplitude * 65536 (127.5*65536 with noise_q12 = 0 gives the reference's hard 0/255 symbols).
 * ------------------------------------------------------------------------------------------------ */
VHIP_API int vhip_noise_q12_from_ebn0(int R, double amp, double ebn0_db);
VHIP_API int vhip_gen_frames_host(int K, int R, const int *poly, uint64_t seed, uint64_t frame0, int nframes,
                         int payload_bytes, int amp_q16, int noise_q12, unsigned char *payload,
                         unsigned char *syms);
VHIP_API int vhip_gen_frames_dev(int K, int R, const int *poly, uint64_t seed, uint64_t frame0, int nframes,
                        int payload_bytes, int amp_q16, int noise_q12, unsigned char *d_payload,
                        unsigned char *d_syms, void *stream);
/* Bit errors between two device byte buffers (BER reduction, src/util.h:64-73), blocking. */
VHIP_API long long vhip_count_bit_errors_dev(const unsigned char *d_a, const unsigned char *d_b, size_t nbytes,
                                    void *stream);

/* ------------------------------------------------------------------------------------------------
 * Reference-shaped entry points: one frame per handle, host pointers, blocking.
 * Same signatures as the ka9q / spiral headers cited above with `_sse2` -> `_hip`.
 * ------------------------------------------------------------------------------------------------ */
#define VHIP_DECLARE_FIVE(T, CREATE, INIT, UPDATE, CHAINBACK, DELETE)                                   \
    struct T;                                                                                           \
    VHIP_API struct T *CREATE(const int *poly, int len);                                                \
    VHIP_API int INIT(struct T *p, int starting_state);                                                 \
    VHIP_API void UPDATE(struct T *p, unsigned char *syms, int nbits);                                  \
    VHIP_API int CHAINBACK(struct T *p, unsigned char *data, unsigned int nbits, unsigned int endstate);\
    VHIP_API void DELETE(struct T *p);

/* replaces ka9q_libfec_port/viterbi27_sse2.h:3-8 */
VHIP_DECLARE_FIVE(v27_hip, create_viterbi27_hip, init_viterbi27_hip, update_viterbi27_blk_hip,
                  chainback_viterbi27_hip, delete_viterbi27_hip)
/* replaces ka9q_libfec_port/viterbi29_sse2.h:3-8 */
VHIP_DECLARE_FIVE(v29_hip, create_viterbi29_hip, init_viterbi29_hip, update_viterbi29_blk_hip,
                  chainback_viterbi29_hip, delete_viterbi29_hip)
/* replaces ka9q_libfec_port/viterbi615_sse2.h:3-8 */
VHIP_DECLARE_FIVE(v615_hip, create_viterbi615_hip, init_viterbi615_hip, update_viterbi615_blk_hip,
                  chainback_viterbi615_hip, delete_viterbi615_hip)
/* replaces ka9q_libfec_port/viterbi224_sse2.h:3-8 */
VHIP_DECLARE_FIVE(v224_hip, create_viterbi224_hip, init_viterbi224_hip, update_viterbi224_blk_hip,
                  chainback_viterbi224_hip, delete_viterbi224_hip)
/* replaces spiral/spiral47.h:5-9 */
VHIP_DECLARE_FIVE(spiral47_hip, create_spiral47_hip, init_spiral47_hip, update_spiral47_hip,
                  chainback_spiral47_hip, delete_spiral47_hip)
/* replaces spiral/spiral49.h:5-9 */
VHIP_DECLARE_FIVE(spiral49_hip, create_spiral49_hip, init_spiral49_hip, update_spiral49_hip,
                  chainback_spiral49_hip, delete_spiral49_hip)

/* replaces spiral/spiral27.h:5-9, spiral29.h, spiral615.h */
VHIP_DECLARE_FIVE(spiral27_hip, create_spiral27_hip, init_spiral27_hip, update_spiral27_hip,
                  chainback_spiral27_hip, delete_spiral27_hip)
VHIP_DECLARE_FIVE(spiral29_hip, create_spiral29_hip, init_spiral29_hip, update_spiral29_hip,
                  chainback_spiral29_hip, delete_spiral29_hip)
VHIP_DECLARE_FIVE(spiral615_hip, create_spiral615_hip, init_spiral615_hip, update_spiral615_hip,
                  chainback_spiral615_hip, delete_spiral615_hip)

#ifdef __cplusplus
}
#endif
#endif
