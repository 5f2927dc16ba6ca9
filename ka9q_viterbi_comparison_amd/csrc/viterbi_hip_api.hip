// viterbi_hip_api.hip -- C-ABI layer of the MI355X Viterbi backend (include/viterbi_hip.h).
//
// Host-side mirror of the reference decoder interface: create / init / update_blk / chainback / delete over
// an opaque handle (ka9q_libfec_port/viterbi27_sse2.h:3-8; adapter src/ka9q_interface.h:28-55), generalised
// to `nframes` independent frames per handle.  The handle owns the device buffers (decision history, path
// metrics, staging) and a stream; kernels live in acs_lds.hip / acs_regs.hip / acs_k24.hip / chainback.hip.
// There is no CPU decode path here: without a gfx950 device every entry point fails loudly.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/viterbi_hip.h"
#include "framegen.h"
#include "k15_layout.h"
#include "k24f_layout.h"
#include "k24t_layout.h"
#include "kernels.h"
#include "viterbi_codes.h"

namespace vh {
void gen_frames_host(int K, int R, const int *poly, uint64_t seed, uint64_t frame0, int nframes, int payload_bytes,
                     int amp_q16, int noise_q12, unsigned char *payload, unsigned char *syms);
}

namespace {

thread_local std::string g_last_error;
thread_local unsigned g_fail_count = 0;

int fail(const char *what, hipError_t e = hipSuccess) {
    char buf[512];
    // a failed runtime call also stays behind as the thread's "last error", and the next kernel launch -- of any handle -- would
    // read it back from hipGetLastError() as its own: one refused hipMalloc must not fail the healthy handle created after it
    (void)hipGetLastError();
    if (e != hipSuccess)
        snprintf(buf, sizeof(buf), "viterbi_hip: %s: %s", what, hipGetErrorString(e));
    else
        snprintf(buf, sizeof(buf), "viterbi_hip: %s", what);
    g_last_error = buf;
    g_fail_count++;
    if (getenv("VHIP_VERBOSE")) fprintf(stderr, "%s\n", buf);
    return -1;
}

#define HIP_TRY(expr)                                  \
    do {                                               \
        hipError_t _e = (expr);                        \
        if (_e != hipSuccess) return fail(#expr, _e);  \
    } while (0)

}  // namespace

struct vhip_decoder {
    int code = 0, K = 0, R = 0;
    unsigned N = 0;
    int poly[8] = {0};
    int len = 0, cap_rows = 0, nframes = 0;
    int incremental = 1;
    int variant = VHIP_VARIANT_LDS;
    int device = 0;
    hipStream_t stream = nullptr;
    size_t row_bytes = 0;
    unsigned char *d_dec = nullptr;  // natural rows [nframes][cap_rows][N/8]
    int16_t *d_metrics = nullptr;    // [nframes][N]; K=24: [nframes][2][N] ping-pong
    int *d_flags = nullptr;
    unsigned char *d_syms_stage = nullptr;
    size_t syms_stage_bytes = 0;
    unsigned char *d_data_stage = nullptr;
    size_t data_stage_bytes = 0;
    unsigned *d_cbseg = nullptr;      // segment-parallel chainback (K >= 15, few frames): ticket / register words, one region per slot
    size_t cbseg_words = 0;           //   words per region
    int cbseg_bits = -1, cbseg_ovl = -1;  // vhip_set_chainback_segments (-1: per-code default)
    int cbseg_last_nseg = 0;              // geometry of the last segment-parallel chainback (vhip_chainback_rewalked)
    unsigned *cbseg_last = nullptr;
    unsigned char *d_ring = nullptr;  // K=15 fused windowed decode: decision rings of the resident workgroups
    size_t ring_bytes = 0;
    int pos = 0;
    std::vector<int> k24_cur;  // per frame: which half of the ping-pong holds the current metrics
    int regs_lb = 0;           // REGS variant: log2(lanes per frame)
    vh::RegsLayout lay{};      // REGS variant: decision layout
    int frames_padded = 0;     // frames rounded up to a whole wave (both decision layouts fit the same buffer)
#ifndef VH_K24_WORKERS
#define VH_K24_WORKERS 3  // tools/k24_inflight.sh builds experiment libraries with other values (at most 7: four flag words per stream)
#endif
    static constexpr int K24_WORKERS = VH_K24_WORKERS;  // K=24 with several frames: this many decodes in flight
    hipStream_t aux_stream[K24_WORKERS] = {};
    unsigned long long *h_report = nullptr;          // K=24: pinned progress words (kernels.h K24Report), one per stream
    unsigned long long *h_report_dev = nullptr;      //        the same words as the device addresses them
    unsigned k24_seq[1 + K24_WORKERS] = {};          //        last sequence number handed out per stream
    unsigned k24_events[1 + K24_WORKERS] = {};       //        renormalisation events since the flag words of that stream were reset
    size_t total_bytes = 0;
    // Pipelined decodes (vhip_set_pipeline_depth): `depth` sets of {decision history, metrics, internal stream}.
    // d_dec / d_metrics / pos above always describe the CURRENT set; vhip_init() rotates to the next one.
    struct Slot {
        unsigned char *d_dec = nullptr;
        int16_t *d_metrics = nullptr;
        int pos = 0;
        hipStream_t stream = nullptr;  // internal, non-blocking
        hipEvent_t done = nullptr;     // last work enqueued on `stream` (recorded by vhip_join)
    };
    static constexpr int MAX_DEPTH = 3;
    Slot slots[MAX_DEPTH];
    int depth = 1, cur_slot = 0;
    hipEvent_t ev_input = nullptr;     // "the caller's stream has reached this call" (inputs are ready)
    int status = 0;                    // 0, or -1 after a failed call on this handle (vhip_status)
    // fast kernels compiled at run time for this handle's polynomials (jit.hip): regs / k15: [0]; k24t: H full, H part, L full, L part
    bool jit = false;
    hipFunction_t jit_fn[4] = {};
    // live kernel timing (vhip_enable_timing): event pairs around the update / chainback launches, on the stream they run on
    bool timing = false;
    std::vector<hipEvent_t> ev_pool;                        // timing-enabled events, reused
    std::vector<std::pair<hipEvent_t, hipEvent_t>> t_upd, t_cb;  // recorded, not yet read
    hipStream_t run_stream() const { return depth > 1 ? slots[cur_slot].stream : stream; }
};

namespace {

// K=24 tiled passes: non-temporal metric stores help a lone decode (the next pass does not wait for an end-of-kernel write-back
// of 16 MiB of dirty lines: 3.95 -> 3.39 ms per 2071-step frame) and cost 7 % when three decodes share the chip (12 frames on
// one handle: 1.92 -> 2.07 ms per frame), where the write-back is hidden behind the other decodes' passes.
constexpr int K24_WORKERS_MIN_PLAIN = 3;
// K=7 handles with at most this many frames take the one-wave-per-frame kernel (tools/small_batch_probe.py)
// (tools/small_batch_probe.py, 2054-step frames, update + chainback in ms.  ka9q27: 2048 frames 0.31 + 0.08 against 0.37 + 0.19 for
// the register kernel, 4096 frames 0.60 + 0.12 against 0.37 + 0.19.  spiral47, whose every step also forms the wave minimum
// (spiral47.cpp:313-331): 1024 frames 0.43 + 0.06 against 0.54 + 0.17, 2048 frames 0.85 + 0.08 against 0.55 + 0.19)
constexpr int WAVE_MAX_FRAMES_MOD = 2048, WAVE_MAX_FRAMES_SAT = 1024;
// K=9 (same probe): ka9q29, four waves per frame: 4096 frames 0.63 + 0.24 against 1.11 + 0.32 for the register kernel, 8192 frames
// 1.17 + 0.52 against 1.12 + 0.32; spiral49, one wave with four states per lane: 2048 frames 1.36 + 0.12 against 1.36 + 0.31, 4096 frames
// 2.71 + 0.25 against 1.37 + 0.32.  The one-workgroup-per-frame kernel (acs_lds) is slower than both everywhere.
constexpr int WAVE9_MAX_FRAMES_MOD = 4096, WAVE9_MAX_FRAMES_SAT = 2048;

// vhip_status(): -1 after a failed init / update / chainback on the handle, 0 after a successful one.  The reference ABI
// returns void from update and a path metric (any int) from chainback_viterbi615, so the return value alone cannot
// carry an error there.
struct StatusScope {
    vhip_decoder *p;
    unsigned before;
    explicit StatusScope(vhip_decoder *h) : p(h), before(g_fail_count) {}
    ~StatusScope() {
        if (p) p->status = g_fail_count != before ? -1 : 0;
    }
};

int init_all_of(int code) {
    switch (code) {
    case VHIP_KA9Q27: return vh::Code27::init_all;
    case VHIP_KA9Q29: return vh::Code29::init_all;
    case VHIP_KA9Q615: return vh::Code615::init_all;
    case VHIP_KA9Q224: return vh::Code224::init_all;
    case VHIP_SPIRAL47: return vh::Code47::init_all;
    case VHIP_SPIRAL49: return vh::Code49::init_all;
    case VHIP_SPIRAL27: case VHIP_SPIRAL29: case VHIP_SPIRAL615: return 63;
    }
    return 0;
}
int init_start_of(int code) {
    switch (code) {
    case VHIP_KA9Q615: return vh::Code615::init_start;
    case VHIP_KA9Q224: return vh::Code224::init_start;
    }
    return 0;
}

// a handle is bound to the device that was current at create(); callers may since have switched devices
int use_device(const vhip_decoder *p) {
    // every entry point comes through here first: drop whatever "last error" another library's failed call left in this thread,
    // or the first kernel launch below would report it as its own (launch_* return hipGetLastError())
    (void)hipGetLastError();
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur == p->device) return 0;
    HIP_TRY(hipSetDevice(p->device));
    return 0;
}

int ensure_stage(unsigned char **buf, size_t *cap, size_t need, size_t *total) {
    if (*cap >= need) return 0;
    if (*buf) {
        (void)hipFree(*buf);
        *total -= *cap;
        *buf = nullptr;
        *cap = 0;
    }
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(buf), need));
    *cap = need;
    *total += need;
    return 0;
}

// Pipelined handles: work of the current decode runs on the slot's internal stream, ordered behind whatever the caller
// has enqueued on the handle's stream so far (its symbols / output buffers are ready).
int order_behind_caller(vhip_decoder *p) {
    if (p->depth <= 1) return 0;
    HIP_TRY(hipEventRecord(p->ev_input, p->stream));
    HIP_TRY(hipStreamWaitEvent(p->run_stream(), p->ev_input, 0));
    return 0;
}
// timing bracket: returns the event to record after the launches (nullptr when timing is off)
hipEvent_t timing_begin(vhip_decoder *p, std::vector<std::pair<hipEvent_t, hipEvent_t>> &list, hipStream_t st) {
    if (!p->timing) return nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    for (int i = 0; i < 2; i++) {
        if (!p->ev_pool.empty()) {
            ev[i] = p->ev_pool.back();
            p->ev_pool.pop_back();
        } else if (hipEventCreate(&ev[i]) != hipSuccess) {
            return nullptr;
        }
    }
    if (hipEventRecord(ev[0], st) != hipSuccess) return nullptr;
    list.emplace_back(ev[0], ev[1]);
    return ev[1];
}
void timing_end(hipEvent_t e1, hipStream_t st) {
    if (e1) (void)hipEventRecord(e1, st);
}

// decision history of the current buffer set: allocated and zeroed on first use (rows never written read as zero)
int ensure_history(vhip_decoder *p) {
    if (p->d_dec) return 0;
    const size_t dec_bytes = (size_t)p->frames_padded * (size_t)p->cap_rows * p->row_bytes;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&p->d_dec), dec_bytes ? dec_bytes : 16);
    if (e != hipSuccess) {
        p->d_dec = nullptr;
        return fail("decision history allocation (frames x steps x 2^(K-1)/8 bytes; the fused windowed decode needs none)", e);
    }
    e = hipMemsetAsync(p->d_dec, 0, dec_bytes ? dec_bytes : 16, p->run_stream());
    if (e != hipSuccess) return fail("decision history memset", e);
    p->total_bytes += dec_bytes;
    if (p->depth > 1) p->slots[p->cur_slot].d_dec = p->d_dec;
    return 0;
}

// Segment-parallel speculative chainback (chainback_spec.hip) for K >= 15 when the handle holds few frames: the walk of one
// frame is a chain of DRAM round trips, so it is cut into segments walked by separate waves and verified afterwards.
// vhip_set_chainback_segments overrides the geometry (tests force a warm-up of 0 rows so that every segment is walked twice).
int setup_segments(vhip_decoder *p, vh::ChainbackRowsArgs &a) {
    // defaults measured with tools/chainback_segments_probe.py (one 2048-bit frame, K=24 at 4 dB / K=15 at 1 dB): 64-bit segments
    // with 128 warm-up rows never needed a second walk and were fastest; 160 rows leave a margin, and a segment whose guess has
    // not merged costs one extra 64-row walk (about 10 round trips), not correctness
    const int seg_bits = p->cbseg_bits >= 0 ? (p->cbseg_bits & ~7) : 64;
    const int seg_ovl = p->cbseg_ovl >= 0 ? p->cbseg_ovl : 160;
    p->cbseg_last_nseg = 0;
    if (seg_bits <= 0 || p->nframes > 64) return 0;  // many frames: one wave per frame already fills the memory system
    const int nseg = (int)((a.nbits + (unsigned)seg_bits - 1) / (unsigned)seg_bits);
    if (nseg <= 1) return 0;
    const size_t words = (size_t)p->nframes * (size_t)(2 + 2 * nseg);
    if (p->cbseg_words < words) {
        HIP_TRY(hipStreamSynchronize(p->stream));
        for (int i = 0; i < p->depth && p->depth > 1; i++) HIP_TRY(hipStreamSynchronize(p->slots[i].stream));
        if (p->d_cbseg) (void)hipFree(p->d_cbseg);
        p->d_cbseg = nullptr;
        p->cbseg_words = 0;
        const size_t cap = words * 2;
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p->d_cbseg), cap * sizeof(unsigned) * vhip_decoder::MAX_DEPTH));
        // hipMemset may return before the fill has run, and the internal streams of a pipelined handle are non-blocking (they
        // do not order themselves behind the null stream): wait for it, or the first chainback could find a ticket word that
        // is cleared under its feet
        HIP_TRY(hipMemset(p->d_cbseg, 0, cap * sizeof(unsigned) * vhip_decoder::MAX_DEPTH));
        HIP_TRY(hipStreamSynchronize(nullptr));
        p->cbseg_words = cap;
    }
    a.nseg = nseg;
    a.seg_bits = seg_bits;
    a.seg_ovl = seg_ovl;
    a.seg_scratch = p->d_cbseg + (size_t)(p->depth > 1 ? p->cur_slot : 0) * p->cbseg_words;
    p->cbseg_last_nseg = nseg;
    p->cbseg_last = a.seg_scratch;
    return 0;
}

int sync_all(vhip_decoder *p) {
    for (int i = 0; i < p->depth && p->depth > 1; i++) HIP_TRY(hipStreamSynchronize(p->slots[i].stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    return 0;
}

// lanes per frame for the REGS kernels.  K=7: the smallest L that still puts a wave on (most of) the 1024 SIMDs -- fewer
// lanes per frame means fewer duplicated branch-metric instructions and fewer half/lane stages; measured on 65536 K=7
// frames: L=1 0.92 ms, L=2 1.10 ms, L=4 1.31 ms.  K=9 (128 packed registers per frame): always four lanes, i.e. the same
// 32 metric registers per lane as K=7 with one -- 64 or 128 registers per lane leave hipcc no room to interleave butterflies;
// decode rate on a double-buffered handle, L = 1 / 2 / 4 (tools/lb_sweep.sh, profiles/r02_lanes_per_frame.txt): K=9 r=1/2
// 32768 frames 60.7 / 57.8 / 66.8 Gsym/s, 131072 frames 62.0 / 58.3 / 67.2; r=1/4 101.5 / 96.4 / 103.3 and 103.7 / 97.2 / 102.9.
int auto_regs_lb(int code, int nframes) {
    int lb = 0;
    while (lb < 2 && (long)nframes * (1 << lb) < 64L * 768) lb++;
    if (vhip_code_K(code) == 9) lb = 2;
    while (!vh::regs_lanes_supported(code, lb) && lb < 2) lb++;
    return lb;  // vhip_set_variant(VHIP_VARIANT_REGS | ((lb + 1) << 8)) forces another value per handle
}

int auto_variant(const vhip_decoder *p) {
    if (p->code == VHIP_KA9Q224) {
        if (vh::k24t_poly_supported(p->poly)) return VHIP_VARIANT_HBM_TILED;
        return vh::k24f_poly_supported(p->poly) ? VHIP_VARIANT_HBM_FUSED : VHIP_VARIANT_HBM;
    }
    // K=7 with few frames: one wave per frame, one state per lane -- a frame is a chain of dependent steps and this geometry
    // issues the fewest instructions per step (acs_wave.hip; 8198-step frame: 0.2 ms against 1.5 ms for the register kernel,
    // which catches up once its 16-64 frames per wave are all in use)
    if (p->K == 7 && vh::wave_code_supported(p->code) && p->nframes <= (p->code == VHIP_KA9Q27 ? WAVE_MAX_FRAMES_MOD : WAVE_MAX_FRAMES_SAT))
        return VHIP_VARIANT_WAVE;
    // K=9: four waves per frame (acs_wave9_kernel)
    if (p->K == 9 && vh::wave_code_supported(p->code) && p->nframes <= (p->code == VHIP_KA9Q29 ? WAVE9_MAX_FRAMES_MOD : WAVE9_MAX_FRAMES_SAT))
        return VHIP_VARIANT_WAVE;
    if (p->K <= 9 && vh::regs_poly_supported(p->code, p->poly)) {
        return VHIP_VARIANT_REGS;
    }
    if ((p->code == VHIP_KA9Q615 || p->code == VHIP_SPIRAL615) && vh::k15_poly_supported(p->poly)) return VHIP_VARIANT_REGS;
    return VHIP_VARIANT_LDS;
}

bool fast_poly_supported(const vhip_decoder *p) {
    if (p->code == VHIP_KA9Q224) return vh::k24t_poly_supported(p->poly);
    if (p->K == 15) return vh::k15_poly_supported(p->poly);
    return vh::regs_poly_supported(p->code, p->poly);
}

// Compile (or fetch from the cache) the fast kernel of this handle's code for ITS polynomials.  lb: lanes-per-frame log2 (K <= 9).
bool setup_jit(vhip_decoder *p, int lb, std::string *err) {
    if (!vh::jit_enabled()) {
        *err = "runtime specialisation is switched off (VHIP_JIT=0)";
        return false;
    }
    if (use_device(p) != 0) {  // the module is loaded on the current device: make that the handle's
        *err = "cannot select the handle's device";
        return false;
    }
    const std::string poly = vh::jit_poly_define(p->poly, p->R);
    if (p->code == VHIP_KA9Q224) {
        // metric stores: non-temporal for a lone decode, plain when several decodes share the chip -- the choice
        // launch_k24t_pass makes for the built-in kernels
        const std::string nt = p->nframes < K24_WORKERS_MIN_PLAIN ? "-DVH_JIT_NT=2" : "-DVH_JIT_NT=0";
        const char *names[4] = {"vh_jit_k24t_h_full", "vh_jit_k24t_h_part", "vh_jit_k24t_l_full", "vh_jit_k24t_l_part"};
        for (int i = 0; i < 4; i++)
            if (!vh::jit_function("acs_k24t.hip", {poly, nt}, names[i], &p->jit_fn[i], err)) return false;
        return true;
    }
    if (p->K == 15)
        return vh::jit_function("acs_k15.hip", {poly, p->code == VHIP_SPIRAL615 ? "-DVH_JIT_SPIRAL=true" : "-DVH_JIT_SPIRAL=false"}, "vh_jit_acs_k15",
                                &p->jit_fn[0], err);
    const char *traits = p->code == VHIP_KA9Q27 ? "Code27" : p->code == VHIP_KA9Q29 ? "Code29" : p->code == VHIP_SPIRAL47 ? "Code47"
                       : p->code == VHIP_SPIRAL49 ? "Code49" : p->code == VHIP_SPIRAL27 ? "CodeS27" : p->code == VHIP_SPIRAL29 ? "CodeS29" : nullptr;
    if (!traits || !vh::regs_lanes_supported(p->code, lb)) {
        *err = "no register kernel for this code / lanes per frame";
        return false;
    }
    const int nr = (int)(p->N >> (lb + 1));  // packed registers per lane: >= 32 takes the two-waves-per-SIMD scheduling hint, as launch_regs
    return vh::jit_function("acs_regs.hip",
                            {std::string("-DVH_JIT_CODE=") + traits, poly, "-DVH_JIT_LB=" + std::to_string(lb),
                             nr >= 32 ? "-DVH_JIT_ATTR=__attribute__((amdgpu_waves_per_eu(1,2)))" : "-DVH_JIT_ATTR="},
                            "vh_jit_acs_regs", &p->jit_fn[0], err);
}

void apply_variant(vhip_decoder *p, int variant, int lb) {
    p->variant = variant;
    if (variant == VHIP_VARIANT_REGS && p->K != 15) {
        p->regs_lb = lb;
        p->lay = vh::regs_layout(p->code, lb);
    }
}

// K=24: run `steps` trellis steps of frame f speculatively, replaying after each renormalisation event.
int k24_update_frame(vhip_decoder *p, int f, const unsigned char *d_syms, int steps, int row0) {
    const size_t NN = p->N;
    int16_t *buf[2] = {p->d_metrics + (size_t)f * 2 * NN, p->d_metrics + (size_t)f * 2 * NN + NN};
    unsigned char *rows = p->d_dec + (size_t)f * p->cap_rows * p->row_bytes;
    int cur = p->k24_cur[f];
    constexpr int BATCH = 128;
    int t = 0;
    while (t < steps) {
        const int end = std::min(steps, t + BATCH);
        for (int i = t; i < end; i++) {
            const int o = cur ^ ((i - t) & 1);
            HIP_TRY(vh::launch_k24_step(buf[o], buf[o ^ 1], rows + (size_t)(row0 + i) * p->row_bytes, d_syms, i, p->poly,
                                        p->d_flags, p->stream));
        }
        int pending = 0;
        HIP_TRY(hipMemcpyAsync(&pending, p->d_flags + vh::K24F_PENDING, sizeof(int), hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
        if (pending == 0) {
            cur ^= (end - t) & 1;
            t = end;
        } else {
            const int r = pending - 1;  // steps t..r ran; r+1.. returned early
            if (r < t || r >= end) return fail("K=24 renormalisation flag out of range");
            cur ^= (r - t + 1) & 1;
            HIP_TRY(vh::launch_k24_renorm(buf[cur], p->d_flags, p->stream));
            t = r + 1;
        }
    }
    p->k24_cur[f] = cur;
    return 0;
}

bool k24_multistep(int variant) { return variant == VHIP_VARIANT_HBM_FUSED || variant == VHIP_VARIANT_HBM_TILED; }

// K=24 multi-step passes: rows are grouped by phase = row mod 23 into passes of 4,4,4,4,7 steps (acs_k24f.hip) or of
// 9,14 steps (acs_k24t.hip).
//
// Speculative renormalisation: passes are enqueued ahead of the device, DEPTH at a time, and every pass
// stores (its sequence number, the sticky flag word) into one pinned host word when its flag-owning thread is done.  The
// host follows the stream by polling that word -- no event, no stream wait, so nothing but kernels enters the stream -- and
// commits each pass whose rows lie before the raised row.  When a pass raised the flag, every later pass has returned at
// once (it sees a flag of an earlier row); the stream is drained, the raising pass is replayed up to the flagged row, the
// metrics are renormalised and the pipeline restarts behind that row.
//
// One K24Run is the update of one frame on one stream as a resumable state machine: pump() enqueues what fits, looks at
// the progress word and returns, so ONE host thread keeps several frames in flight on several streams (vhip_update_dev
// with nframes > 1) -- no helper threads.
struct K24Run {
    struct Pass { int g, s_lo, s_hi, rel, in; unsigned seq; int ctl; };  // rel = index in this call of the row of stage s_lo; ctl: acs_k24t.hip
    vhip_decoder *p = nullptr;
    int f = -1, steps = 0, row0 = 0, slot = 0;
    const unsigned char *d_syms = nullptr;
    hipStream_t stream = nullptr;
    int *flags = nullptr;
    bool tiled = false;
    int16_t *buf[2] = {nullptr, nullptr};
    unsigned char *rows = nullptr;
    int cur = 0, t = 0, tt = 0, c = 0;  // committed buffer / rows committed / next row to enqueue / its input buffer
    size_t depth = 6;
    std::deque<Pass> inflight;
    unsigned spins = 0;
    int next_ctl = 0;  // tiled passes: renormalisation duties of the next pass to be enqueued (subtract a minimum while loading)

    void start(vhip_decoder *h, int frame, const unsigned char *syms, int nsteps, int first_row, hipStream_t st, int *fl) {
        p = h; f = frame; d_syms = syms; steps = nsteps; row0 = first_row; stream = st; flags = fl;
        tiled = h->variant == VHIP_VARIANT_HBM_TILED;
        const size_t NN = h->N;
        buf[0] = h->d_metrics + (size_t)frame * 2 * NN;
        buf[1] = buf[0] + NN;
        rows = h->d_dec + (size_t)frame * h->cap_rows * h->row_bytes;
        cur = c = h->k24_cur[frame];
        t = tt = 0;
        slot = (int)((fl - h->d_flags) / 4);  // 0: handle stream, 1..: the internal streams
        // passes kept enqueued: enough work (>= ~100 us) for the host to see a report and enqueue the next pass, no more --
        // every pass behind a raised flag is cancelled work (it still streams part of its tile in)
        // (tiled, measured per 2071-step frame: depth 2: 3.213 ms, 4: 3.227, 6: 3.248, 8: 3.260 -- the host keeps up easily;
        // 4 leaves a margin for a descheduled host thread)
        depth = tiled ? 4 : 12;
        inflight.clear();
        spins = 0;
        next_ctl = 0;
    }
    int group_of(int phi, int &first, int &np) const {
        if (tiled) {
            const int pass = vh::k24t_pass_of_phase(phi);
            first = vh::k24t_pass_first(pass);
            np = vh::k24t_pass_nphases(pass);
            return pass;
        }
        if (phi < 16) { first = (phi / 4) * 4; np = 4; return phi / 4; }
        first = 16; np = 7; return 4;
    }
    int launch(const Pass &q, bool report) {
        const long row_g0 = (long)row0 + q.rel - q.s_lo;  // absolute row of the group's phase 0 (may precede row0)
        const vh::K24Report rep{report ? p->h_report_dev + slot : nullptr, q.seq};
        if (tiled && p->jit) {
            // the run-time build of acs_k24t.hip for this handle's polynomials: same arguments, same grids as launch_k24t_pass
            const int16_t *oldm = buf[q.in];
            int16_t *newm = buf[q.in ^ 1];
            unsigned char *rw = rows + row_g0 * (long)p->row_bytes;
            const unsigned char *sy = d_syms + ((long)q.rel - q.s_lo) * 2;
            int rel0 = q.rel - q.s_lo, s_lo = q.s_lo, s_hi = q.s_hi;
            int *fl = flags;
            vh::K24Report rp = rep;
            int ctl = q.ctl;
            void *args[] = {&oldm, &newm, &rw, &sy, &rel0, &s_lo, &s_hi, &fl, &rp, &ctl};
            const bool full = s_lo == 0 && s_hi == vh::k24t_pass_nphases(q.g) && ctl == 0;
            HIP_TRY(hipModuleLaunchKernel(p->jit_fn[q.g * 2 + (full ? 0 : 1)], q.g == 0 ? 256 : 512, 1, 1, q.g == 0 ? 512 : 256, 1, 1, 0, stream, args, nullptr));
        } else if (tiled) {
            HIP_TRY(vh::launch_k24t_pass(q.g, buf[q.in], buf[q.in ^ 1], rows + row_g0 * (long)p->row_bytes,
                                         d_syms + ((long)q.rel - q.s_lo) * 2, q.rel - q.s_lo, q.s_lo, q.s_hi, flags, rep, stream,
                                         p->nframes < K24_WORKERS_MIN_PLAIN, q.ctl));
        } else {
            HIP_TRY(vh::launch_k24f_pass(q.g, buf[q.in], buf[q.in ^ 1], rows + row_g0 * (long)p->row_bytes,
                                         d_syms + ((long)q.rel - q.s_lo) * 2, q.rel - q.s_lo, q.s_lo, q.s_hi, flags, rep, stream));
        }
        return 0;
    }
    // 1: the frame is complete (everything it needs is enqueued and committed); 0: waiting for the device; -1: error
    int pump() {
        volatile unsigned long long *h_word = p->h_report + slot;
        unsigned &seq = p->k24_seq[slot];
        for (;;) {
            if (t >= steps) {
                p->k24_cur[f] = cur;
                return 1;
            }
            while (inflight.size() < depth && tt < steps) {
                const int phi = (row0 + tt) % 23;
                int first, np;
                const int g = group_of(phi, first, np);
                Pass ps{g, phi - first, std::min(np, phi - first + (steps - tt)), tt, c, ++seq, next_ctl};
                next_ctl = 0;
                if (launch(ps, true) != 0) return -1;
                inflight.push_back(ps);
                tt += ps.s_hi - ps.s_lo;
                c ^= 1;
            }
            const Pass front = inflight.front();
            const unsigned long long w = *h_word;
            if ((int)((unsigned)(w >> 32) - front.seq) < 0) {  // the oldest pass in flight has not reported yet
                if ((++spins & 0x3fffu) == 0) {
                    // a device fault would leave the word unchanged for ever: ask the runtime now and then
                    const hipError_t qe = hipStreamQuery(stream);
                    if (qe != hipSuccess && qe != hipErrorNotReady) return fail("K=24: stream failed while waiting for a pass", qe);
                    if (qe == hipSuccess && (int)((unsigned)(*h_word >> 32) - front.seq) < 0) return fail("K=24: pass finished without reporting");
                }
                return 0;
            }
            spins = 0;
            const int pending = (int)(unsigned)(w & 0xffffffffull);
            const int adv = front.s_hi - front.s_lo;
            if (pending == 0 || pending - 1 >= front.rel + adv) {  // no flag, or raised by a later row: this pass stands
                t = front.rel + adv;
                cur = front.in ^ 1;
                inflight.pop_front();
                continue;
            }
            const int rr = pending - 1;  // renormalise after this row of the call
            if (rr < front.rel) return fail("K=24: renormalisation flag out of range");
            Pass redo = front;
            redo.s_hi = redo.s_lo + (rr - redo.rel) + 1;  // replay the raising pass up to and including row rr
            redo.seq = ++seq;
            if (tiled && rr + 1 < steps) {
                // Tiled passes renormalise among themselves (acs_k24t.hip ctl): the replay forms the minimum of its final
                // metrics in slot n % 3 and clears the flag when it is done -- it is enqueued behind the younger passes, which
                // have seen the flag and return at once, so nothing has to be waited for here --, and the next pass enqueued
                // subtracts that minimum while it loads and clears the slot event n + 2 will use.  A pass that raises the flag
                // again keeps its own subtract duty when it is replayed.
                const unsigned n = p->k24_events[slot]++;
                redo.ctl = (front.ctl & ~(3 | 64)) | (1 + (int)(n % 3)) | 64;
                if (launch(redo, false) != 0) return -1;
                next_ctl = ((1 + (int)(n % 3)) << 2) | ((1 + (int)((n + 2) % 3)) << 4);
                cur = redo.in ^ 1;
            } else {
                // the k24f passes, or a renormalisation behind the last row of this call (no pass left to subtract the
                // minimum): separate min / subtract kernels
                HIP_TRY(hipStreamSynchronize(stream));  // the younger passes have returned at once
                const unsigned n = tiled ? p->k24_events[slot]++ : 0u;
                if (!tiled) HIP_TRY(vh::launch_k24_flags_reset(flags, stream));
                redo.ctl = tiled ? (front.ctl & ~(3 | 64)) : 0;
                if (launch(redo, false) != 0) return -1;
                cur = redo.in ^ 1;
                // min-reduce, wrapping subtract, clear the flag (and, tiled, the slot event n + 2 will use)
                HIP_TRY(vh::launch_k24_renorm(buf[cur], flags, stream, (int)(n % 3), tiled ? (int)((n + 2) % 3) : 0));
            }
            t = rr + 1;
            tt = t;
            c = cur;
            inflight.clear();  // restart the pipeline behind the renormalised row
        }
    }
};

int k24f_update_frame(vhip_decoder *p, int f, const unsigned char *d_syms, int steps, int row0, hipStream_t stream, int *flags) {
    K24Run run;
    run.start(p, f, d_syms, steps, row0, stream, flags);
    int rc;
    while ((rc = run.pump()) == 0) std::this_thread::yield();
    return rc < 0 ? -1 : 0;
}

// several frames: up to K24_WORKERS of them in flight, one internal stream each, all pumped by the calling thread
int k24f_update_frames(vhip_decoder *p, const unsigned char *d_syms, size_t sym_stride, int steps, int row0) {
    constexpr int W = vhip_decoder::K24_WORKERS;
    K24Run runs[W];
    bool active[W] = {};
    int next = 0, live = 0;
    for (int w = 0; w < W; w++) {
        HIP_TRY(vh::launch_k24_flags_reset(p->d_flags + 4 * (w + 1), p->aux_stream[w]));
        p->k24_events[w + 1] = 0;
    }
    while (next < p->nframes || live > 0) {
        bool progressed = false;
        for (int w = 0; w < W; w++) {
            if (!active[w]) {
                if (next >= p->nframes) continue;
                runs[w].start(p, next, d_syms + (size_t)next * sym_stride, steps, row0, p->aux_stream[w], p->d_flags + 4 * (w + 1));
                next++;
                active[w] = true;
                live++;
            }
            const int rc = runs[w].pump();
            if (rc < 0) {
                for (int v = 0; v < W; v++) (void)hipStreamSynchronize(p->aux_stream[v]);
                return -1;
            }
            if (rc == 1) {
                active[w] = false;
                live--;
                progressed = true;
            }
        }
        if (!progressed) std::this_thread::yield();
    }
    // a pass reports when its flag-owning thread is done, not when its last workgroup is: drain the internal streams so that
    // whatever the caller enqueues next on the handle's stream (the chainback) finds every decision row in place
    for (int w = 0; w < W; w++) HIP_TRY(hipStreamSynchronize(p->aux_stream[w]));
    return 0;
}

}  // namespace

extern "C" {

const char *vhip_last_error(void) { return g_last_error.c_str(); }

int vhip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int vhip_code_K(int code) { return vh::code_info(code).K; }
int vhip_code_R(int code) { return vh::code_info(code).R; }

vhip_decoder *vhip_create(int code, const int *poly, int len, int nframes) {
    (void)hipGetLastError();  // see use_device
    const vh::CodeInfo ci = vh::code_info(code);
    if (ci.K == 0) {
        fail("create: unknown code");
        return nullptr;
    }
    if (!poly || len < 0 || nframes < 1) {
        fail("create: bad arguments");
        return nullptr;
    }
    for (int r = 0; r < ci.R; r++) {
        // butterfly symmetry needs taps on the newest and the oldest register bit (viterbi27_sse2.cpp:149-152)
        if (poly[r] <= 0 || !(poly[r] & 1) || !((poly[r] >> (ci.K - 1)) & 1) || (poly[r] >> ci.K)) {
            fail("create: every polynomial must be a positive K-bit mask with bit 0 and bit K-1 set");
            return nullptr;
        }
    }
    if (vhip_device_count() < 1) {
        fail("create: no HIP device visible (this backend has no CPU fallback)");
        return nullptr;
    }
    vhip_decoder *p = new vhip_decoder();
    p->code = code;
    p->K = ci.K;
    p->R = ci.R;
    p->N = 1u << (ci.K - 1);
    for (int r = 0; r < ci.R; r++) p->poly[r] = poly[r];
    p->len = len;
    p->nframes = nframes;
    p->incremental = ci.incremental;
    // rows: ka9q27/29/615 and spiral allocate len+K-1 (viterbi27_sse2.cpp:72, spiral47.cpp:79), ka9q224 len (:61)
    p->cap_rows = (code == VHIP_KA9Q224) ? len : len + ci.K - 1;
    p->row_bytes = p->N / 8;
    p->frames_padded = (code == VHIP_KA9Q224) ? nframes : ((nframes + 63) / 64) * 64;
    apply_variant(p, auto_variant(p), auto_regs_lb(code, nframes));
    (void)hipGetDevice(&p->device);
    if (p->variant != VHIP_VARIANT_WAVE && !fast_poly_supported(p)) {
        // other polynomials than the harness set: the same fast kernel, compiled for them now (jit.hip); if that is not
        // possible here (no sources / no hipcc / VHIP_JIT=0) the handle keeps the any-polynomial kernel chosen above
        const int lb = auto_regs_lb(code, nframes);
        std::string why;
        if (setup_jit(p, lb, &why)) {
            p->jit = true;
            apply_variant(p, code == VHIP_KA9Q224 ? VHIP_VARIANT_HBM_TILED : VHIP_VARIANT_REGS, lb);
        } else {
            // not an error: the handle works, on the any-polynomial kernels.  vhip_last_error() says why until the next failure.
            g_last_error = "viterbi_hip: no run-time build of the fast kernel for these polynomials: " + why;
            if (getenv("VHIP_VERBOSE")) fprintf(stderr, "%s -- using the any-polynomial kernels\n", g_last_error.c_str());
        }
    }
    const size_t dec_bytes = (size_t)p->frames_padded * (size_t)p->cap_rows * p->row_bytes;
    const size_t met_bytes = (size_t)nframes * p->N * sizeof(int16_t) * (code == VHIP_KA9Q224 ? 2 : 1);
    // The decision history is allocated by the first call that needs it (ensure_history): a handle that is only ever used
    // for the fused windowed decode never holds one, so its batch is not bounded by N/8 bytes per frame-step.
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&p->d_metrics), met_bytes);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&p->d_flags), sizeof(int) * 32);
    if (e == hipSuccess) e = hipMemset(p->d_flags, 0, sizeof(int) * 32);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);  // the fill is complete before any (possibly non-blocking) stream uses the words
    if (e != hipSuccess) {
        fail("create: device allocation", e);
        vhip_delete(p);
        return nullptr;
    }
    if (code == VHIP_KA9Q224 && nframes > 1)
        for (int w = 0; w < vhip_decoder::K24_WORKERS && e == hipSuccess; w++) e = hipStreamCreateWithFlags(&p->aux_stream[w], hipStreamNonBlocking);
    if (code == VHIP_KA9Q224) {
        if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&p->h_report), sizeof(unsigned long long) * 8 * (1 + vhip_decoder::K24_WORKERS), hipHostMallocMapped);
        if (e == hipSuccess) memset(p->h_report, 0, sizeof(unsigned long long) * 8 * (1 + vhip_decoder::K24_WORKERS));
        if (e == hipSuccess) e = hipHostGetDevicePointer(reinterpret_cast<void **>(&p->h_report_dev), p->h_report, 0);
    }
    if (e != hipSuccess) {
        fail("create: stream", e);
        vhip_delete(p);
        return nullptr;
    }
    p->total_bytes = met_bytes + 64;
    (void)dec_bytes;
    p->k24_cur.assign(nframes, 0);
    if (vhip_init(p, 0) != 0) {
        vhip_delete(p);
        return nullptr;
    }
    if (hipStreamSynchronize(p->stream) != hipSuccess) {
        fail("create: init failed");
        vhip_delete(p);
        return nullptr;
    }
    return p;
}

void vhip_delete(vhip_decoder *p) {
    if (!p) return;  // delete(NULL) is a no-op in the reference too (viterbi27_sse2.cpp:108-115)
    (void)use_device(p);
    if (p->depth > 1) {
        for (int i = 0; i < p->depth; i++) {
            if (p->slots[i].stream) (void)hipStreamSynchronize(p->slots[i].stream);
            if (p->slots[i].d_dec) (void)hipFree(p->slots[i].d_dec);
            if (p->slots[i].d_metrics) (void)hipFree(p->slots[i].d_metrics);
            if (p->slots[i].stream) (void)hipStreamDestroy(p->slots[i].stream);
            if (p->slots[i].done) (void)hipEventDestroy(p->slots[i].done);
        }
    } else {
        if (p->d_dec) (void)hipFree(p->d_dec);
        if (p->d_metrics) (void)hipFree(p->d_metrics);
    }
    if (p->ev_input) (void)hipEventDestroy(p->ev_input);
    for (hipEvent_t e : p->ev_pool) (void)hipEventDestroy(e);
    for (auto &pr : p->t_upd) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (auto &pr : p->t_cb) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    if (p->d_flags) (void)hipFree(p->d_flags);
    if (p->d_cbseg) (void)hipFree(p->d_cbseg);
    for (int w = 0; w < vhip_decoder::K24_WORKERS; w++)
        if (p->aux_stream[w]) (void)hipStreamDestroy(p->aux_stream[w]);
    if (p->h_report) (void)hipHostFree(p->h_report);
    if (p->d_syms_stage) (void)hipFree(p->d_syms_stage);
    if (p->d_data_stage) (void)hipFree(p->d_data_stage);
    if (p->d_ring) (void)hipFree(p->d_ring);
    delete p;
}

int vhip_set_stream(vhip_decoder *p, void *stream) {
    if (!p) return fail("set_stream: NULL handle");
    p->stream = reinterpret_cast<hipStream_t>(stream);
    return 0;
}

int vhip_sync(vhip_decoder *p) {
    if (!p) return fail("sync: NULL handle");
    if (use_device(p) != 0) return -1;
    return sync_all(p);
}

int vhip_status(const vhip_decoder *p) { return p ? p->status : -1; }

int vhip_enable_timing(vhip_decoder *p, int on) {
    if (!p) return fail("enable_timing: NULL handle");
    p->timing = on != 0;
    return 0;
}

int vhip_read_timing(vhip_decoder *p, double *update_ms_sum, int *n_update, double *chainback_ms_sum, int *n_chainback) {
    if (!p) return fail("read_timing: NULL handle");
    if (use_device(p) != 0) return -1;
    if (sync_all(p) != 0) return -1;
    double su = 0.0, sc = 0.0;
    for (auto &pr : p->t_upd) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, pr.first, pr.second));
        su += ms;
    }
    for (auto &pr : p->t_cb) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, pr.first, pr.second));
        sc += ms;
    }
    if (update_ms_sum) *update_ms_sum = su;
    if (n_update) *n_update = (int)p->t_upd.size();
    if (chainback_ms_sum) *chainback_ms_sum = sc;
    if (n_chainback) *n_chainback = (int)p->t_cb.size();
    for (auto &pr : p->t_upd) { p->ev_pool.push_back(pr.first); p->ev_pool.push_back(pr.second); }
    for (auto &pr : p->t_cb) { p->ev_pool.push_back(pr.first); p->ev_pool.push_back(pr.second); }
    p->t_upd.clear();
    p->t_cb.clear();
    return 0;
}

int vhip_get_pipeline_depth(const vhip_decoder *p) { return p ? p->depth : -1; }

int vhip_set_pipeline_depth(vhip_decoder *p, int depth) {
    if (!p) return fail("set_pipeline_depth: NULL handle");
    if (use_device(p) != 0) return -1;
    if (depth < 1 || depth > vhip_decoder::MAX_DEPTH) return fail("set_pipeline_depth: depth must be 1..3");
    if (p->depth != 1 || p->pos != 0) return fail("set_pipeline_depth: only once, on a fresh handle");
    if (depth == 1) return 0;
    if (p->code == VHIP_KA9Q224) return fail("set_pipeline_depth: K=24 handles keep several frames in flight by themselves");
    if (ensure_history(p) != 0) return -1;
    HIP_TRY(hipStreamSynchronize(p->stream));
    const size_t dec_bytes = (size_t)p->frames_padded * (size_t)p->cap_rows * p->row_bytes;
    const size_t met_bytes = (size_t)p->nframes * p->N * sizeof(int16_t);
    const size_t bytes_before = p->total_bytes;
    p->slots[0].d_dec = p->d_dec;
    p->slots[0].d_metrics = p->d_metrics;
    hipError_t e = hipEventCreateWithFlags(&p->ev_input, hipEventDisableTiming);
    for (int i = 0; i < depth && e == hipSuccess; i++) {
        vhip_decoder::Slot &sl = p->slots[i];
        if (i > 0) {
            e = hipMalloc(reinterpret_cast<void **>(&sl.d_dec), dec_bytes ? dec_bytes : 16);
            if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&sl.d_metrics), met_bytes);
            if (e == hipSuccess) e = hipMemset(sl.d_dec, 0, dec_bytes ? dec_bytes : 16);
            if (e == hipSuccess) e = vh::launch_init_metrics(sl.d_metrics, p->N, p->nframes, init_all_of(p->code), init_start_of(p->code), 0, p->stream);
            if (e == hipSuccess) p->total_bytes += dec_bytes + met_bytes;
        }
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.done, hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(p->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);  // the hipMemset fills above: the slot streams are non-blocking
    if (e != hipSuccess) {
        // a slot without its stream or history must never be rotated into: release what was created and stay serial
        for (int i = 0; i < depth; i++) {
            vhip_decoder::Slot &sl = p->slots[i];
            if (i > 0 && sl.d_dec) (void)hipFree(sl.d_dec);
            if (i > 0 && sl.d_metrics) (void)hipFree(sl.d_metrics);
            if (sl.stream) (void)hipStreamDestroy(sl.stream);
            if (sl.done) (void)hipEventDestroy(sl.done);
            sl = vhip_decoder::Slot{};
        }
        if (p->ev_input) (void)hipEventDestroy(p->ev_input);
        p->ev_input = nullptr;
        p->total_bytes = bytes_before;
        p->depth = 1;
        p->cur_slot = 0;
        return fail("set_pipeline_depth: allocation (the handle stays at depth 1)", e);
    }
    p->depth = depth;  // from here on vhip_delete releases the slots
    p->cur_slot = 0;
    return 0;
}

// Orders everything the handle has in flight on its internal streams before whatever the caller enqueues next on the
// handle's stream (device-side wait; the host does not block).
int vhip_join(vhip_decoder *p) {
    if (!p) return fail("join: NULL handle");
    if (use_device(p) != 0) return -1;
    for (int i = 0; i < p->depth && p->depth > 1; i++) {
        HIP_TRY(hipEventRecord(p->slots[i].done, p->slots[i].stream));
        HIP_TRY(hipStreamWaitEvent(p->stream, p->slots[i].done, 0));
    }
    return 0;
}

int vhip_set_variant(vhip_decoder *p, int variant) {
    if (!p) return fail("set_variant: NULL handle");
    if (use_device(p) != 0) return -1;
    if (p->pos != 0) return fail("set_variant: only before the first update after init");
    // bits 8.. optionally carry 1 + log2(lanes per frame) for the REGS kernels
    const int lb_req = (variant >> 8) - 1;
    variant &= 0xff;
    if (variant == VHIP_VARIANT_AUTO) variant = auto_variant(p);
    if (variant == VHIP_VARIANT_WAVE) {
        if (!vh::wave_code_supported(p->code)) return fail("set_variant: the wave-per-frame kernels exist for the K=7 and K=9 codes only");
        p->jit = false;
        apply_variant(p, variant, 0);
        return 0;
    }
    const bool wants_fast = (p->code == VHIP_KA9Q224) ? variant == VHIP_VARIANT_HBM_TILED : variant == VHIP_VARIANT_REGS;
    if (wants_fast && !fast_poly_supported(p) && (p->code == VHIP_KA9Q224 || p->K == 15 || p->K <= 9)) {
        const int lb = lb_req >= 0 ? lb_req : auto_regs_lb(p->code, p->nframes);
        std::string why;
        if (!setup_jit(p, lb, &why)) return fail(("set_variant: these polynomials need a run-time build of the fast kernel: " + why).c_str());
        p->jit = true;
        apply_variant(p, variant, lb);
        return 0;
    }
    p->jit = false;
    if (p->code == VHIP_KA9Q224) {
        if (variant != VHIP_VARIANT_HBM && !k24_multistep(variant)) return fail("set_variant: K=24 supports only the HBM variants");
        if (variant == VHIP_VARIANT_HBM_TILED && !vh::k24t_poly_supported(p->poly))
            return fail("set_variant: the tiled K=24 kernel needs the harness polynomials");
        if (variant == VHIP_VARIANT_HBM_FUSED && !vh::k24f_poly_supported(p->poly))
            return fail("set_variant: the fused K=24 kernel needs the harness polynomials");
    } else if (variant == VHIP_VARIANT_REGS && p->K == 15) {
        if (!vh::k15_poly_supported(p->poly)) return fail("set_variant: the K=15 register kernel needs the harness polynomials");
    } else if (variant == VHIP_VARIANT_REGS) {
        if (p->K > 9 || !vh::regs_poly_supported(p->code, p->poly))
            return fail("set_variant: REGS kernels exist for K<=9 with the harness polynomials only");
        if (lb_req >= 0 && !vh::regs_lanes_supported(p->code, lb_req)) return fail("set_variant: unsupported lanes per frame");
    } else if (variant != VHIP_VARIANT_LDS) {
        return fail("set_variant: unsupported for this code");
    }
    apply_variant(p, variant, lb_req >= 0 ? lb_req : auto_regs_lb(p->code, p->nframes));
    return 0;
}
int vhip_get_variant(const vhip_decoder *p) { return p ? p->variant : -1; }
int vhip_rows_written(const vhip_decoder *p) { return p ? p->pos : -1; }
size_t vhip_device_bytes(const vhip_decoder *p) { return p ? p->total_bytes : 0; }

// init_viterbi27_sse2 (viterbi27_sse2.cpp:42-54) for every frame of the handle
int vhip_init(vhip_decoder *p, int starting_state) {
    StatusScope status_scope(p);
    if (!p) return fail("init: NULL handle");  // viterbi224_sse2.cpp:36-37 returns -1 on NULL
    if (use_device(p) != 0) return -1;
    const unsigned start = (unsigned)starting_state & (p->N - 1);
    const int ia = init_all_of(p->code), is = init_start_of(p->code);
    if (p->depth > 1) {
        // a new decode begins: take the next buffer set and its stream (the previous decode's chainback may still be
        // walking the other history)
        p->slots[p->cur_slot].pos = p->pos;
        p->cur_slot = (p->cur_slot + 1) % p->depth;
        p->d_dec = p->slots[p->cur_slot].d_dec;
        p->d_metrics = p->slots[p->cur_slot].d_metrics;
        if (order_behind_caller(p) != 0) return -1;
    }
    if (p->code == VHIP_KA9Q224) {
        // only the first half of each frame's ping-pong needs filling; mark it current
        for (int f = 0; f < p->nframes; f++) {
            HIP_TRY(vh::launch_init_metrics(p->d_metrics + (size_t)f * 2 * p->N, p->N, 1, ia, is, start, p->stream));
            p->k24_cur[f] = 0;
        }
        HIP_TRY(vh::launch_k24_flags_reset(p->d_flags, p->stream));
        p->k24_events[0] = 0;
    } else {
        HIP_TRY(vh::launch_init_metrics(p->d_metrics, p->N, p->nframes, ia, is, start, p->run_stream()));
    }
    p->pos = 0;
    return 0;
}

int vhip_update_dev(vhip_decoder *p, const unsigned char *d_syms, int nbits) {
    StatusScope status_scope(p);
    if (!p) return fail("update: NULL handle");
    if (use_device(p) != 0) return -1;
    if (nbits <= 0) return 0;
    int steps = nbits, row0 = p->pos;
    if (!p->incremental) {  // spiral47.cpp:536-538: restart at row 0, nbits/2 double steps
        row0 = 0;
        steps = (nbits / 2) * 2;
    }
    if (row0 + steps > p->cap_rows) return fail("update: more trellis steps than the handle was created for");
    const size_t sym_stride = (size_t)nbits * p->R;
    if (order_behind_caller(p) != 0) return -1;
    if (ensure_history(p) != 0) return -1;
    struct TimeScope {  // the closing event is recorded on every path out of the launches below
        hipEvent_t e1;
        hipStream_t st;
        ~TimeScope() { timing_end(e1, st); }
    } time_scope{timing_begin(p, p->t_upd, p->run_stream()), p->run_stream()};
    if (p->code == VHIP_KA9Q224) {
        if (k24_multistep(p->variant) && p->nframes > 1) {
            // A single K=24 decode leaves the chip under-occupied between its load / compute / store phases (three decodes
            // in flight run 2x faster per frame than one), so K24_WORKERS frames are decoded at a time on internal streams.
            HIP_TRY(hipStreamSynchronize(p->stream));  // the caller's symbols are ready
            if (k24f_update_frames(p, d_syms, sym_stride, steps, row0) != 0) return -1;
        } else {
            for (int f = 0; f < p->nframes; f++) {
                const int rc = k24_multistep(p->variant)
                                   ? k24f_update_frame(p, f, d_syms + (size_t)f * sym_stride, steps, row0, p->stream, p->d_flags)
                                   : k24_update_frame(p, f, d_syms + (size_t)f * sym_stride, steps, row0);
                if (rc != 0) return -1;
            }
        }
    } else if (p->variant == VHIP_VARIANT_WAVE) {
        vh::AcsLdsArgs a;
        a.syms = d_syms;
        a.sym_stride = sym_stride;
        a.nsteps = steps;
        a.row0 = row0;
        a.cap_rows = p->cap_rows;
        a.nframes = p->nframes;
        a.dec = p->d_dec;
        a.metrics = p->d_metrics;
        for (int r = 0; r < 8; r++) a.poly[r] = p->poly[r];
        HIP_TRY(vh::launch_acs_wave(p->code, a, p->run_stream()));
    } else if (p->variant == VHIP_VARIANT_REGS && p->K == 15) {
        vh::AcsK15Args a;
        a.syms = d_syms;
        a.sym_stride = sym_stride;
        a.nsteps = steps;
        a.row0 = row0;
        a.cap_rows = p->cap_rows;
        a.nframes = p->nframes;
        a.dec = p->d_dec;
        a.metrics = p->d_metrics;
        if (p->jit) {
            void *args[] = {&a};
            HIP_TRY(hipModuleLaunchKernel(p->jit_fn[0], (unsigned)a.nframes, 1, 1, 128, 1, 1, 0, p->run_stream(), args, nullptr));
        } else {
            HIP_TRY(vh::launch_acs_k15(a, p->code == VHIP_SPIRAL615, p->run_stream()));
        }
    } else if (p->variant == VHIP_VARIANT_REGS) {
        vh::AcsRegsArgs a;
        a.syms = d_syms;
        a.sym_stride = sym_stride;
        a.nsteps = steps;
        a.row0 = row0;
        a.cap_rows = p->cap_rows;
        a.nframes = p->nframes;
        a.dec = p->d_dec;
        a.metrics = p->d_metrics;
        if (p->jit) {
            const int waves = (a.nframes + p->lay.fpw - 1) / p->lay.fpw;
            void *args[] = {&a};
            HIP_TRY(hipModuleLaunchKernel(p->jit_fn[0], (unsigned)((waves + 3) / 4), 1, 1, 256, 1, 1, 0, p->run_stream(), args, nullptr));
        } else {
            HIP_TRY(vh::launch_acs_regs(p->code, p->regs_lb, a, p->run_stream()));
        }
    } else {
        vh::AcsLdsArgs a;
        a.syms = d_syms;
        a.sym_stride = sym_stride;
        a.nsteps = steps;
        a.row0 = row0;
        a.cap_rows = p->cap_rows;
        a.nframes = p->nframes;
        a.dec = p->d_dec;
        a.metrics = p->d_metrics;
        for (int r = 0; r < 8; r++) a.poly[r] = p->poly[r];
        HIP_TRY(vh::launch_acs_lds(p->code, a, p->run_stream()));
    }
    p->pos = row0 + steps;
    return 0;
}

int vhip_chainback_dev(vhip_decoder *p, unsigned char *d_data, unsigned int nbits, unsigned int endstate) {
    StatusScope status_scope(p);
    if (!p) return fail("chainback: NULL handle");
    if (use_device(p) != 0) return -1;
    if (nbits == 0) return 0;
    if (order_behind_caller(p) != 0) return -1;
    if (ensure_history(p) != 0) return -1;
    struct TimeScope {
        hipEvent_t e1;
        hipStream_t st;
        ~TimeScope() { timing_end(e1, st); }
    } time_scope{timing_begin(p, p->t_cb, p->run_stream()), p->run_stream()};
    if (k24_multistep(p->variant)) {
        const bool tiled = p->variant == VHIP_VARIANT_HBM_TILED;
        vh::ChainbackRowsArgs a;
        a.dec = p->d_dec;
        a.cap_rows = p->cap_rows;
        a.rows_written = p->pos;
        a.nframes = p->nframes;
        a.data = d_data;
        a.data_stride = (nbits + 7) / 8;
        a.nbits = nbits;
        a.endstate = endstate;
        a.K = p->K;
        a.k224 = 1;
        if (setup_segments(p, a) != 0) return -1;
        HIP_TRY(vh::launch_chainback_spec(tiled ? vh::CB_LAY_K24T : vh::CB_LAY_K24F, a, p->stream));
        return 0;
    }
    if (p->variant == VHIP_VARIANT_WAVE) {
        vh::ChainbackRowsArgs a;
        a.dec = p->d_dec;
        a.cap_rows = p->cap_rows;
        a.rows_written = p->pos;
        a.nframes = p->nframes;
        a.data = d_data;
        a.data_stride = (nbits + 7) / 8;
        a.nbits = nbits;
        a.endstate = endstate;
        a.K = p->K;
        a.k224 = 0;
        HIP_TRY(vh::launch_chainback_wave(a, p->run_stream()));
        return 0;
    }
    if (p->variant == VHIP_VARIANT_REGS && p->K == 15) {
        vh::ChainbackRowsArgs a;
        a.dec = p->d_dec;
        a.cap_rows = p->cap_rows;
        a.rows_written = p->pos;
        a.nframes = p->nframes;
        a.data = d_data;
        a.data_stride = (nbits + 7) / 8;
        a.nbits = nbits;
        a.endstate = endstate;
        a.K = p->K;
        a.k224 = 0;
        a.k15_sign_bytes = p->code == VHIP_KA9Q615;
        if (setup_segments(p, a) != 0) return -1;
        HIP_TRY(vh::launch_chainback_spec(a.k15_sign_bytes ? vh::CB_LAY_K15_SIGN_BYTES : vh::CB_LAY_K15, a, p->run_stream()));
        return 0;
    }
    if (p->variant == VHIP_VARIANT_REGS) {
        vh::ChainbackRegsArgs a;
        a.dec = p->d_dec;
        a.lay = p->lay;
        a.cap_rows = p->cap_rows;
        a.rows_written = p->pos;
        a.nframes = p->nframes;
        a.data = d_data;
        a.data_stride = (nbits + 7) / 8;
        a.nbits = nbits;
        a.endstate = endstate;
        a.K = p->K;
        HIP_TRY(vh::launch_chainback_regs(a, p->run_stream()));
        return 0;
    }
    vh::ChainbackRowsArgs a;
    a.dec = p->d_dec;
    a.cap_rows = p->cap_rows;
    a.rows_written = p->pos;
    a.nframes = p->nframes;
    a.data = d_data;
    a.data_stride = (nbits + 7) / 8;
    a.nbits = nbits;
    a.endstate = endstate;
    a.K = p->K;
    a.k224 = (p->code == VHIP_KA9Q224);
    if (p->K >= 15) {
        if (setup_segments(p, a) != 0) return -1;
        HIP_TRY(vh::launch_chainback_spec(vh::CB_LAY_NATURAL, a, p->run_stream()));
    } else HIP_TRY(vh::launch_chainback_rows(a, p->run_stream()));
    return 0;
}

int vhip_set_chainback_segments(vhip_decoder *p, int seg_bits, int warmup_rows) {
    if (!p) return fail("set_chainback_segments: NULL handle");
    if (seg_bits > 0 && (seg_bits & 7)) return fail("set_chainback_segments: seg_bits must be a multiple of 8 (0 = one walk per frame, < 0 = default)");
    p->cbseg_bits = seg_bits;
    p->cbseg_ovl = warmup_rows;
    return 0;
}

int vhip_chainback_rewalked(vhip_decoder *p, int *nseg) {
    if (!p) return fail("chainback_rewalked: NULL handle");
    if (nseg) *nseg = p->cbseg_last_nseg;
    if (p->cbseg_last_nseg == 0) return 0;
    if (use_device(p) != 0 || sync_all(p) != 0) return -1;
    std::vector<unsigned> w((size_t)p->nframes * 2);  // [frame][ticket, re-walked]
    HIP_TRY(hipMemcpy(w.data(), p->cbseg_last, w.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
    int total = 0;
    for (int f = 0; f < p->nframes; f++) total += (int)w[(size_t)f * 2 + 1];
    return total;
}

// ---------------------------------------------------------------- fused sliding-window decode
int vhip_is_runtime_specialised(const vhip_decoder *p) { return p ? (p->jit ? 1 : 0) : -1; }
int vhip_runtime_build_sources_ok(void) {
    std::string why;
    const int rc = vh::jit_sources_match(&why);
    if (rc != 1) g_last_error = "viterbi_hip: " + why;
    return rc;
}

namespace {
bool window_params(const vhip_decoder *p, int *depth, int *block) {
    if (!p) return false;
    if (p->K == 15 && vh::k15_poly_supported(p->poly)) {
        vh::windowed_k15_params(depth, block);
        return true;
    }
    if (p->K > 9 || !vh::regs_poly_supported(p->code, p->poly)) return false;
    int lb;
    vh::windowed_params(p->code, depth, block, &lb);
    return true;
}
}  // namespace
int vhip_window_depth(const vhip_decoder *p) {
    int d, b;
    return window_params(p, &d, &b) ? d : -1;
}
int vhip_window_block(const vhip_decoder *p) {
    int d, b;
    return window_params(p, &d, &b) ? b : -1;
}

int vhip_decode_windowed_dev(vhip_decoder *p, const unsigned char *d_syms, unsigned int nbits, unsigned char *d_data) {
    StatusScope status_scope(p);
    if (!p) return fail("decode_windowed: NULL handle");
    if (use_device(p) != 0) return -1;
    if (vhip_window_depth(p) < 0) return fail("decode_windowed: built for K <= 9 and K = 15 with the harness polynomials");
    if (nbits == 0) return 0;
    const int steps = (int)nbits + p->K - 1;
    if (steps > p->cap_rows) return fail("decode_windowed: more trellis steps than the handle was created for");
    if (order_behind_caller(p) != 0) return -1;
    struct TimeScope {
        hipEvent_t e1;
        hipStream_t st;
        ~TimeScope() { timing_end(e1, st); }
    } time_scope{timing_begin(p, p->t_upd, p->run_stream()), p->run_stream()};
    const int steps_run = p->incremental ? steps : (steps / 2) * 2;  // update_spiral47 drops an odd last step (spiral47.cpp:536-538)
    if (p->K == 15) {
        // the decision ring of the resident workgroups (320 KiB each, at most 1024): allocated on first use, kept with the handle
        // One ring set per pipeline slot: on a pipelined handle vhip_init() rotates the slot, and the persistent grids of two
        // consecutive decodes then run at the same time on two internal streams.
        const size_t need = vh::windowed_k15_ring_bytes(p->nframes);
        if (p->ring_bytes < need * (size_t)p->depth && sync_all(p) != 0) return -1;  // a smaller ring may still be in use
        if (ensure_stage(&p->d_ring, &p->ring_bytes, need * (size_t)p->depth, &p->total_bytes) != 0) return -1;
        unsigned char *ring = p->d_ring + (size_t)(p->depth > 1 ? p->cur_slot : 0) * need;
        HIP_TRY(vh::launch_decode_windowed_k15(p->code == VHIP_SPIRAL615, d_syms, (size_t)steps * p->R, steps_run, p->nframes, d_data, (nbits + 7) / 8,
                                               nbits, reinterpret_cast<unsigned *>(ring), p->run_stream()));
        return 0;
    }
    HIP_TRY(vh::launch_decode_windowed(p->code, d_syms, (size_t)steps * p->R, steps_run, p->nframes, d_data, (nbits + 7) / 8, nbits, p->run_stream()));
    return 0;
}

int vhip_update(vhip_decoder *p, const unsigned char *syms, int nbits) {
    StatusScope status_scope(p);
    if (!p) return fail("update: NULL handle");
    if (use_device(p) != 0) return -1;
    if (nbits <= 0) return 0;
    const size_t bytes = (size_t)p->nframes * (size_t)nbits * p->R;
    if (ensure_stage(&p->d_syms_stage, &p->syms_stage_bytes, bytes, &p->total_bytes) != 0) return -1;
    if (p->depth > 1 && sync_all(p) != 0) return -1;  // the staging buffer is shared between the decodes in flight
    HIP_TRY(hipMemcpyAsync(p->d_syms_stage, syms, bytes, hipMemcpyHostToDevice, p->stream));
    if (vhip_update_dev(p, p->d_syms_stage, nbits) != 0) return -1;
    HIP_TRY(hipStreamSynchronize(p->run_stream()));
    return 0;
}

int vhip_chainback(vhip_decoder *p, unsigned char *data, unsigned int nbits, unsigned int endstate) {
    StatusScope status_scope(p);
    if (!p) return fail("chainback: NULL handle");
    if (use_device(p) != 0) return -1;
    int ret = 0;
    if (p->code == VHIP_KA9Q615 && p->nframes == 1) {
        // chainback_viterbi615_sse2 returns old_metrics->s[endstate]            viterbi615_sse2.cpp:76,90
        int16_t m = 0;
        HIP_TRY(hipMemcpyAsync(&m, p->d_metrics + (endstate % p->N), sizeof(m), hipMemcpyDeviceToHost, p->run_stream()));
        HIP_TRY(hipStreamSynchronize(p->run_stream()));
        ret = m;
    }
    if (nbits == 0) return ret;
    const size_t stride = (nbits + 7) / 8;
    const size_t bytes = (size_t)p->nframes * stride;
    if (ensure_stage(&p->d_data_stage, &p->data_stage_bytes, bytes, &p->total_bytes) != 0) return -1;
    if (p->depth > 1 && sync_all(p) != 0) return -1;
    if (vhip_chainback_dev(p, p->d_data_stage, nbits, endstate) != 0) return -1;
    HIP_TRY(hipMemcpyAsync(data, p->d_data_stage, bytes, hipMemcpyDeviceToHost, p->run_stream()));
    HIP_TRY(hipStreamSynchronize(p->run_stream()));
    return ret;
}

int vhip_read_decision_rows(vhip_decoder *p, int frame, int row0, int nrows, unsigned char *out) {
    if (!p) return fail("read_decision_rows: NULL handle");
    if (use_device(p) != 0) return -1;
    if (frame < 0 || frame >= p->nframes || row0 < 0 || nrows < 0 || row0 + nrows > p->cap_rows)
        return fail("read_decision_rows: out of range");
    if (ensure_history(p) != 0) return -1;
    if (sync_all(p) != 0) return -1;
    if (k24_multistep(p->variant)) {
        // position bitmap of acs_k24f.hip / acs_k24t.hip -> natural bitmap
        const bool tiled = p->variant == VHIP_VARIANT_HBM_TILED;
        const int NB = 23;
        std::vector<unsigned char> raw(p->row_bytes);
        memset(out, 0, (size_t)nrows * p->row_bytes);
        for (int i = 0; i < nrows; i++) {
            HIP_TRY(hipMemcpy(raw.data(), p->d_dec + ((size_t)frame * p->cap_rows + row0 + i) * p->row_bytes, p->row_bytes, hipMemcpyDeviceToHost));
            const int rot = (row0 + i + 1) % NB, phi = (row0 + i) % NB;
            unsigned char *o = out + (size_t)i * p->row_bytes;
            const unsigned *rw = reinterpret_cast<const unsigned *>(raw.data());
            for (unsigned n = 0; n < p->N; n++) {
                const unsigned pos = rot == 0 ? n : (((n >> rot) | (n << (NB - rot))) & (p->N - 1));
                unsigned widx, wbit;
                if (tiled) vh::k24t_locate(pos, phi, widx, wbit);
                else vh::k24f_locate(pos, phi, widx, wbit);
                if ((rw[widx] >> wbit) & 1u) o[n >> 3] |= (unsigned char)(1u << (n & 7));
            }
        }
        return 0;
    }
    if (p->variant == VHIP_VARIANT_WAVE) {
        // position-ordered ballots of acs_wave.hip (N/64 64-bit words per row: word = position >> 6) -> natural bitmap: new state n of
        // row r sits at position rotr^((r+1) mod (K-1))(n)
        const int NB = p->K - 1;
        const size_t words = p->N / 64;
        std::vector<unsigned long long> raw((size_t)nrows * words);
        HIP_TRY(hipMemcpy(raw.data(), p->d_dec + ((size_t)frame * p->cap_rows + row0) * p->row_bytes, raw.size() * 8, hipMemcpyDeviceToHost));
        memset(out, 0, (size_t)nrows * p->row_bytes);
        for (int i = 0; i < nrows; i++) {
            const int rot = (row0 + i + 1) % NB;
            for (unsigned n = 0; n < p->N; n++) {
                const unsigned pos = rot == 0 ? n : (((n >> rot) | (n << (NB - rot))) & (p->N - 1));
                if ((raw[(size_t)i * words + (pos >> 6)] >> (pos & 63u)) & 1ull) out[(size_t)i * p->row_bytes + (n >> 3)] |= (unsigned char)(1u << (n & 7));
            }
        }
        return 0;
    }
    if (p->variant == VHIP_VARIANT_REGS && p->K == 15) {
        // [row][word][thread] of acs_k15.hip -> natural bitmap
        const int NB = 14;
        std::vector<unsigned> raw((size_t)nrows * 512);
        HIP_TRY(hipMemcpy(raw.data(), p->d_dec + ((size_t)frame * p->cap_rows + row0) * 2048, raw.size() * 4, hipMemcpyDeviceToHost));
        memset(out, 0, (size_t)nrows * p->row_bytes);
        for (int i = 0; i < nrows; i++) {
            const int rot = (row0 + i + 1) % NB, phi = (row0 + i) % NB;
            for (unsigned n = 0; n < p->N; n++) {
                const unsigned pos = rot == 0 ? n : (((n >> rot) | (n << (NB - rot))) & (p->N - 1));
                const unsigned t = phi < 7 ? (pos & 127u) : (pos >> 7), q = phi < 7 ? (pos >> 7) : (pos & 127u);
                const unsigned rho = q >> 1, h = q & 1u;
                const unsigned word = raw[(size_t)i * 512 + (rho >> 4) * 128 + t];
                if ((word >> vh::k15_decision_bit(p->code == VHIP_KA9Q615, rho, h)) & 1u)
                    out[(size_t)i * p->row_bytes + (n >> 3)] |= (unsigned char)(1u << (n & 7));
            }
        }
        return 0;
    }
    if (p->variant == VHIP_VARIANT_REGS) {
        // [group][row][word][lane] -> natural bitmap: new state n of row r sits at position rotr^((r+1) mod NB)(n)
        const vh::RegsLayout &L = p->lay;
        const int NB = p->K - 1, lanes = 1 << L.lb;
        const size_t rowstride = (size_t)L.dw * 64 * L.wbytes;
        const size_t g = frame / L.fpw, fl = frame % L.fpw;
        std::vector<unsigned char> raw((size_t)nrows * rowstride);
        HIP_TRY(hipMemcpy(raw.data(), p->d_dec + (g * p->cap_rows + row0) * rowstride, raw.size(), hipMemcpyDeviceToHost));
        memset(out, 0, (size_t)nrows * p->row_bytes);
        for (int i = 0; i < nrows; i++) {
            const int rot = (row0 + i + 1) % NB;
            for (unsigned n = 0; n < p->N; n++) {
                const unsigned pos = rot == 0 ? n : (((n >> rot) | (n << (NB - rot))) & (p->N - 1));
                const unsigned lam = pos & (lanes - 1), h = (pos >> L.lb) & 1u, rho = pos >> (L.lb + 1);
                const unsigned w = rho / L.nrw, bit = (rho % L.nrw) + L.nrw * h;
                const unsigned char *wp = raw.data() + i * rowstride + ((size_t)w * 64 + fl * lanes + lam) * L.wbytes;
                unsigned word = wp[0] | (wp[1] << 8);
                if (L.wbytes == 4) word |= (wp[2] << 16) | ((unsigned)wp[3] << 24);
                if ((word >> bit) & 1u) out[(size_t)i * p->row_bytes + (n >> 3)] |= (unsigned char)(1u << (n & 7));
            }
        }
        return 0;
    }
    HIP_TRY(hipMemcpy(out, p->d_dec + ((size_t)frame * p->cap_rows + row0) * p->row_bytes, (size_t)nrows * p->row_bytes,
                      hipMemcpyDeviceToHost));
    return 0;
}

int vhip_read_metrics(vhip_decoder *p, int frame, int32_t *out) {
    if (!p) return fail("read_metrics: NULL handle");
    if (use_device(p) != 0) return -1;
    if (frame < 0 || frame >= p->nframes) return fail("read_metrics: out of range");
    if (sync_all(p) != 0) return -1;
    std::vector<int16_t> tmp(p->N);
    const int16_t *src = (p->code == VHIP_KA9Q224)
                             ? p->d_metrics + ((size_t)frame * 2 + p->k24_cur[frame]) * p->N
                             : p->d_metrics + (size_t)frame * p->N;
    HIP_TRY(hipMemcpy(tmp.data(), src, (size_t)p->N * sizeof(int16_t), hipMemcpyDeviceToHost));
    if (k24_multistep(p->variant)) {
        // the multi-step K=24 kernels keep the rotating layout between calls: position q holds state rotl^(pos mod 23)(q)
        const int NB = 23, rot = p->pos % NB;
        for (unsigned q = 0; q < p->N; q++) {
            const unsigned st = rot == 0 ? q : (((q << rot) | (q >> (NB - rot))) & (p->N - 1));
            out[st] = tmp[q];
        }
        return 0;
    }
    for (unsigned i = 0; i < p->N; i++) out[i] = tmp[i];
    return 0;
}

// ---------------------------------------------------------------- synthetic frames
int vhip_noise_q12_from_ebn0(int R, double amp, double ebn0_db) {
    // BPSK amplitude `amp` LSBs, code rate 1/R: Es/N0 = (Eb/N0)/R, sigma^2 = amp^2 / (2 Es/N0)
    const double esn0 = std::pow(10.0, ebn0_db / 10.0) / (double)R;
    const double sigma = amp / std::sqrt(2.0 * esn0);
    return (int)std::llround(sigma * 65536.0 * 4096.0 / VH_FG_SIGMA_C2);
}

int vhip_gen_frames_host(int K, int R, const int *poly, uint64_t seed, uint64_t frame0, int nframes, int payload_bytes,
                         int amp_q16, int noise_q12, unsigned char *payload, unsigned char *syms) {
    if (K < 2 || K > 32 || R < 1 || R > 8 || !poly || nframes < 0 || payload_bytes < 0) return fail("gen_frames: bad arguments");
    vh::gen_frames_host(K, R, poly, seed, frame0, nframes, payload_bytes, amp_q16, noise_q12, payload, syms);
    return 0;
}

int vhip_gen_frames_dev(int K, int R, const int *poly, uint64_t seed, uint64_t frame0, int nframes, int payload_bytes,
                        int amp_q16, int noise_q12, unsigned char *d_payload, unsigned char *d_syms, void *stream) {
    if (K < 2 || K > 32 || R < 1 || R > 8 || !poly || nframes < 0 || payload_bytes < 0) return fail("gen_frames: bad arguments");
    if (nframes == 0) return 0;
    (void)hipGetLastError();  // see use_device
    HIP_TRY(vh::launch_gen_frames(K, R, poly, seed, frame0, nframes, payload_bytes, amp_q16, noise_q12, d_payload, d_syms,
                                  reinterpret_cast<hipStream_t>(stream)));
    return 0;
}

long long vhip_count_bit_errors_dev(const unsigned char *d_a, const unsigned char *d_b, size_t nbytes, void *stream) {
    (void)hipGetLastError();  // see use_device
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    unsigned long long *d_count = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&d_count), sizeof(*d_count)) != hipSuccess) return fail("count_bit_errors: alloc");
    unsigned long long h = 0;
    hipError_t e = hipMemsetAsync(d_count, 0, sizeof(*d_count), s);
    if (e == hipSuccess && nbytes) e = vh::launch_count_bit_errors(d_a, d_b, nbytes, d_count, s);
    if (e == hipSuccess) e = hipMemcpyAsync(&h, d_count, sizeof(h), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d_count);
    if (e != hipSuccess) return fail("count_bit_errors", e);
    return (long long)h;
}

// ---------------------------------------------------------------- reference-shaped five functions per code
#define VHIP_DEFINE_FIVE(T, CODE, CREATE, INIT, UPDATE, CHAINBACK, DELETE)                                            \
    struct T *CREATE(const int *poly, int len) { return reinterpret_cast<struct T *>(vhip_create(CODE, poly, len, 1)); } \
    int INIT(struct T *p, int starting_state) { return vhip_init(reinterpret_cast<vhip_decoder *>(p), starting_state); } \
    void UPDATE(struct T *p, unsigned char *syms, int nbits) {                                                        \
        if (vhip_update(reinterpret_cast<vhip_decoder *>(p), syms, nbits) != 0)                                       \
            fprintf(stderr, "%s\n", vhip_last_error()); /* void in the reference ABI: report loudly */                \
    }                                                                                                                 \
    int CHAINBACK(struct T *p, unsigned char *data, unsigned int nbits, unsigned int endstate) {                      \
        return vhip_chainback(reinterpret_cast<vhip_decoder *>(p), data, nbits, endstate);                            \
    }                                                                                                                 \
    void DELETE(struct T *p) { vhip_delete(reinterpret_cast<vhip_decoder *>(p)); }

VHIP_DEFINE_FIVE(v27_hip, VHIP_KA9Q27, create_viterbi27_hip, init_viterbi27_hip, update_viterbi27_blk_hip,
                 chainback_viterbi27_hip, delete_viterbi27_hip)
VHIP_DEFINE_FIVE(v29_hip, VHIP_KA9Q29, create_viterbi29_hip, init_viterbi29_hip, update_viterbi29_blk_hip,
                 chainback_viterbi29_hip, delete_viterbi29_hip)
VHIP_DEFINE_FIVE(v615_hip, VHIP_KA9Q615, create_viterbi615_hip, init_viterbi615_hip, update_viterbi615_blk_hip,
                 chainback_viterbi615_hip, delete_viterbi615_hip)
VHIP_DEFINE_FIVE(v224_hip, VHIP_KA9Q224, create_viterbi224_hip, init_viterbi224_hip, update_viterbi224_blk_hip,
                 chainback_viterbi224_hip, delete_viterbi224_hip)
VHIP_DEFINE_FIVE(spiral47_hip, VHIP_SPIRAL47, create_spiral47_hip, init_spiral47_hip, update_spiral47_hip,
                 chainback_spiral47_hip, delete_spiral47_hip)
VHIP_DEFINE_FIVE(spiral49_hip, VHIP_SPIRAL49, create_spiral49_hip, init_spiral49_hip, update_spiral49_hip,
                 chainback_spiral49_hip, delete_spiral49_hip)
VHIP_DEFINE_FIVE(spiral27_hip, VHIP_SPIRAL27, create_spiral27_hip, init_spiral27_hip, update_spiral27_hip,
                 chainback_spiral27_hip, delete_spiral27_hip)
VHIP_DEFINE_FIVE(spiral29_hip, VHIP_SPIRAL29, create_spiral29_hip, init_spiral29_hip, update_spiral29_hip,
                 chainback_spiral29_hip, delete_spiral29_hip)
VHIP_DEFINE_FIVE(spiral615_hip, VHIP_SPIRAL615, create_spiral615_hip, init_spiral615_hip, update_spiral615_hip,
                 chainback_spiral615_hip, delete_spiral615_hip)

}  // extern "C"
