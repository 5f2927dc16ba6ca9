/*
 * viterbi_oracle.c -- CPU ORACLE (test infrastructure, NOT a product path).  See viterbi_oracle.h.
 *
 * One scalar butterfly loop parameterised by an arithmetic "family" table.  Notation (SURVEY.md
 * App. A): N = 2^(K-1) states, H = N/2, butterfly j in [0,H) reads old[j], old[j+H] and writes
 * new[2j], new[2j+1]; decision bit 1 = the survivor came from the upper predecessor j+H.
 *
 * Parity status: PINNED against the compiled reference (oracle/_ref, tests/test_oracle_vs_reference.py)
 * and tests/golden/.
 */
#include "viterbi_oracle.h"

#include <limits.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

enum { M_U8MOD = 0, M_U8SAT = 1, M_I16SAT = 2 };
enum { BM_KA9Q_AVG = 0, BM_KA9Q_SUM = 1, BM_SPIRAL_SUM6 = 2, BM_SPIRAL_AVG = 3 };
enum { RN_NONE = 0, RN_KA9Q_WRAP = 1, RN_SPIRAL_SAT = 2 };
enum { CB_GENERIC = 0, CB_K224 = 1 };

typedef struct {
    int K, R;
    int metric;      /* metric arithmetic */
    int bm;          /* branch metric formula */
    int bm_comp;     /* t' = bm_comp - t (u8 saturating at 0 for spiral) */
    int tie_upper;   /* decision on equal candidates: 1 = upper predecessor wins */
    int init_all;    /* initial metric of every state */
    int init_start;  /* initial metric of the start state */
    int renorm;      /* renormalisation rule */
    int renorm_thr;  /* threshold on new[0] */
    int chain;       /* chainback flavour */
    int incremental; /* 1: update appends rows (ka9q); 0: update restarts at row 0 and runs an even
                        number of steps (spiral) */
    int ret_metric;  /* chainback returns the end-state path metric (ka9q615) */
} family_t;

/* One row per enum vo_code.  Sources:
 *  ka9q27  viterbi27_sse2.cpp:46-52 (init), :137-158 (butterfly)
 *  ka9q29  viterbi29_sse2.cpp:36-42, :122-150
 *  ka9q615 viterbi615_sse2.cpp:33-39, :132-148, :160-183
 *  ka9q224 viterbi224_sse2.cpp:39-45, :159-194, :226-246
 *  spiral47 spiral47.cpp:54-61, :164-227, :313-331   spiral49 spiral49.cpp (threshold :790,:1490)
 *  spiral27 spiral27.cpp:164-173,:236  spiral29 spiral29.cpp:507  spiral615 spiral615.cpp:149-269,:31-40
 */
static const family_t FAMILIES[VO_NUM_CODES] = {
    /* K   R  metric    bm              comp  tie init_all        init_start renorm        thr    chain       inc ret */
    {7, 2, M_U8MOD, BM_KA9Q_AVG, 15, 0, 63, 0, RN_NONE, 0, CB_GENERIC, 1, 0},
    {9, 2, M_U8MOD, BM_KA9Q_AVG, 15, 0, 63, 0, RN_NONE, 0, CB_GENERIC, 1, 0},
    {15, 6, M_I16SAT, BM_KA9Q_SUM, 1530, 1, SHRT_MIN + 1000, SHRT_MIN, RN_KA9Q_WRAP, SHRT_MAX - 12750, CB_GENERIC, 1, 1},
    {24, 2, M_I16SAT, BM_KA9Q_SUM, 510, 0, SHRT_MIN + 5000, SHRT_MIN, RN_KA9Q_WRAP, 25000, CB_K224, 1, 0},
    {7, 4, M_U8SAT, BM_SPIRAL_SUM6, 63, 1, 63, 0, RN_SPIRAL_SAT, 126, CB_GENERIC, 0, 0},
    {9, 4, M_U8SAT, BM_SPIRAL_SUM6, 63, 1, 63, 0, RN_SPIRAL_SAT, 103, CB_GENERIC, 0, 0},
    {7, 2, M_U8SAT, BM_SPIRAL_AVG, 63, 1, 63, 0, RN_SPIRAL_SAT, 210, CB_GENERIC, 0, 0},
    {9, 2, M_U8SAT, BM_SPIRAL_AVG, 63, 1, 63, 0, RN_SPIRAL_SAT, 210, CB_GENERIC, 0, 0},
    {15, 6, M_U8SAT, BM_SPIRAL_SUM6, 94, 1, 63, 0, RN_SPIRAL_SAT, 74, CB_GENERIC, 0, 0},
};

struct vo_decoder {
    int code;
    family_t f;
    int N, H;
    int len;            /* trellis steps the caller announced (incl. tail) */
    int cap_rows;       /* rows allocated = len + K-1 */
    int pos;            /* rows written so far (ka9q "dp") */
    int renorms;
    size_t row_bytes;
    int poly[8];
    uint8_t *bt;        /* bt[j] bit r = parity((2j) & poly[r])  (viterbi27_sse2.cpp:64-67) */
    int32_t *old_m, *new_m;
    unsigned char *rows;
};

int vo_code_K(int code) { return (code >= 0 && code < VO_NUM_CODES) ? FAMILIES[code].K : -1; }
int vo_code_R(int code) { return (code >= 0 && code < VO_NUM_CODES) ? FAMILIES[code].R : -1; }

/* src/parity.h:46-55: fold to 8 bits, table lookup == parity of all bits */
static inline int parity32(uint32_t x) {
    x ^= x >> 16;
    x ^= x >> 8;
    x ^= x >> 4;
    x ^= x >> 2;
    x ^= x >> 1;
    return (int)(x & 1u);
}

/* create_viterbi27_sse2 viterbi27_sse2.cpp:57-75 (and 29:47-66, 615:44-62, 224:50-76,
 * spiral47.cpp:64-82).  Unlike the reference the branch table is per handle, not a process global. */
vo_decoder *vo_create(int code, const int *poly, int len) {
    if (code < 0 || code >= VO_NUM_CODES || len < 0) return NULL;
    vo_decoder *p = (vo_decoder *)calloc(1, sizeof(*p));
    if (!p) return NULL;
    p->code = code;
    p->f = FAMILIES[code];
    p->N = 1 << (p->f.K - 1);
    p->H = p->N / 2;
    p->len = len;
    p->cap_rows = len + p->f.K - 1;
    p->row_bytes = (size_t)p->N / 8;
    for (int r = 0; r < p->f.R; r++) p->poly[r] = poly[r];
    p->bt = (uint8_t *)malloc((size_t)p->H);
    p->old_m = (int32_t *)malloc(sizeof(int32_t) * (size_t)p->N);
    p->new_m = (int32_t *)malloc(sizeof(int32_t) * (size_t)p->N);
    p->rows = (unsigned char *)calloc((size_t)p->cap_rows, p->row_bytes);
    if (!p->bt || !p->old_m || !p->new_m || !p->rows) {
        vo_delete(p);
        return NULL;
    }
    for (int j = 0; j < p->H; j++) {
        uint8_t c = 0;
        for (int r = 0; r < p->f.R; r++) {
            /* spiral47.cpp:71: (poly<0) ^ parity((2*state) & abs(poly)) */
            int pr = poly[r];
            int neg = pr < 0;
            uint32_t ap = (uint32_t)(neg ? -pr : pr);
            if (neg ^ parity32((2u * (uint32_t)j) & ap)) c |= (uint8_t)(1u << r);
        }
        p->bt[j] = c;
    }
    vo_init(p, 0);
    return p;
}

/* init_viterbi27_sse2 viterbi27_sse2.cpp:42-54; 29:32-44; 615:29-41; 224:32-47; spiral47.cpp:54-61 */
int vo_init(vo_decoder *p, int starting_state) {
    if (!p) return -1;
    for (int i = 0; i < p->N; i++) p->old_m[i] = p->f.init_all;
    p->old_m[starting_state & (p->N - 1)] = p->f.init_start;
    p->pos = 0;
    p->renorms = 0;
    return 0;
}

void vo_delete(vo_decoder *p) {
    if (!p) return;
    free(p->bt);
    free(p->old_m);
    free(p->new_m);
    free(p->rows);
    free(p);
}

static inline int sat_u8(int x) { return x > 255 ? 255 : (x < 0 ? 0 : x); }
static inline int sat_i16(int x) { return x > SHRT_MAX ? SHRT_MAX : (x < SHRT_MIN ? SHRT_MIN : x); }

/* Branch metric t for butterfly class c (bit r of c = branch-table bit of polynomial r). */
static inline int branch_metric(const family_t *f, const unsigned char *s, unsigned c) {
    int R = f->R;
    switch (f->bm) {
    case BM_KA9Q_AVG: { /* viterbi27_sse2.cpp:137-146: avg_epu8 of the two xors, >>4, &15 */
        int a0 = s[0] ^ ((c & 1u) ? 255 : 0), a1 = s[1] ^ ((c & 2u) ? 255 : 0);
        return (((a0 + a1 + 1) >> 1) >> 4) & 15;
    }
    case BM_KA9Q_SUM: { /* viterbi615_sse2.cpp:132-136, viterbi224_sse2.cpp:159: plain 16-bit sum */
        int t = 0;
        for (int r = 0; r < R; r++) t += s[r] ^ (((c >> r) & 1u) ? 255 : 0);
        return t;
    }
    case BM_SPIRAL_SUM6: { /* spiral47.cpp:164-219, spiral615.cpp:149-242: 6-bit terms, adds_epu8 chain */
        int q = 0;
        for (int r = 0; r < R; r++) q = sat_u8(q + (((s[r] ^ (((c >> r) & 1u) ? 255 : 0)) >> 2) & 63));
        return (q >> 2) & 63;
    }
    default: { /* BM_SPIRAL_AVG spiral27.cpp:164-169: avg_epu8, srli 2, &63 */
        int a0 = s[0] ^ ((c & 1u) ? 255 : 0), a1 = s[1] ^ ((c & 2u) ? 255 : 0);
        return (((a0 + a1 + 1) >> 1) >> 2) & 63;
    }
    }
}

/* One trellis step: butterflies + optional renormalisation.  Writes row `row`. */
static void acs_step(vo_decoder *p, const unsigned char *s, unsigned char *row) {
    const family_t *f = &p->f;
    const int H = p->H, N = p->N;
    const int32_t *old = p->old_m;
    int32_t *nw = p->new_m;
    int tcache[64], tvalid[64];
    memset(tvalid, 0, sizeof(tvalid));
    memset(row, 0, p->row_bytes);
    for (int j = 0; j < H; j++) {
        unsigned c = p->bt[j];
        if (!tvalid[c]) {
            tcache[c] = branch_metric(f, s, c);
            tvalid[c] = 1;
        }
        int t = tcache[c];
        int tc = f->bm_comp - t; /* m_metric; spiral: subs_epu8(const, t) */
        if (f->metric == M_U8SAT && tc < 0) tc = 0;
        int m0, m1, m2, m3, d0, d1, s0, s1;
        if (f->metric == M_U8MOD) {
            /* viterbi27_sse2.cpp:149-158: add_epi8 (wrap), cmpgt_epi8(sub_epi8(m0,m1),0) */
            m0 = (old[j] + t) & 255;
            m1 = (old[j + H] + tc) & 255;
            m2 = (old[j] + tc) & 255;
            m3 = (old[j + H] + t) & 255;
            d0 = (int8_t)(uint8_t)(m0 - m1) > 0;
            d1 = (int8_t)(uint8_t)(m2 - m3) > 0;
            s0 = d0 ? m1 : m0;
            s1 = d1 ? m3 : m2;
        } else if (f->metric == M_U8SAT) {
            /* spiral47.cpp:220-227: adds_epu8, min_epu8(m1,m0), cmpeq(min,m1) -> tie picks upper */
            m0 = sat_u8(old[j] + t);
            m1 = sat_u8(old[j + H] + tc);
            m2 = sat_u8(old[j] + tc);
            m3 = sat_u8(old[j + H] + t);
            s0 = m1 < m0 ? m1 : m0;
            s1 = m3 < m2 ? m3 : m2;
            d0 = (s0 == m1);
            d1 = (s1 == m3);
        } else {
            /* viterbi615_sse2.cpp:139-148 (min + cmpeq, tie -> upper);
             * viterbi224_sse2.cpp:163-194 (cmpgt then min, tie -> lower) */
            m0 = sat_i16(old[j] + t);
            m1 = sat_i16(old[j + H] + tc);
            m2 = sat_i16(old[j] + tc);
            m3 = sat_i16(old[j + H] + t);
            s0 = m1 < m0 ? m1 : m0;
            s1 = m3 < m2 ? m3 : m2;
            if (f->tie_upper) {
                d0 = (s0 == m1);
                d1 = (s1 == m3);
            } else {
                d0 = (m0 > m1);
                d1 = (m2 > m3);
            }
        }
        nw[2 * j] = s0;
        nw[2 * j + 1] = s1;
        /* viterbi27_sse2.cpp:161-162 / viterbi615_sse2.cpp:151: bit n of the row = new state n */
        row[(2 * j) >> 3] |= (unsigned char)(d0 << ((2 * j) & 7));
        row[(2 * j + 1) >> 3] |= (unsigned char)(d1 << ((2 * j + 1) & 7));
    }
    if (f->renorm == RN_KA9Q_WRAP) {
        /* viterbi615_sse2.cpp:160-183, viterbi224_sse2.cpp:226-246: if new[0] >= thr subtract
         * (min - SHRT_MIN) with 16-bit wrap-around */
        if (nw[0] >= f->renorm_thr) {
            int mn = nw[0];
            for (int i = 1; i < N; i++)
                if (nw[i] < mn) mn = nw[i];
            int adjust = mn - SHRT_MIN;
            for (int i = 0; i < N; i++) nw[i] = (int16_t)(uint16_t)(nw[i] - adjust);
            p->renorms++;
        }
    } else if (f->renorm == RN_SPIRAL_SAT) {
        /* spiral47.cpp:313-331 (subs_epu8 of the minimum); spiral615.cpp:31-40 (scalar, plain minus) */
        if (nw[0] > f->renorm_thr) {
            int mn = nw[0];
            for (int i = 1; i < N; i++)
                if (nw[i] < mn) mn = nw[i];
            for (int i = 0; i < N; i++) nw[i] = sat_u8(nw[i] - mn);
            p->renorms++;
        }
    }
    /* swap old/new (viterbi27_sse2.cpp:169-172) */
    int32_t *tmp = p->old_m;
    p->old_m = p->new_m;
    p->new_m = tmp;
}

/* update_viterbi27_blk_sse2 viterbi27_sse2.cpp:119-175 (and 29:108-164, 615:104-191, 224:135-258):
 * incremental, rows appended at dp.  update_spiral47 spiral47.cpp:536-538: always starts at row 0
 * and runs nbits/2 double-steps (an odd last step is dropped). */
void vo_update_blk(vo_decoder *p, const unsigned char *syms, int nbits) {
    if (!p || nbits <= 0) return;
    int steps = nbits;
    if (!p->f.incremental) {
        p->pos = 0;
        steps = (nbits / 2) * 2;
    }
    for (int i = 0; i < steps; i++) {
        if (p->pos >= p->cap_rows) break; /* the reference would overrun its malloc here */
        acs_step(p, syms + (size_t)i * (size_t)p->f.R, p->rows + (size_t)p->pos * p->row_bytes);
        p->pos++;
    }
}

/* chainback_viterbi27_sse2 viterbi27_sse2.cpp:78-105; 29:69-94; 615:65-91 (32-bit word semantics,
 * SURVEY.md §0.3); spiral47.cpp:84-121; chainback_viterbi224_sse2 viterbi224_sse2.cpp:79-121. */
int vo_chainback(vo_decoder *p, unsigned char *data, unsigned int nbits, unsigned int endstate) {
    if (!p) return -1;
    const int K = p->f.K;
    const unsigned N = (unsigned)p->N;
    int ret = 0;
    if (p->f.chain == CB_K224) {
        /* no tail skip; emits the bit falling off the right end; stores only when nbits%8==0 */
        unsigned e = endstate & (N - 1);
        unsigned char dbyte = 0;
        while (nbits-- > 0) {
            dbyte = (unsigned char)(((e & 1u) << 7) | (dbyte >> 1));
            if ((nbits & 7u) == 0) data[nbits >> 3] = dbyte;
            unsigned bit = 0;
            if ((int)nbits < p->pos) { /* unwritten rows read as 0 (reference: uninitialised malloc) */
                const unsigned char *row = p->rows + (size_t)nbits * p->row_bytes;
                bit = (row[e >> 3] >> (e & 7u)) & 1u;
            }
            e = (bit << (K - 2)) | (e >> 1);
        }
        return 0;
    }
    const int add = (K - 1 < 8) ? 8 - (K - 1) : 0;
    const int sub = (K - 1 > 8) ? (K - 1) - 8 : 0;
    unsigned e = (endstate % N) << add;
    if (p->f.ret_metric) ret = (int)(int16_t)p->old_m[endstate % N]; /* viterbi615_sse2.cpp:76 */
    while (nbits-- != 0) {
        unsigned st = e >> add;
        unsigned k = 0;
        size_t r = (size_t)nbits + (size_t)(K - 1); /* "d += K-1: look past tail" */
        if (r < (size_t)p->pos) { /* unwritten rows read as 0 (reference: uninitialised malloc) */
            const unsigned char *row = p->rows + r * p->row_bytes;
            k = (row[st >> 3] >> (st & 7u)) & 1u;
        }
        e = (e >> 1) | (k << (K - 2 + add));
        data[nbits >> 3] = (unsigned char)(e >> sub);
    }
    return ret;
}

/* Sliding-window traceback over the rows the update produced.  NOT a reference function: the
 * ka9q / spiral decoders only have the whole-frame chainback above, and the library whose decoders take a traceback
 * length (williamyang98/ViterbiDecoderCpp, core->set_traceback_length, src/main.cpp:169) is an un-vendored submodule --
 * PARITY UNPINNED.  The semantics are defined here and the fused GPU decode (vhip_decode_windowed_dev) is tested against
 * them bit for bit; with depth >= the frame length they reduce to vo_chainback(nbits, endstate 0).
 *
 * Payload bit i is the decision read at row i+K-1 (the generic walk above).  The payload is cut into blocks of `block`
 * bits; block b = bits [lo, hi) is decoded by a walk that starts in state 0 at row min(hi+K-1+depth, rows)-1 -- `depth`
 * steps ahead of the block, where the survivors have merged with high probability, or at the frame's last row, where
 * the tail forces state 0 -- and keeps the decisions of the rows [lo+K-1, hi+K-1).  Bits are packed MSB-first; the unused
 * low bits of a last partial byte are 0. */
int vo_chainback_windowed(vo_decoder *p, unsigned char *data, unsigned int nbits, unsigned int depth, unsigned int block) {
    if (!p || block == 0 || p->f.chain != CB_GENERIC) return -1;
    const int K = p->f.K;
    memset(data, 0, (nbits + 7) / 8);
    for (unsigned lo = 0; lo < nbits; lo += block) {
        const unsigned hi = lo + block < nbits ? lo + block : nbits;
        long top = (long)hi + (K - 1) + (long)depth;
        if (top > p->pos) top = p->pos;
        if (top < (long)hi + (K - 1)) top = (long)hi + (K - 1); /* rows never written read as 0, as in vo_chainback */
        unsigned st = 0;
        for (long r = top - 1; r >= (long)lo + (K - 1); r--) {
            unsigned k = 0;
            if (r < p->pos) k = (p->rows[(size_t)r * p->row_bytes + (st >> 3)] >> (st & 7u)) & 1u;
            st = (st >> 1) | (k << (K - 2));
            const long i = r - (K - 1);
            if (i < (long)hi && k) data[i >> 3] |= (unsigned char)(0x80u >> (i & 7));
        }
    }
    return 0;
}

const unsigned char *vo_decision_rows(const vo_decoder *p) { return p->rows; }
size_t vo_row_bytes(const vo_decoder *p) { return p->row_bytes; }
int vo_rows_written(const vo_decoder *p) { return p->pos; }
int vo_num_states(const vo_decoder *p) { return p->N; }
int vo_renorm_count(const vo_decoder *p) { return p->renorms; }
void vo_get_metrics(const vo_decoder *p, int32_t *out) { memcpy(out, p->old_m, sizeof(int32_t) * (size_t)p->N); }

/* Encoder convention recovered from the decoders: sr = (sr<<1)|bit, coded bit r
 * = parity(sr & poly[r]) over the low K bits, payload bytes MSB-first, K-1 zero tail bits. */
size_t vo_encode(int K, int R, const int *poly, const unsigned char *payload, size_t nbytes,
                 unsigned char *coded_bits) {
    uint32_t sr = 0;
    const uint32_t kmask = (K >= 32) ? 0xffffffffu : ((1u << K) - 1u);
    size_t o = 0;
    size_t total = nbytes * 8 + (size_t)(K - 1);
    for (size_t i = 0; i < total; i++) {
        unsigned bit = 0;
        if (i < nbytes * 8) bit = (payload[i >> 3] >> (7 - (i & 7))) & 1u;
        sr = ((sr << 1) | bit) & kmask;
        for (int r = 0; r < R; r++) {
            int pr = poly[r];
            uint32_t ap = (uint32_t)(pr < 0 ? -pr : pr);
            coded_bits[o++] = (unsigned char)((pr < 0) ^ parity32(sr & ap));
        }
    }
    return o;
}

/* Timed decode loop for bench.py's cpu_baseline leg (kind "port"): reset + update + chainback per frame as
 * src/main.cpp:257-280, cycling over nsample frames until `seconds` have elapsed. */
long vo_bench_loop(vo_decoder *p, const unsigned char *syms, int nsample, long frame_stride, int steps, unsigned nbits,
                   double seconds, double *elapsed) {
    unsigned char *out = (unsigned char *)malloc((nbits + 7) / 8 + 8);
    struct timespec t0, t1;
    long n = 0;
    double el = 0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (;;) {
        vo_init(p, 0);
        vo_update_blk(p, syms + (size_t)(n % nsample) * (size_t)frame_stride, steps);
        vo_chainback(p, out, nbits, 0);
        n++;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        el = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
        if (el >= seconds) break;
    }
    free(out);
    *elapsed = el;
    return n;
}

/* Decodes nframes frames one after the other with one decoder (reset + update + chainback per frame) and keeps every frame's
 * bytes: what the whole-batch parity tests compare a full GPU batch with where the compiled reference is absent. */
void vo_decode_batch(vo_decoder *p, const unsigned char *syms, long nframes, long frame_stride, int steps, unsigned nbits,
                     unsigned char *out, long out_stride) {
    for (long f = 0; f < nframes; f++) {
        vo_init(p, 0);
        vo_update_blk(p, syms + (size_t)f * (size_t)frame_stride, steps);
        memset(out + (size_t)f * (size_t)out_stride, 0, (size_t)out_stride);
        vo_chainback(p, out + (size_t)f * (size_t)out_stride, nbits, 0);
    }
}

