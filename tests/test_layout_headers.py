"""The decision-row layouts shared by the ACS kernels, the chainback kernels and the host-side row conversion
(csrc/k15_layout.h, csrc/k24f_layout.h, csrc/k24t_layout.h) must be bijections: every state's decision has exactly one bit of the row.
The headers are plain C++ on the host side, so this compiles a small checker with g++ (no GPU, no HIP)."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ka9q_viterbi_comparison_amd", "csrc")

CHECKER = r"""
#include <cstdio>
#include <vector>
#include "k15_layout.h"
#include "k24f_layout.h"
#include "k24t_layout.h"
int main() {
    // K=15: 16 registers x 2 halves -> 32 distinct bits, both bit orders
    for (int sb = 0; sb < 2; sb++) {
        unsigned seen = 0;
        for (unsigned rho = 0; rho < 16; rho++)
            for (unsigned h = 0; h < 2; h++) {
                const unsigned b = vh::k15_decision_bit(sb != 0, rho, h);
                if (b >= 32 || ((seen >> b) & 1u)) { printf("k15 layout %d: clash at rho %u half %u\n", sb, rho, h); return 1; }
                seen |= 1u << b;
            }
        // the word index uses rho >> 4 only: rho and rho + 16 must agree
        for (unsigned rho = 0; rho < 48; rho++)
            for (unsigned h = 0; h < 2; h++)
                if (vh::k15_decision_bit(sb != 0, rho, h) != vh::k15_decision_bit(sb != 0, rho + 16, h)) { printf("k15 period\n"); return 1; }
    }
    {
        unsigned seen = 0;
        for (unsigned rho = 0; rho < 16; rho++)
            for (unsigned h = 0; h < 2; h++) {
                const unsigned b = vh::k24f_decision_bit(rho, h);
                if (b >= 32 || ((seen >> b) & 1u)) { printf("k24f bit clash\n"); return 1; }
                seen |= 1u << b;
            }
    }
    // K=24: for one phase of every group, position -> (word, bit) covers the 1 MiB row exactly once
    const int phases[5] = {0, 5, 10, 15, 20};
    for (int g = 0; g < 5; g++) {
        if (vh::k24f_group_of_phase(phases[g]) != g) { printf("group of phase\n"); return 1; }
        std::vector<unsigned> row(1u << 18, 0u);
        for (unsigned p = 0; p < (1u << 23); p++) {
            unsigned w, b;
            vh::k24f_locate(p, phases[g], w, b);
            if (w >= row.size() || b >= 32 || ((row[w] >> b) & 1u)) { printf("k24f group %d: clash at position %u\n", g, p); return 1; }
            row[w] |= 1u << b;
        }
        // the thread base of thread u is the position whose vector index and in-vector bits are zero
        const unsigned nthreads = (1u << 23) / ((1u << vh::k24f_lw(g)) * 16u);
        for (unsigned u = 0; u < nthreads; u += 977) {
            unsigned w, b;
            vh::k24f_locate(vh::k24f_thread_base(g, u), phases[g], w, b);
            const unsigned wpt = (16u << vh::k24f_lw(g)) / 32u;
            if (w != u * wpt || b != vh::k24f_decision_bit(0, 0)) { printf("k24f group %d: thread base of %u\n", g, u); return 1; }
        }
    }
    // K=24 two-pass layout: every phase's position -> (word, bit) map covers the row exactly once; thread_base | spos(rho)
    // | half reproduces the position a (tile, thread, register, field) holds; register bits match the paired position bits
    for (int phi = 0; phi < 23; phi++) {
        const int g = vh::k24t_group_of_phase(phi);
        if (phi < vh::k24t_first_phase(g) || phi >= vh::k24t_first_phase(g) + vh::k24t_nphases(g)) { printf("k24t group range\n"); return 1; }
        const int b = 22 - phi;
        if (b >= 1) {
            const int rb = vh::k24t_regbit(g, b);
            if (rb < 0 || (1 << rb) >= vh::k24t_nr(g) || vh::k24t_spos(g, 1u << rb) != (1u << b)) { printf("k24t regbit phase %d\n", phi); return 1; }
        }
        if (phi != vh::k24t_first_phase(g)) continue;  // one full sweep per group is enough for the bijection
        std::vector<unsigned> row(1u << 18, 0u);
        for (unsigned p = 0; p < (1u << 23); p++) {
            unsigned w, bit;
            vh::k24t_locate(p, phi, w, bit);
            if (w >= row.size() || bit >= 32 || ((row[w] >> bit) & 1u)) { printf("k24t group %d: clash at position %u\n", g, p); return 1; }
            row[w] |= 1u << bit;
        }
        const unsigned tiles = g <= vh::K24T_H2 ? 256u : 512u, threads = (unsigned)vh::k24t_threads(g), nr = (unsigned)vh::k24t_nr(g);
        if ((unsigned long long)tiles * threads * nr * 2ull != (1ull << 23)) { printf("k24t group %d: size\n", g); return 1; }
        for (unsigned tile = 0; tile < tiles; tile += 37)
            for (unsigned tid = 0; tid < threads; tid += 13)
                for (unsigned rho = 0; rho < nr; rho += 5)
                    for (unsigned h = 0; h < 2; h++) {
                        const unsigned p = vh::k24t_thread_base(g, tile, tid) | vh::k24t_spos(g, rho) | h;
                        unsigned w, bit;
                        vh::k24t_locate(p, phi, w, bit);
                        if (w != (tile * threads + tid) * (nr / 16) + (rho >> 4) || bit != vh::k24t_decision_bit(rho, h)) {
                            printf("k24t group %d: tile %u tid %u rho %u half %u\n", g, tile, tid, rho, h);
                            return 1;
                        }
                    }
    }
    printf("ok\n");
    return 0;
}
"""


def test_decision_row_layouts_are_bijections():
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "check.cpp")
        exe = os.path.join(d, "check")
        open(src, "w").write(CHECKER)
        subprocess.run(["g++", "-std=c++17", "-O2", "-I", CSRC, src, "-o", exe], check=True)
        out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0 and out.stdout.strip() == "ok", out.stdout + out.stderr
