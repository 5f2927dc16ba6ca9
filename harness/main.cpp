// harness/main.cpp -- stand-alone benchmark harness for the HIP backend, shaped after the reference's
// src/main.cpp: the same three flags (-t/--sampling-time, -n/--minimum-samples, -o/--output; main.cpp:300-316), one
// block per (K,R) code with the reference's polynomials (main.cpp:363-419), the same time-boxed sampling loop around
// reset / update / chainback (main.cpp:257-280) and the same JSON schema (main.cpp:80-118), so the reference's own
// scripts/tabulate_data.py can read the file.  Additions: a batch dimension (--frames), AWGN input (--ebn0, --hard),
// --codes, --seed, and per-entry "frames" / "hbm_bytes_per_update" / "roofline_fraction" keys.
//
// The reference harness cannot be built (its `viterbi` submodule is not vendored), hence this one.
// Timing is host wall clock around blocking calls on device-resident buffers (inputs already in HBM), i.e. what the
// reference's Timer measures for its CPU decoders.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../include/hip_interface.h"

namespace {

struct Args {
    float sampling_time = 1.0f;
    size_t minimum_samples = 8;
    std::string output = "./data/benchmark_hip.json";
    int frames = 0;  // 0 = per-code default
    std::string codes = "27,47,29,49,615,224";
    double ebn0 = 1e9;  // 1e9 = per-code default
    bool hard = false;
    unsigned long long seed = 0x5EED;
    int payload_bytes = 0;  // 0 = the reference's size for the code
    int gpus = 1;           // frame-range shards, one per device
    bool host_api = false;  // the reference's own methodology: one frame per handle, host pointers, the five blocking functions
};

struct CodeBlock {
    const char *name;
    int code, K, R;
    int poly[6];
    int ref_payload_bytes;  // src/main.cpp:366,376,385,395,404,414
    int default_frames;
    double default_ebn0;
};
const CodeBlock BLOCKS[] = {
    {"27", VHIP_KA9Q27, 7, 2, {0x6d, 0x4f}, 1024, 65536, 4.0},
    {"47", VHIP_SPIRAL47, 7, 4, {121, 117, 91, 111}, 1024, 65536, 2.0},
    {"29", VHIP_KA9Q29, 9, 2, {0x1af, 0x11d}, 512, 32768, 4.0},
    {"49", VHIP_SPIRAL49, 9, 4, {501, 441, 331, 315}, 512, 32768, 2.0},
    {"615", VHIP_KA9Q615, 15, 6, {042631, 047245, 056507, 073363, 077267, 064537}, 256, 4096, 1.0},
    {"224", VHIP_KA9Q224, 24, 2, {062650457, 062650455}, 8, 1, 4.0},
};

void usage(const char *argv0) {
    fprintf(stderr,
            "usage: %s [-t seconds] [-n samples] [-o file.json] [--frames N] [--codes 27,47,...] [--ebn0 dB | --hard]\n"
            "          [--payload-bytes B] [--seed S] [--gpus N] [--host-api]\n"
            "  --host-api: the reference's methodology (src/main.cpp:257-280) -- ONE frame per handle, host buffers, the five\n"
            "              blocking reference-shaped functions (create_viterbi27_hip ...): PCIe copies and launch latency included\n",
            argv0);
}

bool parse(int argc, char **argv, Args &a) {
    for (int i = 1; i < argc; i++) {
        std::string k = argv[i];
        auto val = [&]() -> const char * { return i + 1 < argc ? argv[++i] : nullptr; };
        const char *v = nullptr;
        if (k == "-t" || k == "--sampling-time") { if (!(v = val())) return false; a.sampling_time = (float)atof(v); }
        else if (k == "-n" || k == "--minimum-samples") { if (!(v = val())) return false; a.minimum_samples = (size_t)atoll(v); }
        else if (k == "-o" || k == "--output") { if (!(v = val())) return false; a.output = v; }
        else if (k == "--frames") { if (!(v = val())) return false; a.frames = atoi(v); }
        else if (k == "--codes") { if (!(v = val())) return false; a.codes = v; }
        else if (k == "--ebn0") { if (!(v = val())) return false; a.ebn0 = atof(v); }
        else if (k == "--hard") a.hard = true;
        else if (k == "--seed") { if (!(v = val())) return false; a.seed = strtoull(v, nullptr, 0); }
        else if (k == "--payload-bytes") { if (!(v = val())) return false; a.payload_bytes = atoi(v); }
        else if (k == "--gpus") { if (!(v = val())) return false; a.gpus = atoi(v); }
        else if (k == "--host-api") a.host_api = true;
        else if (k == "-h" || k == "--help") return false;
        else { fprintf(stderr, "unknown argument %s\n", k.c_str()); return false; }
    }
    // same validation as src/main.cpp:344-351
    if (a.sampling_time <= 0.0f) { fprintf(stderr, "Sampling time must be positive (%.3f)\n", a.sampling_time); return false; }
    if (a.minimum_samples == 0) { fprintf(stderr, "Minimum number of samples must be non-zero\n"); return false; }
    return true;
}

bool selected(const std::string &list, const char *name) {
    size_t pos = 0;
    while (pos <= list.size()) {
        size_t e = list.find(',', pos);
        if (e == std::string::npos) e = list.size();
        if (list.substr(pos, e - pos) == name) return true;
        pos = e + 1;
    }
    return false;
}

using clk = std::chrono::steady_clock;
uint64_t ns_since(clk::time_point t0) { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(clk::now() - t0).count(); }

#define HIPCHK(e)                                                                        \
    do {                                                                                 \
        hipError_t _e = (e);                                                             \
        if (_e != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(_e)); exit(2); } \
    } while (0)

void print_u64_array(FILE *fp, const std::vector<uint64_t> &v) {
    fprintf(fp, "[");
    for (size_t i = 0; i < v.size(); i++) fprintf(fp, "%s%llu", i ? "," : "", (unsigned long long)v[i]);
    fprintf(fp, "]");
}

}  // namespace

int main(int argc, char **argv) {
    Args args;
    if (!parse(argc, argv, args)) { usage(argv[0]); return 1; }
    if (vhip_device_count() < 1) { fprintf(stderr, "no HIP device: this harness has no CPU decode path\n"); return 2; }
    if (args.gpus < 1 || args.gpus > vhip_device_count()) { fprintf(stderr, "--gpus %d but %d device(s) visible\n", args.gpus, vhip_device_count()); return 1; }
    FILE *fp = fopen(args.output.c_str(), "w");
    if (!fp) { fprintf(stderr, "Failed to open file for writing: '%s'\n", args.output.c_str()); return 1; }
    fprintf(fp, "[\n");
    bool first = true;
    for (const CodeBlock &cb : BLOCKS) {
        if (!selected(args.codes, cb.name)) continue;
        int frames = args.frames > 0 ? args.frames : cb.default_frames;  // per GPU until the shards are built
        const int bytes = args.payload_bytes > 0 ? args.payload_bytes : cb.ref_payload_bytes;
        const size_t decode_bits = (size_t)bytes * 8, transmit_bits = decode_bits + cb.K - 1, symbols = transmit_bits * cb.R;
        fprintf(stderr, "[test_run]\nK=%d, R=%d\ntotal_input_bytes = %d, frames = %d\n", cb.K, cb.R, bytes, frames);
        if (args.host_api) {
            // What a maintainer gets after INTEGRATION.md §2a: test_third_party's loop (src/main.cpp:257-280) over the
            // reference-shaped entry points -- one frame, host pointers, every call blocking (H2D / D2H copies inside).
            frames = 1;
            std::vector<unsigned char> payload((size_t)bytes), syms(symbols), out((size_t)(transmit_bits + 7) / 8);
            const int amp = args.hard ? (int)(127.5 * 65536) : 64 * 65536;
            const int nqh = args.hard ? 0 : vhip_noise_q12_from_ebn0(cb.R, 64.0, args.ebn0 > 1e8 ? cb.default_ebn0 : args.ebn0);
            if (vhip_gen_frames_host(cb.K, cb.R, cb.poly, args.seed, 0, 1, bytes, amp, nqh, payload.data(), syms.data()) != 0) return 2;
            vhip_decoder *dec = vhip_create(cb.code, cb.poly, (int)transmit_bits, 1);
            if (!dec) { fprintf(stderr, "%s\n", vhip_last_error()); return 2; }
            std::vector<uint64_t> init_ns, update_ns, chainback_ns;
            const auto t_total = clk::now();
            for (size_t i = 0;; i++) {
                const float elapsed = (float)ns_since(t_total) * 1e-9f;
                if (elapsed > args.sampling_time && i > args.minimum_samples) break;
                memset(out.data(), 0, out.size());  // main.cpp:262
                auto t = clk::now();
                vhip_init(dec, 0);
                vhip_sync(dec);
                init_ns.push_back(ns_since(t));
                t = clk::now();
                if (vhip_update(dec, syms.data(), (int)transmit_bits) != 0) { fprintf(stderr, "%s\n", vhip_last_error()); return 2; }
                update_ns.push_back(ns_since(t));
                t = clk::now();
                (void)vhip_chainback(dec, out.data(), (unsigned)decode_bits, 0);
                if (vhip_status(dec) != 0) { fprintf(stderr, "%s\n", vhip_last_error()); return 2; }
                chainback_ns.push_back(ns_since(t));
            }
            if (cb.K == 24) (void)vhip_chainback(dec, out.data(), (unsigned)transmit_bits, 0);  // BER from the nbits+K-1 call
            long long errors = 0;
            for (int b = 0; b < bytes; b++) errors += __builtin_popcount((unsigned)(out[(size_t)b] ^ payload[(size_t)b]));
            vhip_delete(dec);
            double upd_mean = 0;
            for (uint64_t v : update_ns) upd_mean += (double)v;
            upd_mean /= (double)update_ns.size();
            fprintf(fp, "%s{\n", first ? "" : ",\n");
            first = false;
            fprintf(fp, "  \"name\": \"hip_host_1frame\",\n  \"K\": %d,\n  \"R\": %d,\n  \"poly\": [", cb.K, cb.R);
            for (int r = 0; r < cb.R; r++) fprintf(fp, "%s%d", r ? "," : "", cb.poly[r]);
            fprintf(fp, "],\n  \"frames\": 1,\n  \"gpus\": 1,\n  \"frames_per_gpu\": 1,\n");
            fprintf(fp, "  \"total_input_bytes\": %d,\n  \"total_transmit_bits\": %zu,\n  \"total_output_symbols\": %zu,\n", bytes, transmit_bits, symbols);
            fprintf(fp, "  \"sampling_time\": %f,\n  \"minimum_samples\": %zu,\n  \"total_samples\": %zu,\n", args.sampling_time, args.minimum_samples, update_ns.size());
            fprintf(fp, "  \"init_ns\": ");
            print_u64_array(fp, init_ns);
            fprintf(fp, ",\n  \"update_ns\": ");
            print_u64_array(fp, update_ns);
            fprintf(fp, ",\n  \"chainback_ns\": ");
            print_u64_array(fp, chainback_ns);
            fprintf(fp, ",\n  \"total_bits\": %zu,\n  \"total_bit_errors\": %lld,\n  \"bit_error_rate\": %f\n}", decode_bits, errors, (double)errors / (double)decode_bits);
            fprintf(stderr, "o hip_host_1frame (%.6f)  update %.3f Msym/s (%.1f us per call)\n", (double)errors / (double)decode_bits,
                    (double)symbols / (upd_mean * 1e-9) / 1e6, upd_mean * 1e-3);
            continue;
        }
        // One shard of `frames` frames per GPU (frames are independent -> frame-range sharding, no
        // collective).  All shards are driven from this thread: every call is asynchronous on its handle's stream, so
        // a phase is issued to all devices and then awaited; the phase time is first issue -> last completion.
        struct Shard {
            int device = 0;
            unsigned char *d_payload = nullptr, *d_syms = nullptr, *d_out = nullptr;
            vhip_decoder *dec = nullptr;
        };
        std::vector<Shard> shards((size_t)args.gpus);
        const double ebn0 = args.ebn0 > 1e8 ? cb.default_ebn0 : args.ebn0;
        const int amp_q16 = args.hard ? (int)(127.5 * 65536) : 64 * 65536;
        const int nq = args.hard ? 0 : vhip_noise_q12_from_ebn0(cb.R, 64.0, ebn0);
        for (int d = 0; d < args.gpus; d++) {
            Shard &sh = shards[(size_t)d];
            sh.device = d;
            HIPCHK(hipSetDevice(d));
            HIPCHK(hipMalloc((void **)&sh.d_payload, (size_t)frames * bytes));
            HIPCHK(hipMalloc((void **)&sh.d_syms, (size_t)frames * symbols));
            HIPCHK(hipMalloc((void **)&sh.d_out, (size_t)frames * bytes));
            if (vhip_gen_frames_dev(cb.K, cb.R, cb.poly, args.seed, (uint64_t)d * (uint64_t)frames, frames, bytes, amp_q16, nq,
                                    sh.d_payload, sh.d_syms, nullptr) != 0) {
                fprintf(stderr, "%s\n", vhip_last_error());
                return 2;
            }
            HIPCHK(hipDeviceSynchronize());
            sh.dec = vhip_create(cb.code, cb.poly, (int)transmit_bits, frames);  // bound to the current device
            if (!sh.dec) { fprintf(stderr, "%s\n", vhip_last_error()); return 2; }
        }
        auto sync_all = [&]() { for (Shard &sh : shards) vhip_sync(sh.dec); };
        std::vector<uint64_t> init_ns, update_ns, chainback_ns;
        const auto t_total = clk::now();
        for (size_t i = 0;; i++) {
            const float elapsed = (float)ns_since(t_total) * 1e-9f;
            if (elapsed > args.sampling_time && i > args.minimum_samples) break;
            for (Shard &sh : shards) {
                HIPCHK(hipSetDevice(sh.device));
                HIPCHK(hipMemsetAsync(sh.d_out, 0, (size_t)frames * bytes, nullptr));  // main.cpp:262
                HIPCHK(hipDeviceSynchronize());
            }
            auto t = clk::now();
            for (Shard &sh : shards) vhip_init(sh.dec, 0);
            sync_all();
            init_ns.push_back(ns_since(t));
            t = clk::now();
            if (cb.K == 24 && args.gpus > 1) {
                // the K=24 update blocks its caller (the host steers the speculative renormalisation): one thread per GPU,
                // so that the shards run side by side and update_ns is the slowest shard, not the sum
                std::vector<std::thread> th;
                std::vector<int> rcs((size_t)args.gpus, 0);
                for (int d = 0; d < args.gpus; d++)
                    th.emplace_back([&, d]() { rcs[(size_t)d] = vhip_update_dev(shards[(size_t)d].dec, shards[(size_t)d].d_syms, (int)transmit_bits); });
                for (auto &x : th) x.join();
                for (int rc : rcs)
                    if (rc != 0) { fprintf(stderr, "K=24 shard update failed\n"); return 2; }
            } else {
                for (Shard &sh : shards)
                    if (vhip_update_dev(sh.dec, sh.d_syms, (int)transmit_bits) != 0) { fprintf(stderr, "%s\n", vhip_last_error()); return 2; }
            }
            sync_all();
            update_ns.push_back(ns_since(t));
            t = clk::now();
            // K=24: the reference's own call (nbits = payload bits) is what is timed; it does not decode correctly
            // (SURVEY.md §0.4), so its BER is reported from a separate nbits+K-1 call below.
            for (Shard &sh : shards)
                if (vhip_chainback_dev(sh.dec, sh.d_out, (unsigned)decode_bits, 0) != 0) { fprintf(stderr, "%s\n", vhip_last_error()); return 2; }
            sync_all();
            chainback_ns.push_back(ns_since(t));
        }
        long long errors = 0;
        for (Shard &sh : shards) {
            HIPCHK(hipSetDevice(sh.device));
            if (cb.K == 24) {
                unsigned char *d_long;
                const size_t lb = (transmit_bits + 7) / 8;
                HIPCHK(hipMalloc((void **)&d_long, (size_t)frames * lb));
                vhip_chainback_dev(sh.dec, d_long, (unsigned)transmit_bits, 0);
                vhip_sync(sh.dec);
                for (int f = 0; f < frames; f++)
                    errors += vhip_count_bit_errors_dev(d_long + (size_t)f * lb, sh.d_payload + (size_t)f * bytes, (size_t)bytes, nullptr);
                HIPCHK(hipFree(d_long));
            } else {
                errors += vhip_count_bit_errors_dev(sh.d_out, sh.d_payload, (size_t)frames * bytes, nullptr);
            }
        }
        const int frames_per_gpu = frames;
        frames *= args.gpus;  // whole-job sizes below
        const size_t total_bits = (size_t)frames * decode_bits;
        // bytes the ACS kernels move per update: K <= 15 the algorithmic R + N/8 per frame-step; K=24 by
        // variant -- per-step kernel 34 603 010 B per step, multi-step passes (passes x 32 MiB + 23 x (1 MiB + 2)) per 23 steps
        const int k24_variant = cb.K == 24 ? vhip_get_variant(shards[0].dec) : 0;
        const double k24_step = k24_variant == VHIP_VARIANT_HBM_TILED ? (2 * 33554432.0 + 23 * 1048578.0) / 23.0
                              : k24_variant == VHIP_VARIANT_HBM_FUSED ? (5 * 33554432.0 + 23 * 1048578.0) / 23.0 : 34603010.0;
        const double hbm_bytes = cb.K == 24 ? k24_step * transmit_bits * frames
                                            : (double)(cb.R + (1 << (cb.K - 1)) / 8) * transmit_bits * frames;
        double upd_mean = 0;
        for (uint64_t v : update_ns) upd_mean += (double)v;
        upd_mean /= (double)update_ns.size();
        // reference schema (src/main.cpp:80-118); per-frame sizes as in the reference, batch size in "frames"
        fprintf(fp, "%s{\n", first ? "" : ",\n");
        first = false;
        fprintf(fp, "  \"name\": \"hip\",\n  \"K\": %d,\n  \"R\": %d,\n  \"poly\": [", cb.K, cb.R);
        for (int r = 0; r < cb.R; r++) fprintf(fp, "%s%d", r ? "," : "", cb.poly[r]);
        fprintf(fp, "],\n  \"frames\": %d,\n  \"gpus\": %d,\n  \"frames_per_gpu\": %d,\n", frames, args.gpus, frames_per_gpu);
        fprintf(fp, "  \"total_input_bytes\": %zu,\n  \"total_transmit_bits\": %zu,\n  \"total_output_symbols\": %zu,\n",
                (size_t)bytes * frames, transmit_bits * frames, symbols * frames);
        fprintf(fp, "  \"sampling_time\": %f,\n  \"minimum_samples\": %zu,\n  \"total_samples\": %zu,\n", args.sampling_time,
                args.minimum_samples, update_ns.size());
        fprintf(fp, "  \"init_ns\": ");
        print_u64_array(fp, init_ns);
        fprintf(fp, ",\n  \"update_ns\": ");
        print_u64_array(fp, update_ns);
        fprintf(fp, ",\n  \"chainback_ns\": ");
        print_u64_array(fp, chainback_ns);
        fprintf(fp, ",\n  \"hbm_bytes_per_update\": %.0f,\n  \"roofline_fraction\": %.5f,\n", hbm_bytes, hbm_bytes / (upd_mean * 1e-9) / (8e12 * args.gpus));
        fprintf(fp, "  \"total_bits\": %zu,\n  \"total_bit_errors\": %lld,\n  \"bit_error_rate\": %f\n}", total_bits, errors,
                (double)errors / (double)total_bits);
        fprintf(stderr, "o hip (%.6f)  update %.1f Msym/s\n", (double)errors / (double)total_bits,
                (double)symbols * frames / (upd_mean * 1e-9) / 1e6);
        for (Shard &sh : shards) {
            HIPCHK(hipSetDevice(sh.device));
            vhip_delete(sh.dec);
            HIPCHK(hipFree(sh.d_payload));
            HIPCHK(hipFree(sh.d_syms));
            HIPCHK(hipFree(sh.d_out));
        }
    }
    fprintf(fp, "\n]\n");
    fclose(fp);
    return 0;
}
