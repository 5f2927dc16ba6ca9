// jit.hip -- runtime specialisation of the fast ACS kernels for polynomials other than the harness set.
//
// The reference builds its branch table from whatever `poly` the caller passes (ka9q_libfec_port/viterbi27_sse2.cpp:57-75,
// viterbi224_sse2.cpp:68-73) at no speed cost.  The fast kernels here (acs_regs.hip, acs_k15.hip, acs_k24t.hip) need the
// branch-table class of every register at every unrolled phase as a compile-time constant -- a per-register table lookup
// costs more than the add-compare-select it feeds -- so for other polynomials the SAME source file is compiled once more,
// at create time, with those polynomials as its constants (`hipcc --genco`, 2-6 s per kernel family), cached on disk by a
// hash of sources + options, and launched through hipModuleLaunchKernel.  Kernel code, data layouts and chainback kernels
// are identical to the harness-polynomial path.  When the sources or the compiler are not there the handle falls back to
// the any-polynomial kernels (acs_lds.hip / acs_k24.hip); nothing here ever decodes on the CPU.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>

#include "kernels.h"

namespace vh {

namespace {

std::mutex g_mu;
std::map<std::string, hipModule_t> g_modules;  // key: device id + cache file

std::string source_dir() {
    Dl_info info;
    if (!dladdr(reinterpret_cast<void *>(&jit_function), &info) || !info.dli_fname) return "";
    std::string p = info.dli_fname;
    const size_t slash = p.rfind('/');
    return slash == std::string::npos ? "." : p.substr(0, slash);
}

std::string compiler() {
    if (const char *e = getenv("VHIP_HIPCC")) return e;
    std::string root = getenv("ROCM_PATH") ? getenv("ROCM_PATH") : "/opt/rocm";
    return root + "/bin/hipcc";
}

unsigned long long fnv1a(const std::string &s, unsigned long long h = 1469598103934665603ull) {
    for (unsigned char c : s) {
        h ^= c;
        h *= 1099511628211ull;
    }
    return h;
}

bool stat_tag(const std::string &path, std::string *tag) {
    struct stat st;
    if (stat(path.c_str(), &st) != 0) return false;
    char buf[96];
    snprintf(buf, sizeof(buf), "%lld:%lld;", (long long)st.st_size, (long long)st.st_mtime);
    *tag += buf;
    return true;
}

}  // namespace

bool jit_enabled() {
    const char *e = getenv("VHIP_JIT");
    return !(e && e[0] == '0');
}

// Compiles `src` (a file next to the library) with -DVH_JIT_KERNEL and `defs`, loads the code object on the current device
// and returns the kernel `kname`.  Failure leaves *fn untouched and explains itself in *err.
bool jit_function(const char *src, const std::string &defs, const char *kname, hipFunction_t *fn, std::string *err) {
    const std::string dir = source_dir(), cc = compiler();
    const std::string path = dir + "/" + src;
    std::string tag;
    const char *deps[] = {src, "kernels.h", "viterbi_codes.h", "k15_layout.h", "k24t_layout.h", "../../include/viterbi_hip.h"};
    for (const char *d : deps)
        if (!stat_tag(dir + "/" + d, &tag)) {
            *err = std::string("runtime specialisation needs the kernel sources next to the library (missing ") + dir + "/" + d + ")";
            return false;
        }
    if (access(cc.c_str(), X_OK) != 0) {
        *err = "runtime specialisation needs hipcc (" + cc + " is not executable; set VHIP_HIPCC or ROCM_PATH)";
        return false;
    }
    std::string cache = getenv("VHIP_JIT_CACHE") ? getenv("VHIP_JIT_CACHE") : "/tmp/vhip_jit_cache_" + std::to_string((long)getuid());
    (void)mkdir(cache.c_str(), 0700);
    const std::string flags = "--genco --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DVH_JIT_KERNEL " + defs;
    char name[64];
    snprintf(name, sizeof(name), "/%016llx.hsaco", fnv1a(path + "|" + tag + "|" + flags + "|" + cc));
    const std::string obj = cache + name;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const std::string key = std::to_string(dev) + ":" + obj;

    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_modules.find(key);
    if (it == g_modules.end()) {
        if (access(obj.c_str(), R_OK) != 0) {
            const std::string tmp = obj + ".tmp" + std::to_string((long)getpid()), log = tmp + ".log";
            const std::string cmd = "'" + cc + "' " + flags + " -I '" + dir + "' '" + path + "' -o '" + tmp + "' > '" + log + "' 2>&1";
            if (getenv("VHIP_VERBOSE")) fprintf(stderr, "viterbi_hip: %s\n", cmd.c_str());
            const int rc = system(cmd.c_str());  // a child process; this one keeps running
            if (rc != 0 || rename(tmp.c_str(), obj.c_str()) != 0) {
                std::string tail;
                if (FILE *fp = fopen(log.c_str(), "r")) {
                    char buf[2048];
                    const size_t n = fread(buf, 1, sizeof(buf) - 1, fp);
                    buf[n] = 0;
                    tail = buf;
                    fclose(fp);
                }
                (void)unlink(tmp.c_str());
                (void)unlink(log.c_str());
                *err = "hipcc --genco failed for " + std::string(src) + " (" + defs + "): " + tail;
                return false;
            }
            (void)unlink(log.c_str());
        }
        hipModule_t mod = nullptr;
        const hipError_t e = hipModuleLoad(&mod, obj.c_str());
        if (e != hipSuccess) {
            *err = std::string("hipModuleLoad(") + obj + "): " + hipGetErrorString(e);
            return false;
        }
        it = g_modules.emplace(key, mod).first;
    }
    hipFunction_t f = nullptr;
    const hipError_t e = hipModuleGetFunction(&f, it->second, kname);
    if (e != hipSuccess) {
        *err = std::string("hipModuleGetFunction(") + kname + "): " + hipGetErrorString(e);
        return false;
    }
    *fn = f;
    return true;
}

std::string jit_poly_define(const int *poly, int n) {
    std::string s = "'-DVH_JIT_POLY={";
    for (int r = 0; r < n; r++) s += (r ? "," : "") + std::to_string(poly[r]);
    return s + "}'";
}

}  // namespace vh
