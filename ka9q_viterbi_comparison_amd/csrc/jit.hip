// jit.hip -- runtime specialisation of the fast ACS kernels for polynomials other than the harness set.
//
// The reference builds its branch table from whatever `poly` the caller passes (ka9q_libfec_port/viterbi27_sse2.cpp:57-75,
// viterbi224_sse2.cpp:68-73) at no speed cost.  The fast kernels here (acs_regs.hip, acs_k15.hip, acs_k24t.hip) need the
// branch-table class of every register at every unrolled phase as a compile-time constant -- a per-register table lookup
// costs more than the add-compare-select it feeds -- so for other polynomials the SAME source file is compiled once more,
// at create time, with those polynomials as its constants (`hipcc --genco`, 2-6 s per kernel family), cached on disk by a
// hash of source contents + options, and launched through hipModuleLaunchKernel.  Kernel code, data layouts and chainback
// kernels are identical to the harness-polynomial path.  When the sources or the compiler are not there the handle falls
// back to the any-polynomial kernels (acs_lds.hip / acs_k24.hip); nothing here ever decodes on the CPU.
//
// What is trusted, and how it is checked:
//  * the kernel sources next to the library must be the ones the library itself was built from: the Makefile bakes an
//    FNV-1a fingerprint of their contents into the library (jit_sources_hash.h) and a mismatch refuses the run-time build
//    (a run-time kernel with another decision layout or argument struct than the library's chainback kernels expect would
//    decode silently wrong);
//  * the cache directory ($VHIP_JIT_CACHE, else $XDG_CACHE_HOME/viterbi_hip_jit, else ~/.cache/viterbi_hip_jit) must be a
//    real directory owned by the caller and closed to group and others; a cached code object is opened with O_NOFOLLOW, must
//    be a regular file of the caller's, and is loaded from memory (hipModuleLoadData), so nobody else can plant or swap it;
//  * the compiler is started with posix_spawn and an argument vector -- no shell, nothing to quote -- and with an
//    environment from which profiler / preload variables are removed (under rocprofv3 the preload would initialise the GPU
//    in hipcc and each program it starts in turn; pre-warm the cache outside the profiler anyway: DESIGN.md §4.8).
#include <dlfcn.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <pwd.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "jit_sources_hash.h"
#include "kernels.h"

extern char **environ;

namespace vh {

namespace {

std::mutex g_mu;
std::map<std::string, hipModule_t> g_modules;  // key: device id + cache file

// the files a run-time build reads, in the order the Makefile hashes them (JIT_SRCS)
const char *const kJitSources[] = {"acs_regs.hip", "acs_k15.hip", "acs_k24t.hip", "kernels.h", "viterbi_codes.h", "k24t_layout.h", "k15_layout.h"};

std::string source_dir() {
    Dl_info info;
    if (!dladdr(reinterpret_cast<void *>(&jit_enabled), &info) || !info.dli_fname) return "";
    std::string p = info.dli_fname;
    const size_t slash = p.rfind('/');
    return slash == std::string::npos ? "." : p.substr(0, slash);
}

std::string compiler() {
    if (const char *e = getenv("VHIP_HIPCC")) return e;
    const char *rocm = getenv("ROCM_PATH");
    return std::string(rocm ? rocm : "/opt/rocm") + "/bin/hipcc";
}

constexpr unsigned long long FNV_BASIS = 0xCBF29CE484222325ull, FNV_PRIME = 0x100000001B3ull;  // as tools/kernel_hash.py
unsigned long long fnv1a(const void *data, size_t n, unsigned long long h) {
    const unsigned char *p = static_cast<const unsigned char *>(data);
    for (size_t i = 0; i < n; i++) {
        h ^= p[i];
        h *= FNV_PRIME;
    }
    return h;
}
unsigned long long fnv1a(const std::string &s, unsigned long long h = FNV_BASIS) { return fnv1a(s.data(), s.size(), h); }

bool read_file(int fd, std::string *out) {
    char buf[65536];
    for (;;) {
        const ssize_t n = read(fd, buf, sizeof(buf));
        if (n < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        if (n == 0) return true;
        out->append(buf, (size_t)n);
    }
}

// FNV-1a over the contents of the run-time-compiled sources, as tools/kernel_hash.py --fnv computes it for the Makefile
bool sources_fingerprint(const std::string &dir, unsigned long long *h, std::string *err) {
    unsigned long long acc = FNV_BASIS;
    for (const char *name : kJitSources) {
        const std::string path = dir + "/" + name;
        const int fd = open(path.c_str(), O_RDONLY | O_CLOEXEC);
        std::string body;
        const bool ok = fd >= 0 && read_file(fd, &body);
        if (fd >= 0) close(fd);
        if (!ok) {
            *err = "runtime specialisation needs the kernel sources next to the library (cannot read " + path + ")";
            return false;
        }
        acc = fnv1a(body.data(), body.size(), acc);
    }
    *h = acc;
    return true;
}

bool mkdir_p(const std::string &path) {
    for (size_t i = 1; i <= path.size(); i++) {
        if (i != path.size() && path[i] != '/') continue;
        const std::string part = path.substr(0, i);
        struct stat st;
        if (mkdir(part.c_str(), 0700) != 0 && !(stat(part.c_str(), &st) == 0 && S_ISDIR(st.st_mode))) return false;
    }
    return true;
}

// A directory only this user can write to, or "" with the reason in *err.  Candidates in order: $VHIP_JIT_CACHE (if set, the
// only one), $XDG_CACHE_HOME/viterbi_hip_jit, ~/.cache/viterbi_hip_jit, /tmp/viterbi_hip_jit_<uid>.  The checks are the same
// for all of them, which is what makes the /tmp candidate safe: a directory somebody else created there first is not the
// caller's and is refused.
bool private_dir(const std::string &dir, std::string *why) {
    if (!mkdir_p(dir)) {
        *why = "cannot create " + dir + ": " + strerror(errno);
        return false;
    }
    struct stat st;
    if (lstat(dir.c_str(), &st) != 0 || !S_ISDIR(st.st_mode) || st.st_uid != geteuid() || (st.st_mode & 077) != 0) {
        *why = dir + " must be a directory (not a link) owned by the caller with mode 0700";
        return false;
    }
    if (access(dir.c_str(), W_OK | X_OK) != 0) {
        *why = dir + " is not writable";
        return false;
    }
    return true;
}

std::string cache_dir(std::string *err) {
    std::vector<std::string> cand;
    if (const char *e = getenv("VHIP_JIT_CACHE")) cand.push_back(e);
    else {
        if (const char *x = getenv("XDG_CACHE_HOME"); x && x[0] == '/') cand.push_back(std::string(x) + "/viterbi_hip_jit");
        const char *home = getenv("HOME");
        if (!home || home[0] != '/') {
            if (const struct passwd *pw = getpwuid(geteuid())) home = pw->pw_dir;
        }
        if (home && home[0] == '/' && home[1]) cand.push_back(std::string(home) + "/.cache/viterbi_hip_jit");
        cand.push_back("/tmp/viterbi_hip_jit_" + std::to_string((long)geteuid()));
    }
    std::string reasons;
    for (const std::string &dir : cand) {
        std::string why;
        if (private_dir(dir, &why)) return dir;
        reasons += (reasons.empty() ? "" : "; ") + why;
    }
    *err = "no private cache directory for run-time builds (" + reasons + "): run-time builds are off";
    return "";
}

// the code object as bytes: only a regular file of the caller's, never through a symbolic link
bool read_code_object(const std::string &path, std::string *image) {
    const int fd = open(path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
    if (fd < 0) return false;
    struct stat st;
    const bool ok = fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_uid == geteuid() && (st.st_mode & 022) == 0 && st.st_size > 0 && read_file(fd, image);
    close(fd);
    return ok && image->size() == (size_t)st.st_size;
}

bool scrubbed(const char *kv) {
    static const char *const drop[] = {"LD_PRELOAD=", "HSA_TOOLS_LIB=", "HSA_TOOLS_REPORT_LOAD_FAILURE=", "ROCP", "ROCPROF", "ROCTRACER", "ROCTX", "HIP_TOOLS"};
    for (const char *d : drop)
        if (strncmp(kv, d, strlen(d)) == 0) return true;
    return false;
}

// runs argv[0] with argv, stdout + stderr into `log`; returns the exit status (or -1)
int run_compiler(const std::vector<std::string> &argv, const std::string &log) {
    std::vector<char *> av;
    for (const std::string &a : argv) av.push_back(const_cast<char *>(a.c_str()));
    av.push_back(nullptr);
    std::vector<char *> ev;
    for (char **e = environ; e && *e; e++)
        if (!scrubbed(*e)) ev.push_back(*e);
    ev.push_back(nullptr);
    posix_spawn_file_actions_t fa;
    if (posix_spawn_file_actions_init(&fa) != 0) return -1;
    posix_spawn_file_actions_addopen(&fa, 1, log.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0600);
    posix_spawn_file_actions_adddup2(&fa, 1, 2);
    posix_spawn_file_actions_addopen(&fa, 0, "/dev/null", O_RDONLY, 0);
    pid_t pid = -1;
    const int rc = posix_spawn(&pid, av[0], &fa, nullptr, av.data(), ev.data());
    posix_spawn_file_actions_destroy(&fa);
    if (rc != 0) return -1;
    int status = 0;
    while (waitpid(pid, &status, 0) < 0)
        if (errno != EINTR) return -1;
    return WIFEXITED(status) ? WEXITSTATUS(status) : -1;
}

}  // namespace

// 1: the kernel sources next to the library are the ones it was built from; 0: they differ (run-time builds refused); -1: unreadable
int jit_sources_match(std::string *why) {
    unsigned long long h = 0;
    if (!sources_fingerprint(source_dir(), &h, why)) return -1;
    if (h != VH_JIT_SOURCES_HASH) {
        *why = "the kernel sources in " + source_dir() + " are not the ones this library was built from (edited without a rebuild?)";
        return 0;
    }
    return 1;
}

bool jit_enabled() {
    const char *e = getenv("VHIP_JIT");
    return !(e && e[0] == '0');
}

// Compiles `src` (a file next to the library) with -DVH_JIT_KERNEL and `defs` (one compiler argument each), loads the code
// object on the current device and returns the kernel `kname`.  Failure leaves *fn untouched and explains itself in *err.
bool jit_function(const char *src, const std::vector<std::string> &defs, const char *kname, hipFunction_t *fn, std::string *err) {
    const std::string dir = source_dir(), cc = compiler();
    const std::string path = dir + "/" + src;
    if (jit_sources_match(err) != 1) {
        *err += ": run-time builds refused";
        return false;
    }
    const unsigned long long srch = VH_JIT_SOURCES_HASH;
    if (access(cc.c_str(), X_OK) != 0) {
        *err = "runtime specialisation needs hipcc (" + cc + " is not executable; set VHIP_HIPCC or ROCM_PATH)";
        return false;
    }
    const std::string cache = cache_dir(err);
    if (cache.empty()) return false;
    std::vector<std::string> flags = {"--genco", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-DVH_JIT_KERNEL"};
    flags.insert(flags.end(), defs.begin(), defs.end());
    std::string flat;
    for (const std::string &f : flags) flat += f + '\x1f';
    char name[64];
    snprintf(name, sizeof(name), "/%016llx%016llx.hsaco", srch, fnv1a(std::string(src) + "|" + flat + "|" + cc));
    const std::string obj = cache + name;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const std::string key = std::to_string(dev) + ":" + obj;

    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_modules.find(key);
    if (it == g_modules.end()) {
        std::string image;
        if (!read_code_object(obj, &image)) {
            image.clear();
            const std::string tmp = obj + ".tmp" + std::to_string((long)getpid()), log = tmp + ".log";
            std::vector<std::string> argv = {cc};
            argv.insert(argv.end(), flags.begin(), flags.end());
            argv.insert(argv.end(), {"-I", dir, path, "-o", tmp});
            if (getenv("VHIP_VERBOSE")) {
                std::string line;
                for (const std::string &a : argv) line += a + " ";
                fprintf(stderr, "viterbi_hip: %s\n", line.c_str());
            }
            (void)unlink(tmp.c_str());
            const int rc = run_compiler(argv, log);  // a child process; this one keeps running
            if (rc != 0 || chmod(tmp.c_str(), 0600) != 0 || rename(tmp.c_str(), obj.c_str()) != 0 || !read_code_object(obj, &image)) {
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
                *err = "hipcc --genco failed for " + std::string(src) + ": " + tail;
                return false;
            }
            (void)unlink(log.c_str());
        }
        hipModule_t mod = nullptr;
        const hipError_t e = hipModuleLoadData(&mod, image.data());
        if (e != hipSuccess) {
            *err = std::string("hipModuleLoadData(") + obj + "): " + hipGetErrorString(e);
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
    std::string s = "-DVH_JIT_POLY={";
    for (int r = 0; r < n; r++) s += (r ? "," : "") + std::to_string(poly[r]);
    return s + "}";
}

}  // namespace vh
