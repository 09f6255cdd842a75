// n1k_jit.cpp — run-time instantiation of scan_spec_body for one plan shape (see n1k_jit.h).
#include "n1k_jit.h"
#include "n1k_kernels.h"

#include <dlfcn.h>
#include <hip/hiprtc.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>

#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <sstream>
#include <vector>

namespace n1k {
namespace {

std::mutex g_mu;
std::map<std::string, JitKernel*> g_cache;

std::string csrc_dir() {
    if (const char* e = getenv("N1K_CSRC")) return e;
    Dl_info info;
    if (dladdr((const void*)&csrc_dir, &info) && info.dli_fname) {
        std::string p = info.dli_fname;
        size_t slash = p.rfind('/');
        if (slash != std::string::npos) return p.substr(0, slash) + "/csrc";
    }
    return "query_amd/csrc";
}

std::string sig_key(const SpecSig& s) { return std::string((const char*)&s, sizeof s); }

// ---- code objects on disk: a shape is compiled once per installation, not once per process -----------------------
//
// <directory of libn1k.so>/jit_cache/<hash>.co (N1K_JIT_CACHE names another directory; "0" or "off" disables it).  The
// hash covers the generated source text (the shape), every header under csrc/ the translation unit can include, the
// compiler options and the hiprtc version: a changed kernel source can never meet a stale object.  Files appear by rename.
uint64_t fnv1a(const void* p, size_t n, uint64_t h) {
    const unsigned char* b = (const unsigned char*)p;
    for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 0x100000001B3ull;
    return h;
}

std::string cache_dir() {
    const char* e = getenv("N1K_JIT_CACHE");
    if (e && (!strcmp(e, "0") || !strcmp(e, "off"))) return "";
    if (e && *e) return e;
    std::string c = csrc_dir();  // .../query_amd/csrc -> .../query_amd/jit_cache
    size_t slash = c.rfind('/');
    return (slash == std::string::npos ? std::string(".") : c.substr(0, slash)) + "/jit_cache";
}

uint64_t headers_hash() {
    static uint64_t cached = 0;
    if (cached) return cached;
    uint64_t h = 0xCBF29CE484222325ull;
    static const char* names[] = {"n1k_types.h", "n1k_device.h", "n1k_tables.h", "n1k_scatter.h", "n1k_spec.h"};
    for (const char* n : names) {
        std::string path = csrc_dir() + "/" + n;
        FILE* f = fopen(path.c_str(), "rb");
        if (!f) {  // a header the key cannot cover: no cache at all rather than a key that silently leaves it out
            cached = 0;
            return 0;
        }
        char buf[65536];
        size_t got;
        while ((got = fread(buf, 1, sizeof buf, f)) > 0) h = fnv1a(buf, got, h);
        fclose(f);
        h = fnv1a(n, strlen(n), h);
    }
    int major = 0, minor = 0;
    (void)hiprtcVersion(&major, &minor);
    h = fnv1a(&major, sizeof major, h);
    h = fnv1a(&minor, sizeof minor, h);
    cached = h ? h : 1;
    return cached;
}

// The cached object carries a trailer — magic, size, two independent 64-bit checksums of the bytes — which cache_read
// verifies: a truncated, corrupted or foreign file is recompiled, not loaded onto the GPU.
struct CacheTrailer {
    char magic[8];
    uint64_t size, sum_a, sum_b;
};
constexpr char kTrailerMagic[8] = {'N', '1', 'K', 'C', 'O', 'v', '1', 0};
void trailer_of(const char* p, size_t n, CacheTrailer& t) {
    memcpy(t.magic, kTrailerMagic, 8);
    t.size = n;
    t.sum_a = fnv1a(p, n, 0xCBF29CE484222325ull);
    uint64_t b = 0x9E3779B97F4A7C15ull;  // a second, unrelated mix over 8-byte words
    for (size_t i = 0; i + 8 <= n; i += 8) {
        uint64_t w;
        memcpy(&w, p + i, 8);
        b = (b ^ w) * 0xFF51AFD7ED558CCDull;
        b ^= b >> 29;
    }
    for (size_t i = n & ~(size_t)7; i < n; i++) b = (b ^ (unsigned char)p[i]) * 0x100000001B3ull;
    t.sum_b = b;
}

std::string cache_path(const std::string& src) {
    const std::string dir = cache_dir();
    if (dir.empty() || headers_hash() == 0) return "";
    uint64_t h = fnv1a(src.data(), src.size(), headers_hash());
    static const char kOpts[] = "--offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics";
    h = fnv1a(kOpts, sizeof kOpts, h);
    char name[40];
    snprintf(name, sizeof name, "/%016llx.co", (unsigned long long)h);
    return dir + name;
}

bool cache_read(const std::string& path, std::vector<char>& code) {
    if (path.empty()) return false;
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    bool ok = n > 64 + (long)sizeof(CacheTrailer);
    if (ok) {
        code.resize((size_t)n);
        ok = fread(code.data(), 1, (size_t)n, f) == (size_t)n && !memcmp(code.data(), "\177ELF", 4);
    }
    fclose(f);
    if (ok) {
        CacheTrailer have, want;
        const size_t body = (size_t)n - sizeof(CacheTrailer);
        memcpy(&have, code.data() + body, sizeof have);
        trailer_of(code.data(), body, want);
        ok = !memcmp(have.magic, kTrailerMagic, 8) && have.size == body && have.sum_a == want.sum_a && have.sum_b == want.sum_b;
        if (ok) code.resize(body);
    }
    if (!ok) code.clear();
    return ok;
}

void cache_write(const std::string& path, const std::vector<char>& code) {
    if (path.empty() || code.empty()) return;
    const std::string dir = path.substr(0, path.rfind('/'));
    (void)mkdir(dir.c_str(), 0755);  // (only its owner writes code objects that this library will load)
    char tmp[64];
    snprintf(tmp, sizeof tmp, ".tmp.%d.%p", (int)getpid(), (void*)&code);
    const std::string t = dir + "/" + tmp;
    FILE* f = fopen(t.c_str(), "wb");
    if (!f) return;
    CacheTrailer tr;
    trailer_of(code.data(), code.size(), tr);
    const bool ok = fwrite(code.data(), 1, code.size(), f) == code.size() && fwrite(&tr, 1, sizeof tr, f) == sizeof tr;
    fclose(f);
    if (!ok || rename(t.c_str(), path.c_str()) != 0) (void)remove(t.c_str());
}

bool compile(const SpecSig& sig, std::vector<char>& code, std::string& log, bool use_cache = true) {
    std::string src = jit_source(sig);
    const std::string cpath = use_cache ? cache_path(src) : std::string();
    if (cache_read(cpath, code)) return true;
    hiprtcProgram prog = nullptr;
    if (hiprtcCreateProgram(&prog, src.c_str(), "n1k_jit_shape.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) {
        log = "hiprtcCreateProgram failed";
        return false;
    }
    std::string inc = "-I" + csrc_dir();
    const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", inc.c_str()};
    hiprtcResult r = hiprtcCompileProgram(prog, 5, opts);
    size_t ls = 0;
    hiprtcGetProgramLogSize(prog, &ls);
    if (ls > 1) {
        log.resize(ls);
        hiprtcGetProgramLog(prog, &log[0]);
    }
    bool ok = r == HIPRTC_SUCCESS;
    if (ok) {
        size_t cs = 0;
        hiprtcGetCodeSize(prog, &cs);
        code.resize(cs);
        ok = cs > 0 && hiprtcGetCode(prog, code.data()) == HIPRTC_SUCCESS;
    }
    hiprtcDestroyProgram(&prog);
    if (ok) cache_write(cpath, code);
    return ok;
}

}  // namespace

std::string jit_source(const SpecSig& g) {
    std::ostringstream o;
    o << "#include \"n1k_spec.h\"\nnamespace n1k {\n";
    auto term = [&](int t) {
        std::ostringstream x;
        if (t < g.nterms) x << "SpecTerm{" << g.terms[t].op << "u, " << g.terms[t].col << "u, " << g.terms[t].const_int << "u}";
        else x << "SpecTerm{0u, 0u, 0u}";
        return x.str();
    };
    auto agg = [&](int a) {
        std::ostringstream x;
        if (a < g.naggs) x << "SpecAgg{" << g.aggs[a].kind << "u, " << g.aggs[a].has_operand << "u, " << g.aggs[a].col << "u, " << g.aggs[a].distinct << "u}";
        else x << "SpecAgg{0u, 0u, 0u, 0u}";
        return x.str();
    };
    auto derived = [&](int d) {
        std::ostringstream x;
        if (d < g.nderived) {
            x << "SpecDerived{" << g.derived[d].op << "u, " << g.derived[d].nops << "u, {";
            for (int k = 0; k < 4; k++) x << (k ? ", " : "") << "SpecOperand{" << g.derived[d].ops[k].is_const << "u, " << g.derived[d].ops[k].v << "u}";
            x << "}}";
        } else
            x << "SpecDerived{0u, 0u, {SpecOperand{0u, 0u}, SpecOperand{0u, 0u}, SpecOperand{0u, 0u}, SpecOperand{0u, 0u}}}";
        return x.str();
    };
    o << "struct SpecJ {\n"
      << "    static constexpr int ncols = " << g.ncols << ", nterms = " << g.nterms << ", nkeys = " << g.nkeys
      << ", naggs = " << g.naggs << ", nderived = " << g.nderived << ";\n"
      << "    static constexpr SpecDerived derived[kFastDerived] = {" << derived(0) << ", " << derived(1) << ", " << derived(2) << "};\n"
      << "    static constexpr uint32_t col_kind[kFastCols] = {" << g.col_kind[0] << "u, " << g.col_kind[1] << "u, "
      << g.col_kind[2] << "u};\n"
      << "    static constexpr SpecTerm terms[kFastTerms] = {" << term(0) << ", " << term(1) << "};\n"
      << "    static constexpr uint32_t key_col[kFastKeys] = {" << g.key_col[0] << "u, " << g.key_col[1] << "u};\n"
      << "    static constexpr SpecAgg aggs[kFastAggs] = {" << agg(0) << ", " << agg(1) << ", " << agg(2) << ", " << agg(3)
      << ", " << agg(4) << "};\n};\n}  // namespace n1k\n";
    if (g.mode == 1) {
        if (const char* a = getenv("N1K_JIT_PART_ATTR")) o << "#define N1K_PART_ATTR " << a << "\n";  // (tuning experiments)
        else o << "#define N1K_PART_ATTR\n";

        o << "extern \"C\" __global__ __launch_bounds__(512) N1K_PART_ATTR void n1k_jit_part_wide(const n1k::Program P, const n1k::FastArgs F,\n"
          << "        const n1k::PartArgs A) {\n    n1k::scan_spec_partition_body<n1k::SpecJ, 2, 512, true>(P, F, A);\n}\n"
          << "extern \"C\" __global__ __launch_bounds__(256) N1K_PART_ATTR void n1k_jit_part_wide256(const n1k::Program P, const n1k::FastArgs F,\n"
          << "        const n1k::PartArgs A) {\n    n1k::scan_spec_partition_body<n1k::SpecJ, 2, 256, true, false>(P, F, A);\n}\n"
          << "extern \"C\" __global__ __launch_bounds__(512) N1K_PART_ATTR void n1k_jit_part_narrow(const n1k::Program P, const n1k::FastArgs F,\n"
          << "        const n1k::PartArgs A) {\n    n1k::scan_spec_partition_body<n1k::SpecJ, 4, 512, false>(P, F, A);\n}\n";
        return o.str();
    }
    o << "extern \"C\" __global__ __launch_bounds__(512) void n1k_jit_wide(const n1k::Program P, const n1k::FastArgs F,\n"
      << "        const n1k::GlobalTable G, unsigned long long* ngroups, const n1k::WordLogArgs L) {\n"
      << "    n1k::scan_spec_body<n1k::SpecJ, 2, 512, true, " << (g.seg ? "true" : "false") << ">(P, F, G, ngroups, L);\n}\n"
      << "extern \"C\" __global__ __launch_bounds__(512) void n1k_jit_narrow(const n1k::Program P, const n1k::FastArgs F,\n"
      << "        const n1k::GlobalTable G, unsigned long long* ngroups, const n1k::WordLogArgs L) {\n"
      << "    n1k::scan_spec_body<n1k::SpecJ, 4, 512, false, " << (g.seg ? "true" : "false") << ">(P, F, G, ngroups, L);\n}\n"
      << "extern \"C\" __global__ __launch_bounds__(512) void n1k_jit_rec_wide(const n1k::Program P, const n1k::FastArgs F,\n"
      << "        const n1k::WordLogArgs L) {\n    n1k::scan_spec_records_body<n1k::SpecJ, 2, 512, true>(P, F, L);\n}\n"
      << "extern \"C\" __global__ __launch_bounds__(512) void n1k_jit_rec_narrow(const n1k::Program P, const n1k::FastArgs F,\n"
      << "        const n1k::WordLogArgs L) {\n    n1k::scan_spec_records_body<n1k::SpecJ, 4, 512, false>(P, F, L);\n}\n";
    return o.str();
}

bool jit_compile_check(const SpecSig& sig, std::string* log) {
    std::vector<char> code;
    std::string l;
    bool ok = compile(sig, code, l, false);  // (a check compiles: it neither reads nor fills the cache)
    if (log) *log = l;
    return ok;
}

static bool load_module(const SpecSig& sig, JitKernel* k, const std::vector<char>& code) {
    if (hipModuleLoadData(&k->module, code.data()) != hipSuccess) return false;
    if (sig.mode == 1)
        return hipModuleGetFunction(&k->part_wide, k->module, "n1k_jit_part_wide") == hipSuccess &&
               hipModuleGetFunction(&k->part_wide256, k->module, "n1k_jit_part_wide256") == hipSuccess &&
               hipModuleGetFunction(&k->part_narrow, k->module, "n1k_jit_part_narrow") == hipSuccess;
    return hipModuleGetFunction(&k->wide, k->module, "n1k_jit_wide") == hipSuccess &&
           hipModuleGetFunction(&k->narrow, k->module, "n1k_jit_narrow") == hipSuccess &&
           hipModuleGetFunction(&k->rec_wide, k->module, "n1k_jit_rec_wide") == hipSuccess &&
           hipModuleGetFunction(&k->rec_narrow, k->module, "n1k_jit_rec_narrow") == hipSuccess;
}

const JitKernel* jit_get(const SpecSig& sig) {
    std::lock_guard<std::mutex> lock(g_mu);
    std::string key = sig_key(sig);
    auto it = g_cache.find(key);
    if (it != g_cache.end()) return it->second;
    JitKernel* k = new JitKernel();
    g_cache[key] = k;
    std::vector<char> code;
    if (!compile(sig, code, k->log)) {
        k->failed = true;
        return k;
    }
    if (!load_module(sig, k, code)) {
        // (an object from the disk cache that does not load — truncated, another driver: compile afresh, once)
        if (k->module) (void)hipModuleUnload(k->module);
        *k = JitKernel();
        const std::string cpath = cache_path(jit_source(sig));
        if (!cpath.empty()) (void)remove(cpath.c_str());
        code.clear();
        if (!compile(sig, code, k->log, false) || !load_module(sig, k, code)) {
            k->failed = true;
            k->log += "\nhipModuleLoadData / hipModuleGetFunction failed";
        } else
            cache_write(cpath, code);
    }
    return k;
}

hipError_t jit_launch(const JitKernel* k, const Program& P, const FastArgs& F, const GlobalTable& G,
                      unsigned long long* ngroups, uint32_t grid, bool wide, const WordLogArgs& L, uint32_t ndistinct, hipStream_t st) {
    size_t shmem = (size_t)F.lds_slots * P.lds_words * 8 + (size_t)L.dcache_slots * ndistinct * 8;
    void* args[] = {(void*)&P, (void*)&F, (void*)&G, (void*)&ngroups, (void*)&L};
    return hipModuleLaunchKernel(wide ? k->wide : k->narrow, grid, 1, 1, 512, 1, 1, (unsigned)shmem, st, args, nullptr);
}

hipError_t jit_launch_records(const JitKernel* k, const Program& P, const FastArgs& F, uint32_t grid, bool wide, const WordLogArgs& L,
                              hipStream_t st) {
    const unsigned shmem = (unsigned)spec_records_lds_bytes();  // sizeof(ScatterLds<Rec16, 512, 4>), n1k_scatter.h
    hipFunction_t f = wide ? k->rec_wide : k->rec_narrow;
    void* args[] = {(void*)&P, (void*)&F, (void*)&L};
    return hipModuleLaunchKernel(f, grid, 1, 1, 512, 1, 1, shmem, st, args, nullptr);
}

hipError_t jit_launch_partition(const JitKernel* k, const Program& P, const FastArgs& F, const PartArgs& A, uint32_t grid, bool wide,
                                uint32_t block, hipStream_t st) {
    void* args[] = {(void*)&P, (void*)&F, (void*)&A};
    if (wide && block == 256) return hipModuleLaunchKernel(k->part_wide256, grid, 1, 1, 256, 1, 1, 0, st, args, nullptr);
    return hipModuleLaunchKernel(wide ? k->part_wide : k->part_narrow, grid, 1, 1, 512, 1, 1, 0, st, args, nullptr);
}

}  // namespace n1k
