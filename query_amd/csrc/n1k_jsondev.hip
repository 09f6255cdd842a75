// n1k_jsondev.hip — raw JSON documents -> the plan's leaf columns, on the device.
//
// Replaces, for a whole batch at once, what the reference does per row and per referenced field while the operators run:
// Field.Apply over a parsedValue (expression/nav_field.go:134-160 -> value/parsed.go:159-207: go_json.FirstFind of the field
// in the raw bytes) and value.NewValue's typing of what it finds (value/value.go:367-430: integer literals that fit int64 are
// INT, a float64 with no fraction folds to INT).  The document bytes cross PCIe once, as they are; the columns are born in
// HBM and the scan kernels read them where they lie.
//
// One lane per document.  A wave stages the bytes of its (up to 64, consecutive) documents in LDS with coalesced 16-byte
// loads and every lane then scans its own document there: an iterative descent along the wanted field chains (first field
// of a name wins, as FirstFind), a validating skip of everything else.  The kernel takes the common shapes itself and hands
// the rest to the host's scalar extractor by marking the document (status 1): escapes in a wanted string or in a name that
// is compared, array / object values of a wanted path, numbers beyond the exactly convertible range (> 18 digits; > 15
// digits or |exponent| > 22 with a fraction or exponent), documents that are not objects, documents larger than the wave's
// LDS share, anything that does not parse — the host then produces the value or names the malformed document, exactly as
// n1k_extract_json does (n1k_json.cpp).  Strings become provisional ids: slots of a per-batch open-addressed table keyed by
// a 64-bit hash of the bytes and verified against the first occurrence; only the DISTINCT strings travel to the host, whose
// dictionary gives them their codes, and a second small kernel rewrites the ids.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "n1k_device.h"
#include "n1k_kernels.h"

namespace n1k {

namespace {

constexpr uint32_t kJsonWaveBytes = 16 * 1024;  // LDS per wave
constexpr uint32_t kJsonBlock = 256;

__constant__ double kPow10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                  1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

struct Cur {
    const uint8_t* s;  // the document in LDS
    uint32_t p, e;     // position, end
    __device__ __forceinline__ int peek() const { return p < e ? (int)s[p] : -1; }
    __device__ __forceinline__ void ws() {
        while (p < e) {
            const uint8_t c = s[p];
            if (c != ' ' && c != '\n' && c != '\t' && c != '\r') break;
            p++;
        }
    }
};

__device__ __forceinline__ bool is_hex(uint8_t c) { return (c >= '0' && c <= '9') || (c >= 'a' && c <= 'f') || (c >= 'A' && c <= 'F'); }

// a string from its opening quote to behind its closing one; `escaped` = a backslash was met (escapes are validated)
__device__ bool scan_string(Cur& c, uint32_t& b, uint32_t& n, bool& escaped, uint64_t& hash) {
    if (c.peek() != '"') return false;
    c.p++;
    b = c.p;
    escaped = false;
    uint64_t h = 0xcbf29ce484222325ull;
    while (c.p < c.e) {
        const uint8_t ch = c.s[c.p];
        if (ch == '"') {
            n = c.p - b;
            c.p++;
            hash = mix64(h) | 1ull;
            return true;
        }
        if (ch == '\\') {
            escaped = true;
            if (c.p + 1 >= c.e) return false;
            const uint8_t x = c.s[c.p + 1];
            if (x == 'u') {
                if (c.p + 6 > c.e) return false;
                for (int i = 2; i < 6; i++)
                    if (!is_hex(c.s[c.p + i])) return false;
                c.p += 6;
            } else if (x == '"' || x == '\\' || x == '/' || x == 'b' || x == 'f' || x == 'n' || x == 'r' || x == 't')
                c.p += 2;
            else
                return false;
            continue;
        }
        h = (h ^ ch) * 0x100000001b3ull;
        c.p++;
    }
    return false;
}

__device__ bool scan_literal(Cur& c, const char* lit, uint32_t n) {
    if (c.p + n > c.e) return false;
    for (uint32_t i = 0; i < n; i++)
        if (c.s[c.p + i] != (uint8_t)lit[i]) return false;
    c.p += n;
    return true;
}

// number -> tagged value; 0 = malformed, 1 = typed here, 2 = well formed but beyond the exact fast paths (host)
__device__ int scan_number(Cur& c, uint32_t& tag, uint64_t& payload) {
    bool neg = false;
    if (c.peek() == '-') {
        neg = true;
        c.p++;
    }
    int ch = c.peek();
    if (ch < '0' || ch > '9') return 0;
    unsigned long long m = 0;
    int nd = 0;  // significant digits taken into m (leading zeros do not count)
    bool over = false;
    while ((ch = c.peek()) >= '0' && ch <= '9') {
        if (m || ch != '0') {
            if (nd < 18) { m = m * 10ull + (unsigned)(ch - '0'); nd++; }
            else over = true;
        }
        c.p++;
    }
    bool integral = true;
    int frac = 0;
    if (c.peek() == '.') {
        integral = false;
        c.p++;
        ch = c.peek();
        if (ch < '0' || ch > '9') return 0;
        while ((ch = c.peek()) >= '0' && ch <= '9') {
            if (m || ch != '0') {
                if (nd < 18) { m = m * 10ull + (unsigned)(ch - '0'); nd++; frac++; }
                else over = true;
            } else
                frac++;
            c.p++;
        }
    }
    int ex = 0;
    if (c.peek() == 'e' || c.peek() == 'E') {
        integral = false;
        c.p++;
        bool eneg = false;
        if (c.peek() == '+' || c.peek() == '-') {
            eneg = c.peek() == '-';
            c.p++;
        }
        ch = c.peek();
        if (ch < '0' || ch > '9') return 0;
        while ((ch = c.peek()) >= '0' && ch <= '9') {
            if (ex < 100000) ex = ex * 10 + (ch - '0');
            c.p++;
        }
        if (eneg) ex = -ex;
    }
    if (over) return 2;
    if (integral) {  // at most 18 digits: an int64, exactly (value/value.go:375-376)
        tag = T_INT;
        payload = (uint64_t)(neg ? -(long long)m : (long long)m);
        return 1;
    }
    // Clinger's exact case: m < 2^53 and |e10| <= 22 -> one correctly rounded multiplication or division
    const int e10 = ex - frac;
    if (nd > 15 || e10 > 22 || e10 < -22) return 2;
    double d = (double)m;
    d = e10 >= 0 ? d * kPow10[e10] : d / kPow10[-e10];
    if (neg) d = -d;
    if (d >= -9223372036854775808.0 && d < 9223372036854775808.0 && d == (double)(long long)d) {  // NewValue folds it
        tag = T_INT;
        payload = (uint64_t)(long long)d;
    } else {
        tag = T_FLOAT;
        payload = (uint64_t)__double_as_longlong(d);
    }
    return 1;
}

// any value, validated and skipped; objects and arrays without recursion (kinds of the open brackets in a 64-bit stack)
__device__ bool skip_value(Cur& c) {
    unsigned long long kinds = 0;  // bit i: bracket at depth i is an object
    int depth = 0;
    for (;;) {
        c.ws();
        const int ch = c.peek();
        bool opened = false;
        uint32_t b, n;
        bool esc;
        uint64_t hh;
        if (ch == '"') {
            if (!scan_string(c, b, n, esc, hh)) return false;
        } else if (ch == '{' || ch == '[') {
            c.p++;
            c.ws();
            if (c.peek() == (ch == '{' ? '}' : ']')) c.p++;
            else {
                if (depth >= 64) return false;  // (deeper documents go to the host)
                kinds = ch == '{' ? (kinds | (1ull << depth)) : (kinds & ~(1ull << depth));
                depth++;
                if (ch == '{') {
                    if (!scan_string(c, b, n, esc, hh)) return false;
                    c.ws();
                    if (c.peek() != ':') return false;
                    c.p++;
                }
                opened = true;
            }
        } else if (ch == 't') {
            if (!scan_literal(c, "true", 4)) return false;
        } else if (ch == 'f') {
            if (!scan_literal(c, "false", 5)) return false;
        } else if (ch == 'n') {
            if (!scan_literal(c, "null", 4)) return false;
        } else {
            uint32_t t;
            uint64_t pv;
            if (!scan_number(c, t, pv)) return false;
        }
        if (opened) continue;
        for (;;) {  // a value ended: close what it completes, or move on to the next member / element
            if (depth == 0) return true;
            c.ws();
            const bool obj = (kinds >> (depth - 1)) & 1ull;
            const int x = c.peek();
            if (x == ',') {
                c.p++;
                if (obj) {
                    c.ws();
                    if (!scan_string(c, b, n, esc, hh)) return false;
                    c.ws();
                    if (c.peek() != ':') return false;
                    c.p++;
                }
                break;
            }
            if (x == (obj ? '}' : ']')) {
                c.p++;
                depth--;
                continue;
            }
            return false;
        }
    }
}

// provisional id of a string (slot of the batch's table), or false: a 64-bit collision of different bytes / a full table
__device__ bool string_id(const JsonDevArgs& A, const uint8_t* lds_bytes, uint32_t b, uint32_t n, uint64_t hash, uint64_t goff,
                          uint64_t& id) {
    const uint64_t mask = (1ull << A.tab_bits) - 1ull;
    uint64_t slot = hash & mask;
    if (n >= (1u << 24)) return false;
    for (uint32_t probe = 0; probe < 128; probe++, slot = (slot + 1) & mask) {
        unsigned long long cur = __hip_atomic_load(&A.tab_hash[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == 0ull) {
            unsigned long long expected = 0ull;
            if (__hip_atomic_compare_exchange_strong(&A.tab_hash[slot], &expected, (unsigned long long)hash, __ATOMIC_RELAXED,
                                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                // the first occurrence: [offset in the batch's bytes : 40][length : 24], published by one 8-byte store
                __hip_atomic_store(&A.tab_first[slot], (unsigned long long)((goff << 24) | n | (1ull << 63)), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long at = atomicAdd(A.new_count, 1ull);
                if (at < A.new_cap) A.new_list[at] = (uint32_t)slot;
                id = slot;
                return true;
            }
            cur = expected;
        }
        if (cur != hash) continue;
        unsigned long long first;
        do {  // (its owner publishes right after winning the slot)
            first = __hip_atomic_load(&A.tab_first[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } while (first == 0ull);
        first &= ~(1ull << 63);
        if ((uint32_t)(first & 0xFFFFFFull) != n) return false;
        const uint8_t* o = A.bytes + (first >> 24);
        for (uint32_t i = 0; i < n; i++)
            if (o[i] != lds_bytes[b + i]) return false;
        id = slot;
        return true;
    }
    return false;
}

}  // namespace

__global__ __launch_bounds__(kJsonBlock) void json_extract_kernel(const JsonDevArgs A) {
    extern __shared__ uint8_t jlds[];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint8_t* const buf = jlds + (size_t)wave * kJsonWaveBytes;
    const uint64_t nwaves = (uint64_t)gridDim.x * (kJsonBlock / 64);
    const uint64_t wid = (uint64_t)blockIdx.x * (kJsonBlock / 64) + wave;
    for (uint64_t base = wid * 64; base < A.ndocs; base += nwaves * 64) {
        const uint64_t d = base + lane;
        const bool have = d < A.ndocs;
        const uint64_t g0 = have ? A.offsets[d] - A.base : 0, g1 = have ? A.offsets[d + 1] - A.base : 0;
        // sub-batches of consecutive documents whose bytes fit the wave's LDS share
        uint32_t first = 0;
        const uint32_t ndocs_here = (uint32_t)min((uint64_t)64, A.ndocs - base);
        while (first < ndocs_here) {
            const uint64_t start = __shfl(g0, (int)first, 64);
            const uint64_t a0 = start & ~15ull;  // staged from a 16-byte boundary
            const bool fits = have && lane >= first && g1 >= g0 && g0 >= a0 && g1 - a0 <= kJsonWaveBytes - 16;
            // the lanes from `first` on that fit form a prefix (offsets are monotone): its length
            const unsigned long long fm = __ballot(fits) >> first;
            const uint32_t cnt = ~fm == 0ull ? 64u : (uint32_t)__ffsll((long long)~fm) - 1u;  // run of ones from bit 0
            if (cnt == 0) {  // this document alone is larger than the wave's share (or its offsets are not monotone): the host takes it
                if (lane == first) A.status[d] = 1;
                first++;
                continue;
            }
            const uint64_t stop = __shfl(g1, (int)(first + cnt - 1), 64);
            const uint32_t nbytes = (uint32_t)(stop - a0);
            for (uint32_t i = lane * 16; i < nbytes; i += 64 * 16)  // (the byte buffer carries 32 spare bytes behind its end)
                *(uint4*)(buf + i) = *(const uint4*)(A.bytes + a0 + i);
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if (lane >= first && lane < first + cnt) {
                Cur c{buf, (uint32_t)(g0 - a0), (uint32_t)(g1 - a0)};
                uint32_t tg[kMaxCols];
                uint64_t pv[kMaxCols];
#pragma unroll
                for (int i = 0; i < kMaxCols; i++) {
                    tg[i] = T_MISSING;
                    pv[i] = 0;
                }
                bool ok = true;
                c.ws();
                if (c.peek() != '{') ok = false;  // (a scalar or array document: every path is MISSING; the host validates it)
                // levels of wanted objects: the paths still alive at each level, and those found there (first field wins)
                uint32_t active[kJsonMaxSteps + 1], found[kJsonMaxSteps + 1];
                int level = 0;
                active[0] = A.npaths >= 32 ? 0xFFFFFFFFu : ((1u << A.npaths) - 1u);
                found[0] = 0;
                if (ok) {
                    c.p++;
                    c.ws();
                    if (c.peek() == '}') {
                        c.p++;
                        level = -1;
                    }
                }
                while (ok && level >= 0) {
                    // a member of the object at `level`
                    c.ws();
                    uint32_t kb, kn;
                    bool kesc;
                    uint64_t kh;
                    if (!scan_string(c, kb, kn, kesc, kh)) { ok = false; break; }
                    c.ws();
                    if (c.peek() != ':') { ok = false; break; }
                    c.p++;
                    c.ws();
                    uint32_t hit = 0;
                    const uint32_t cand = active[level] & ~found[level];
                    if (cand) {
                        if (kesc) { ok = false; break; }  // (an escaped name could spell a wanted one: the host compares)
                        for (uint32_t i = 0; i < A.npaths; i++) {
                            if (!((cand >> i) & 1u)) continue;
                            const JsonDevPath& P = A.paths[i];
                            if (P.name_len[level] != kn) continue;
                            bool same = true;
                            for (uint32_t k = 0; k < kn && same; k++) same = (uint8_t)A.names[P.name_off[level] + k] == c.s[kb + k];
                            if (same) hit |= 1u << i;
                        }
                    }
                    found[level] |= hit;
                    uint32_t leafs = 0, deeper = 0;
                    for (uint32_t i = 0; i < A.npaths; i++)
                        if ((hit >> i) & 1u) {
                            if (A.paths[i].nsteps == (uint32_t)level + 1u) leafs |= 1u << i;
                            else deeper |= 1u << i;
                        }
                    const int v0 = c.peek();
                    bool descended = false;
                    if (leafs) {
                        uint32_t t = T_MISSING;
                        uint64_t v = 0;
                        if (v0 == '"') {
                            uint32_t sb, sn;
                            bool sesc;
                            uint64_t sh;
                            if (!scan_string(c, sb, sn, sesc, sh) || sesc) { ok = false; break; }
                            t = T_STRING;
                            if (!string_id(A, c.s, sb, sn, sh, a0 + sb, v)) { ok = false; break; }
                        } else if (v0 == '{' || v0 == '[') {
                            ok = false;  // (canonical text of arrays / objects: the host)
                            break;
                        } else if (v0 == 't') {
                            if (!scan_literal(c, "true", 4)) { ok = false; break; }
                            t = T_TRUE;
                        } else if (v0 == 'f') {
                            if (!scan_literal(c, "false", 5)) { ok = false; break; }
                            t = T_FALSE;
                        } else if (v0 == 'n') {
                            if (!scan_literal(c, "null", 4)) { ok = false; break; }
                            t = T_NULL;
                        } else {
                            if (scan_number(c, t, v) != 1) { ok = false; break; }
                        }
#pragma unroll
                        for (int i = 0; i < kMaxCols; i++)
                            if ((leafs >> i) & 1u) {
                                tg[i] = t;
                                pv[i] = v;
                            }
                        // (paths that go on below a scalar stay MISSING: a field of a non-object, value/parsed.go:159-163)
                    } else if (deeper && v0 == '{') {
                        c.p++;
                        c.ws();
                        if (c.peek() == '}') c.p++;  // an empty object: nothing below it
                        else {
                            level++;
                            active[level] = deeper;
                            found[level] = 0;
                            descended = true;
                        }
                    } else if (!skip_value(c)) {
                        ok = false;
                        break;
                    }
                    if (descended) continue;
                    // behind a value: the next member, or the end of this object (and of the ones it completes)
                    for (;;) {
                        c.ws();
                        const int x = c.peek();
                        if (x == ',') {
                            c.p++;
                            break;
                        }
                        if (x == '}') {
                            c.p++;
                            level--;
                            if (level < 0) break;
                            continue;
                        }
                        ok = false;
                        break;
                    }
                }
                if (ok) {
                    c.ws();
                    if (c.p != c.e) ok = false;  // trailing bytes
                }
                A.status[d] = ok ? 0 : 1;
                if (ok) {
#pragma unroll
                    for (int i = 0; i < kMaxCols; i++)
                        if ((uint32_t)i < A.npaths) {
                            A.out_tags[i][d] = (uint8_t)tg[i];
                            A.out_payload[i][d] = pv[i];
                        }
                }
            }
            __builtin_amdgcn_wave_barrier();
            first += cnt;
        }
    }
}

// provisional string ids -> dictionary codes (the host interned the batch's new strings and uploaded code_of[slot])
__global__ void json_remap_kernel(const JsonDevArgs A, const uint32_t* code_of) {
    const uint64_t d = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= A.ndocs || A.status[d]) return;
    for (uint32_t i = 0; i < A.npaths; i++)
        if (A.out_tags[i][d] == T_STRING) A.out_payload[i][d] = code_of[A.out_payload[i][d]];
}

hipError_t launch_json_extract(const JsonDevArgs& A, uint32_t num_cus, hipStream_t st) {
    if (A.ndocs == 0) return hipSuccess;
    const size_t shmem = (size_t)(kJsonBlock / 64) * kJsonWaveBytes;
    auto k = json_extract_kernel;
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    const uint64_t waves = (A.ndocs + 63) / 64;
    const uint32_t grid = (uint32_t)std::min<uint64_t>((waves + 3) / 4, (uint64_t)num_cus * 2);
    hipLaunchKernelGGL(k, dim3(grid), dim3(kJsonBlock), shmem, st, A);
    return hipGetLastError();
}

// what the host needs of the batch's new strings: where their first occurrence lies
__global__ void json_gather_first_kernel(const JsonDevArgs A, uint64_t n, unsigned long long* out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = A.tab_first[A.new_list[i]] & ~(1ull << 63);
}
__global__ void json_scatter_codes_kernel(const uint32_t* new_list, const uint32_t* codes, uint64_t n, uint32_t* code_of) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) code_of[new_list[i]] = codes[i];
}
// the documents the host extracted itself: docs[i]'s values of every path
__global__ void json_patch_kernel(const JsonDevArgs A, const uint64_t* docs, const uint8_t* tags, const uint64_t* payload, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (uint32_t c = 0; c < A.npaths; c++) {
        A.out_tags[c][docs[i]] = tags[i * A.npaths + c];
        A.out_payload[c][docs[i]] = payload[i * A.npaths + c];
    }
}
hipError_t launch_json_gather_first(const JsonDevArgs& A, uint64_t n, unsigned long long* out, hipStream_t st) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(json_gather_first_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, A, n, out);
    return hipGetLastError();
}
hipError_t launch_json_scatter_codes(const uint32_t* new_list, const uint32_t* codes, uint64_t n, uint32_t* code_of, hipStream_t st) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(json_scatter_codes_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, new_list, codes, n, code_of);
    return hipGetLastError();
}
hipError_t launch_json_patch(const JsonDevArgs& A, const uint64_t* docs, const uint8_t* tags, const uint64_t* payload, uint64_t n, hipStream_t st) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(json_patch_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, A, docs, tags, payload, n);
    return hipGetLastError();
}

hipError_t launch_json_remap(const JsonDevArgs& A, const uint32_t* code_of, hipStream_t st) {
    if (A.ndocs == 0) return hipSuccess;
    hipLaunchKernelGGL(json_remap_kernel, dim3((uint32_t)((A.ndocs + 255) / 256)), dim3(256), 0, st, A, code_of);
    return hipGetLastError();
}

}  // namespace n1k
