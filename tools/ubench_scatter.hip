// Calibration (not product code): ONE-pass scatter of 16-byte records (or 8-byte words) into thousands of bins on gfx950.
// Question: can the first partition pass of the partitioned GROUP BY / COUNT(DISTINCT) write straight into bins that fit an
// LDS table (4096 .. 16384 bins), i.e. without a second pass over the records?
//   mode 0: one AGENT-scope returning atomic per record on cursor[bin] (memory side: the known ~20 G/s wall)
//   mode 1: sub-bin = the XCD the workgroup runs on (HW_REG_XCC_ID), one WORKGROUP-scope atomic per record on
//           cursor[xcd][bin] — the cursor is only ever touched from one XCD, whose L2 executes the atomic
//   mode 2: mode 1 + the wave first combines lanes with equal bins (match loop) — fewer atomics when bins repeat
//   mode 3: tile sort in LDS by (bin >> s) ... not here
// Output check: sum of cursors == n and every record sits in the bin its key hashes to (sampled).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); return 1;}}while(0)
__host__ __device__ inline uint64_t mix(uint64_t x){uint64_t z=x+0x9E3779B97F4A7C15ull; z=(z^(z>>30))*0xBF58476D1CE4E5B9ull; z=(z^(z>>27))*0x94D049BB133111EBull; return z^(z>>31);}

struct Rec16 { uint64_t k, v; };

__device__ inline uint32_t xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & 7u; }  // HW_REG_XCC_ID[3:0]

template <int MODE, class E>
__global__ __launch_bounds__(256) void k_scatter(const uint64_t* keys, uint64_t n, uint32_t nbins, uint32_t bin_shift, uint64_t cap,
                                                 unsigned long long* cursor, E* out, uint32_t* overflow, uint32_t groups) {
    const uint32_t sub = MODE == 0 ? 0u : xcc_id();
    unsigned long long* cur = cursor + (size_t)sub * nbins;
    E* dst = out + (size_t)sub * nbins * cap;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t raw = __builtin_nontemporal_load(keys + i);
        const uint64_t key = groups ? raw % groups : raw;
        const uint32_t bin = (uint32_t)(mix(key) >> bin_shift);
        unsigned long long pos;
        if (MODE == 0) pos = __hip_atomic_fetch_add(&cur[bin], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (MODE == 1) pos = __hip_atomic_fetch_add(&cur[bin], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else {
            // lanes of the wave with the same bin: one atomic for all of them
            unsigned long long todo = __ballot(1);
            pos = 0;
            const uint32_t lane = threadIdx.x & 63;
            while (todo) {
                const int leader = __ffsll((long long)todo) - 1;
                const uint32_t lb = __shfl(bin, leader, 64);
                const unsigned long long same = __ballot(bin == lb) & todo;
                unsigned long long base = 0;
                if ((int)lane == leader) base = __hip_atomic_fetch_add(&cur[lb], (unsigned long long)__popcll(same), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                base = __shfl(base, leader, 64);
                if (bin == lb) pos = base + __popcll(same & ((1ull << lane) - 1ull));
                todo &= ~same;
            }
        }
        if (pos < cap) {
            if constexpr (sizeof(E) == 16) {
                typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
                u64x2 v; v.x = key; v.y = raw ^ 0x55ull;
                *(u64x2*)&dst[(size_t)bin * cap + pos] = v;
            } else
                dst[(size_t)bin * cap + pos] = (E)key;
        } else
            *overflow = 1;
    }
}

template <class E>
int run(const char* name, uint64_t n, const uint64_t* d_keys, uint32_t groups) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    uint32_t* d_over; CK(hipMalloc(&d_over, 4));
    for (uint32_t lb : {11u, 12u, 13u, 14u}) {
        const uint32_t nbins = 1u << lb;
        for (int mode = 0; mode < 3; mode++) {
            const uint32_t subs = mode == 0 ? 1 : 8;
            const uint64_t mean = n / nbins / subs;
            const uint64_t cap = groups ? (mean * 4 + 4096) : (mean + mean / 4 + 1024);
            unsigned long long* cur; CK(hipMalloc(&cur, (size_t)subs * nbins * 8));
            E* out; CK(hipMalloc(&out, (size_t)subs * nbins * cap * sizeof(E)));
            float best = 1e9;
            for (int rep = 0; rep < 3; rep++) {
                CK(hipMemset(cur, 0, (size_t)subs * nbins * 8)); CK(hipMemset(d_over, 0, 4)); CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                const dim3 g(256 * 8), b(256);
                if (mode == 0) hipLaunchKernelGGL((k_scatter<0, E>), g, b, 0, 0, d_keys, n, nbins, 64 - lb, cap, cur, out, d_over, groups);
                if (mode == 1) hipLaunchKernelGGL((k_scatter<1, E>), g, b, 0, 0, d_keys, n, nbins, 64 - lb, cap, cur, out, d_over, groups);
                if (mode == 2) hipLaunchKernelGGL((k_scatter<2, E>), g, b, 0, 0, d_keys, n, nbins, 64 - lb, cap, cur, out, d_over, groups);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            std::vector<unsigned long long> hc((size_t)subs * nbins);
            CK(hipMemcpy(hc.data(), cur, hc.size() * 8, hipMemcpyDeviceToHost));
            uint32_t over = 0; CK(hipMemcpy(&over, d_over, 4, hipMemcpyDeviceToHost));
            unsigned long long total = 0, mx = 0, submax[8] = {0};
            for (size_t i = 0; i < hc.size(); i++) { total += hc[i]; if (hc[i] > mx) mx = hc[i]; submax[i / nbins] += hc[i]; }
            // sampled content check: the first records of a few bins hash to their bin
            bool ok = total == n || over;
            for (uint32_t s = 0; s < subs && ok; s++)
                for (uint32_t bin : {0u, nbins / 3, nbins - 1}) {
                    const unsigned long long c = hc[(size_t)s * nbins + bin] < cap ? hc[(size_t)s * nbins + bin] : cap;
                    const size_t take = c < 64 ? c : 64;
                    std::vector<E> h(take ? take : 1);
                    if (take) CK(hipMemcpy(h.data(), out + ((size_t)s * nbins + bin) * cap, take * sizeof(E), hipMemcpyDeviceToHost));
                    for (size_t i = 0; i < take; i++) {
                        uint64_t key; memcpy(&key, &h[i], 8);
                        if ((uint32_t)(mix(key) >> (64 - lb)) != bin) ok = false;
                    }
                }
            printf("%s n=%llu groups=%u bins=%5u mode=%d : %7.3f ms  %6.1f G rec/s  total=%llu max/bin=%llu over=%u subs=[%llu %llu %llu %llu %llu %llu %llu %llu] %s\n", name,
                   (unsigned long long)n, groups, nbins, mode, best, n / best / 1e6, total, mx, over, submax[0], submax[1], submax[2], submax[3], submax[4], submax[5],
                   submax[6], submax[7], ok ? "OK" : "WRONG");
            CK(hipFree(cur)); CK(hipFree(out));
        }
    }
    return 0;
}

__global__ void k_fill(uint64_t* keys, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) keys[i] = mix(i * 7919ull + 13ull);
}

int main(int argc, char** argv) {
    const uint64_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 100000000ull;
    uint64_t* d_keys; CK(hipMalloc(&d_keys, n * 8));
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, d_keys, n);
    CK(hipDeviceSynchronize());
    if (run<Rec16>("rec16 unique", n, d_keys, 0)) return 1;
    if (run<Rec16>("rec16 6.4M groups", n, d_keys, 6400000u)) return 1;
    if (run<uint64_t>("word8 unique", n, d_keys, 0)) return 1;
    return 0;
}
