// n1k_bins.hip — the back end of the partitioned GROUP BY over 16-byte records (Rec16, n1k_tables.h).
//
// The plan-specialised scan (n1k_spec.h, records mode) leaves every surviving row as (packed key, operand) in one of
// 256 hash regions.  Here: the second partition pass (radix_scatter16_kernel: regions -> bins of fixed capacity) and
// InitialGroup bin by bin in an LDS table (agg_bins16_kernel), whose groups leave as one compact region of partial
// groups with unique keys — what FinalGroup (finalize_region_kernel), the merge into the global table and the multi-GPU
// exchange take.  Reference: execution/group_initial.go:56-100 (one map per operator copy; here one LDS table per bin).
#include <hip/hip_runtime.h>

#include "n1k_kernels.h"
#include "n1k_spec.h"

namespace n1k {

namespace {
constexpr int kBlock = 512;
}  // namespace

// The second partition pass, of 16-byte records (the partitioned GROUP BY) or 8-byte member words (COUNT(DISTINCT)).
// Input: 256 hash regions x 8 sub-regions of fixed capacity, written by the plan-specialised scan (n1k_spec.h;
// sub-region i holds seg_count[i * kCursorStride] elements at src[i * seg_stride ..]).  Output: `nb` bins per region,
// bin (r, b) = dst[(r * nb + b) * bin_cap ..], its count in cursor[r * nb + b] — the keys were spread by mix64, so a bin
// holds its share unless few keys own most rows, in which case *overflow is set and the engine takes the exact path.
// `wpr` workgroups share a region: they take its tiles (of one sub-region each) in turn.  They append to the same nb
// bin tails, so they get ids that are equal mod 8 — workgroups are dealt round-robin over the 8 XCDs, whose L2s are
// private: one L2 sees all the partial lines of a tail and merges them before they leave for HBM.  Speed only; any
// placement is correct.
namespace {
__device__ __forceinline__ uint64_t elem_key(uint64_t w) { return w; }
__device__ __forceinline__ uint64_t elem_key(const Rec16& r) { return r.k & ~kRecIntFlag; }
__device__ __forceinline__ bool elem_none(uint64_t w) { return w == kEmptyKey; }
__device__ __forceinline__ bool elem_none(const Rec16& r) { return r.k == kEmptyKey; }
__device__ __forceinline__ void elem_load(const uint64_t* p, uint64_t& e) { e = __builtin_nontemporal_load(p); }
__device__ __forceinline__ void elem_load(const Rec16* p, Rec16& e) {
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    const u64x2 v = __builtin_nontemporal_load((const u64x2*)p);
    e.k = v.x;
    e.v = v.y;
}
__device__ __forceinline__ void elem_clear(uint64_t& e) { e = kEmptyKey; }
__device__ __forceinline__ void elem_clear(Rec16& e) { e.k = kEmptyKey; e.v = 0; }
}  // namespace

template <class E, int PER>
__global__ __launch_bounds__(kBlock) void radix_scatter_sub_kernel(const RadixArgs A, uint32_t wpr, uint32_t bin_mask) {
    extern __shared__ uint64_t dyn16[];
    ScatterLds<E, kBlock, PER>& S = *(ScatterLds<E, kBlock, PER>*)dyn16;
    constexpr uint32_t kTile = kBlock * PER;
    const uint32_t tid = threadIdx.x, nb = bin_mask + 1u, shift = A.shift;
    // region and turn of this workgroup: ids equal mod 8 share a region (nseg / kRecSubs regions, a multiple of 8)
    const uint32_t id = blockIdx.x, lane = id & 7u, rest = id >> 3;
    const uint32_t w = rest % wpr, r = (rest / wpr) * 8u + lane;
    scatter_init<kBlock>(S.cnt);
    __syncthreads();
    const E* src = (const E*)A.src;
    E* dst = (E*)A.dst + (size_t)r * nb * A.bin_cap;
    unsigned long long* cursor = A.cursor + (size_t)r * nb;
    uint32_t* const flag = A.overflow;
    uint32_t par = 0, turn = 0;
    for (uint32_t x = 0; x < kRecSubs; x++) {
        const uint32_t seg = r * kRecSubs + x;
        const uint64_t c0 = A.seg_count[(size_t)seg * kCursorStride];
        const uint64_t s0 = (uint64_t)seg * A.seg_stride, s1 = s0 + (c0 < A.seg_stride ? c0 : A.seg_stride);
        for (uint64_t tile = s0; tile < s1; tile += kTile, turn++) {
            if (turn % wpr != w) continue;
            const uint32_t n = (uint32_t)(s1 - tile < (uint64_t)kTile ? s1 - tile : (uint64_t)kTile);
            E e[PER];
            uint32_t bins[PER];
#pragma unroll
            for (int j = 0; j < PER; j++) {
                const uint32_t p = (uint32_t)j * kBlock + tid;
                elem_clear(e[j]);
                if (p < n) elem_load(src + tile + p, e[j]);
            }
#pragma unroll
            for (int j = 0; j < PER; j++) bins[j] = elem_none(e[j]) ? kScatterNone : (radix_bin(elem_key(e[j]), shift) & bin_mask);
            scatter_tile<kBlock, PER>(S, par, e, bins, cursor, 1u, dst, A.bin_cap, A.bin_cap,
                                      [=](uint32_t, const E&) { *(volatile uint32_t*)flag = 1u; });
            par ^= 1u;
        }
    }
}

// `A.nseg` = sub-regions (256 regions x kRecSubs); `wpr` workgroups per region; `bins_per_region` (a power of two <= 256)
template <class E, int PER>
static hipError_t launch_scatter_sub(const RadixArgs& A, uint32_t wpr, uint32_t bins_per_region, hipStream_t st) {
    const uint32_t nreg = A.nseg / kRecSubs;
    (void)hipMemsetAsync(A.cursor, 0, (size_t)nreg * bins_per_region * sizeof(unsigned long long), st);
    auto k = radix_scatter_sub_kernel<E, PER>;
    const size_t shmem = sizeof(ScatterLds<E, kBlock, PER>);
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    hipLaunchKernelGGL(k, dim3(wpr * nreg), dim3(kBlock), shmem, st, A, wpr, bins_per_region - 1u);
    return hipGetLastError();
}
hipError_t launch_radix_scatter16(const RadixArgs& A, uint32_t wpr, uint32_t bins_per_region, hipStream_t st) {
    return launch_scatter_sub<Rec16, 8>(A, wpr, bins_per_region, st);  // tiles of 4096 records = 64 KB staged
}
hipError_t launch_radix_scatter_words(const RadixArgs& A, uint32_t wpr, uint32_t bins_per_region, hipStream_t st) {
    return launch_scatter_sub<uint64_t, 16>(A, wpr, bins_per_region, st);  // tiles of 8192 words = 64 KB staged
}

namespace {

// one LDS slot back to "empty": key and the accumulators' identities (what lds_table_init writes for every slot)
__device__ __forceinline__ void lds_slot_reset(const Program& P, uint64_t* lds, uint32_t S, uint32_t s) {
    lds[s] = kEmptyKey;
    for (uint32_t a = 0; a < P.naggs; a++) {
        const AggSpec& ag = P.aggs[a];
        uint64_t* w = lds + (size_t)ag.lds_off * S + s;
        const uint32_t nw = (ag.kind == AGG_COUNT || ag.kind == AGG_COUNTN) ? 1u : (ag.kind == AGG_SUM ? kLdsWordsSum : (ag.kind == AGG_AVG ? kLdsWordsAvg : kWordsMinMax));
        for (uint32_t i = 0; i < nw; i++) {
            uint64_t ident = 0;
            if (ag.kind == AGG_MIN) ident = i == 1 ? (uint64_t)INT64_MAX : (i >= 2 ? ~0ull : 0ull);
            if (ag.kind == AGG_MAX) ident = i == 1 ? (uint64_t)INT64_MIN : 0ull;
            w[(size_t)i * S] = ident;
        }
    }
}

// CumulateIntermediate of one LDS slot into a FRESH global row nobody else touches: plain stores of what glob_row_init +
// merge_slot would leave there (no atomics: 6.4 M groups x 4 read-modify-writes at the memory side were a third of the
// kernel)
__device__ __forceinline__ void store_slot(const Program& P, const uint64_t* lds, uint32_t S, uint32_t slot, uint64_t* g) {
    for (uint32_t a = 0; a < P.naggs; a++) {
        const AggSpec& ag = P.aggs[a];
        const uint64_t* l = lds + (size_t)ag.lds_off * S + slot;  // word i at l[i * S]
        uint64_t* w = g + ag.glob_off;
        switch (ag.kind) {
            case AGG_COUNT:
            case AGG_COUNTN: w[0] = l[0]; break;
            case AGG_SUM:
            case AGG_AVG: {
                const uint64_t fl = l[2 * (size_t)S];
                const int64_t x = (fl & (SF_NONNEG_INT | SF_NEG_INT)) ? (int64_t)l[0] : 0;
                w[0] = (uint64_t)(uint32_t)x;
                w[1] = (uint64_t)(x >> 32);
                w[2] = (fl & SF_FLOAT) ? l[(size_t)S] : 0ull;
                w[3] = fl;
                if (ag.kind == AGG_AVG) w[4] = fl ? l[3 * (size_t)S] : 0ull;
                break;
            }
            default: {  // MIN / MAX: the LDS identities are the global ones
                w[0] = l[0];
                w[1] = l[(size_t)S];
                w[2] = l[2 * (size_t)S];
                w[3] = l[3 * (size_t)S];
                break;
            }
        }
    }
}

// a record outside the bin's table (table full, or an integer the narrow LDS sum does not take): a partial group of its own
__device__ __forceinline__ void emit_single(const Program& P, const BinAggArgs& A, uint64_t key, const AggSpec* only, uint32_t tag,
                                            uint64_t p) {
    const unsigned long long q = atomicAdd((unsigned long long*)&A.emit[0], 1ull);
    atomicAdd(A.emit_singletons, 1ull);  // keys in the region are no longer unique
    if (q >= A.emit_cap) {
        atomicOr(A.err_flags, (uint32_t)ERR_TABLE_FULL);
        return;
    }
    A.emit[2 + q] = key;
    uint64_t* row = A.emit + 2 + A.emit_cap + q * P.glob_words;
    glob_row_init(P, row);
    if (only) {
        acc_global(P, *only, row, tag, p);
        return;
    }
    for (uint32_t a = 0; a < P.naggs; a++) {
        const bool has = A.agg_src[a] < kRecOperands;  // (records carry one operand: slot 0)
        acc_global(P, P.aggs[a], row, has ? tag : (uint32_t)T_NULL, has ? p : 0ull);
    }
}

}  // namespace

// One workgroup per bin (persistent over bins): InitialGroup over the bin's records in an LDS table, whose groups are
// appended to the compact region ([count][0][keys: emit_cap][accumulators]).  The table is initialised once; every slot
// a bin used is reset as it is emitted.  U records per thread are loaded before any is processed.
template <int BLOCK, int U>
__global__ __launch_bounds__(BLOCK) void agg_bins16_kernel(const Program P, const BinAggArgs A) {
    extern __shared__ uint64_t lds[];
    __shared__ uint32_t lds_fill, emit_n;
    __shared__ unsigned long long emit_base;
    const uint32_t S = A.lds_slots, tid = threadIdx.x;
    const Rec16* const rec = (const Rec16*)A.rec;
    const uint32_t cstride = A.bin_count_stride ? A.bin_count_stride : 1u;
    lds_table_init<BLOCK>(P, lds, S, tid);
    if (tid == 0) {
        lds_fill = 0;
        emit_n = 0;
    }
    __syncthreads();
    // The workgroup's chunks (BLOCK x U records of one bin) in order, the next one always requested before the current one
    // is processed — across the end of a bin too, so that loads stay in flight through the barriers and the emit step.
    struct Chunk {
        uint32_t bin;
        uint64_t base, hi;
    };
    auto enter = [&](Chunk& c) {  // first chunk of bin c.bin or of the next non-empty bin of this workgroup
        while (c.bin < A.nbins) {
            const uint64_t n = A.bin_count[(size_t)c.bin * cstride];
            c.base = (uint64_t)c.bin * A.bin_stride;
            c.hi = c.base + (n < A.bin_stride ? n : A.bin_stride);
            if (c.base < c.hi) return;
            c.bin += gridDim.x;
        }
    };
    auto load = [&](const Chunk& c, Rec16 (&r)[U]) {
#pragma unroll
        for (int j = 0; j < U; j++) {
            const uint64_t i = c.base + (uint64_t)j * BLOCK + tid;
            r[j].k = kEmptyKey;
            r[j].v = 0;
            if (c.bin < A.nbins && i < c.hi) {
                typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
                const u64x2 v = __builtin_nontemporal_load((const u64x2*)(rec + i));
                r[j].k = v.x;
                r[j].v = v.y;
            }
        }
    };
    Chunk cur;
    cur.bin = blockIdx.x;
    cur.base = cur.hi = 0;
    enter(cur);
    Rec16 r[U], rn[U];
    load(cur, r);
    while (cur.bin < A.nbins) {
        Chunk nxt = cur;
        nxt.base += (uint64_t)BLOCK * U;
        if (nxt.base >= nxt.hi) {
            nxt.bin += gridDim.x;
            enter(nxt);
        }
        load(nxt, rn);
#pragma unroll
        for (int j = 0; j < U; j++) {
            if (r[j].k == kEmptyKey) continue;  // beyond the bin
            uint64_t key, p;
            uint32_t t;
            rec16_decode(r[j], key, t, p);
            const int slot = lds_find_or_insert(lds, S, key, &lds_fill, A.lds_max_fill);
            if (slot < 0) {  // more groups in the bin than the LDS table takes
                emit_single(P, A, key, nullptr, t, p);
                continue;
            }
            for (uint32_t a = 0; a < P.naggs; a++) {
                const bool has = A.agg_src[a] < kRecOperands;
                if (!acc_lds(P, P.aggs[a], lds, S, (uint32_t)slot, has ? t : (uint32_t)T_NULL, has ? p : 0ull))
                    emit_single(P, A, key, &P.aggs[a], t, p);  // |int| >= 2^40: leaves with just this contribution
            }
        }
        if (nxt.bin != cur.bin) {
            // end of the bin: its groups join the compact region, every slot they used is reset
            lds_barrier();  // (LDS only: the next chunk's loads stay in flight)
            if (tid == 0) {
                const uint32_t n = lds_fill;
                emit_base = n ? atomicAdd((unsigned long long*)&A.emit[0], (unsigned long long)n) : 0ull;
                lds_fill = 0;
                emit_n = 0;
            }
            lds_barrier();  // (LDS only: the next chunk's loads stay in flight)
            const unsigned long long q0 = emit_base;
            for (uint32_t s = tid; s < S; s += BLOCK) {
                const uint64_t key = lds[s];
                if (key == kEmptyKey) continue;
                const unsigned long long q = q0 + atomicAdd(&emit_n, 1u);
                if (q < A.emit_cap) {
                    A.emit[2 + q] = key;
                    store_slot(P, lds, S, s, A.emit + 2 + A.emit_cap + q * P.glob_words);
                } else
                    atomicOr(A.err_flags, (uint32_t)ERR_TABLE_FULL);
                lds_slot_reset(P, lds, S, s);
            }
            lds_barrier();  // (LDS only: the next chunk's loads stay in flight)
        }
        cur = nxt;
#pragma unroll
        for (int j = 0; j < U; j++) r[j] = rn[j];
    }
}

hipError_t launch_agg_bins16(const Program& P, const BinAggArgs& A, uint32_t grid, uint32_t block, uint32_t per_thread, hipStream_t st) {
    const size_t shmem = (size_t)A.lds_slots * P.lds_words * 8;
#define N1K_AGG16(BB, UU)                                                                                         \
    {                                                                                                             \
        auto k = agg_bins16_kernel<BB, UU>;                                                                       \
        if (shmem > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); \
        hipLaunchKernelGGL(k, dim3(grid), dim3(BB), shmem, st, P, A);                                              \
    }
    // (8 records in flight per thread spilled to scratch)
    if (block <= 256) {
        if (per_thread <= 2) N1K_AGG16(256, 2) else N1K_AGG16(256, 4)
    } else {
        if (per_thread <= 2) N1K_AGG16(512, 2) else N1K_AGG16(512, 4)
    }
#undef N1K_AGG16
    return hipGetLastError();
}

}  // namespace n1k
