// n1k_bins.hip — the back end of the partitioned GROUP BY over 16-byte records (Rec16, n1k_tables.h).
//
// The plan-specialised scan (n1k_spec.h, records mode) leaves every surviving row as (packed key, operand) in one of
// 256 hash regions.  Here: the second partition pass (radix_scatter16_kernel: regions -> bins of fixed capacity) and
// InitialGroup bin by bin in an LDS table (agg_bins16_kernel), whose groups leave as one compact region of partial
// groups with unique keys — what FinalGroup (finalize_region_kernel), the merge into the global table and the multi-GPU
// exchange take.  Reference: execution/group_initial.go:56-100 (one map per operator copy; here one LDS table per bin).
#include <hip/hip_runtime.h>

#include <algorithm>

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

// slot of a key in a bin's table: one multiply and the high half of another (lds_hash takes four: at ~ 100 vector instructions
// per record — what bounds the kernel — they were a tenth of them).  The keys of a bin share the bits of part_hash the two
// partition passes took; this is another function of the key.
__device__ __forceinline__ uint32_t bin_slot(uint64_t key, uint32_t S) {
    const uint32_t hi = (uint32_t)(key >> 32);
    const uint32_t x = ((uint32_t)key ^ ((hi << 13) | (hi >> 19))) * 0x9E3779B1u;
    return __umulhi(x ^ (x >> 15), S);
}

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
// a bin used is reset as it is emitted.
//
// The next chunk (BLOCK x U records) is requested before the current one is processed, across the ends of bins too.  Every
// load is issued unconditionally — beyond its bin a thread re-reads the first record and drops it — and what was loaded is
// not looked at until the copy at the head of the loop: with predicated loads, or a first chunk loaded straight into the
// registers the loop works on, the compiler waits for everything outstanding (s_waitcnt vmcnt(0)) — the prefetch it has
// just issued included — before the first record of every chunk.  The record counts of the workgroup's bins sit in LDS
// before the loop (read from global memory when the walk reached a bin, each was one exposed memory latency).
// Measured at 100 M records in 32 Ki bins: 0.57 ms, of which 0.285 ms with the records loaded but not looked at; a third
// chunk in flight (three buffers, the loop unrolled three times) did not help (0.62 ms).
// KIND >= 0: the plan has ONE aggregate and this is its kind — the accumulate step is then straight-line code.  The thread's
// U records walk the table TOGETHER (one LDS wait per round of probes).  Records that cannot enter the table (it is full; an
// integer the narrow LDS sum does not take) are rare: they are only marked, and leave as partial groups of their own from
// ONE copy of that code.
constexpr uint32_t kAggOwnBins = 256;  // bins per workgroup of agg_bins16_kernel, at most

template <int BLOCK, int U, int KIND>
__global__ __launch_bounds__(BLOCK) void agg_bins16_kernel(const Program P, const BinAggArgs A) {
    extern __shared__ uint64_t lds[];
    __shared__ unsigned long long own_count[kAggOwnBins];  // (the launch gives every workgroup at most that many bins)
    __shared__ uint32_t lds_fill;
    __shared__ unsigned long long emit_base;
    const uint32_t S = A.lds_slots, tid = threadIdx.x;
    uint16_t* const slot_list = (uint16_t*)(lds + (size_t)S * P.lds_words);  // the slots the bin's groups took, in the order they came
    const Rec16* const rec = (const Rec16*)A.rec;
    const uint32_t cstride = A.bin_count_stride ? A.bin_count_stride : 1u;
    const uint32_t own = blockIdx.x < A.nbins ? (A.nbins - blockIdx.x + gridDim.x - 1) / gridDim.x : 0u;  // bins blockIdx.x + i * gridDim.x
    for (uint32_t i = tid; i < own && i < kAggOwnBins; i += BLOCK) own_count[i] = A.bin_count[(size_t)(blockIdx.x + i * gridDim.x) * cstride];
    lds_table_init<BLOCK>(P, lds, S, tid);
    if (tid == 0) lds_fill = 0;
    __syncthreads();
    struct Chunk {
        uint32_t ord;  // ordinal among the workgroup's bins (>= own: past the end)
        uint64_t base, hi;
    };
    auto enter = [&](Chunk& c) {  // first chunk of bin c.ord or of the next non-empty bin of this workgroup
        while (c.ord < own) {
            const uint32_t bin = blockIdx.x + c.ord * gridDim.x;
            const uint64_t n = own_count[c.ord];
            c.base = (uint64_t)bin * A.bin_stride;
            c.hi = c.base + (n < A.bin_stride ? n : A.bin_stride);
            if (c.base < c.hi) return;
            c.ord++;
        }
    };
    auto after = [&](Chunk c) {  // the chunk behind c
        if (c.ord >= own) return c;
        c.base += (uint64_t)BLOCK * U;
        if (c.base >= c.hi) {
            c.ord++;
            enter(c);
        }
        return c;
    };
    // (returns which of the U records are the chunk's: what was loaded is not looked at here — any use is a wait)
    auto load = [&](const Chunk& c, Rec16 (&r)[U]) -> uint32_t {
        typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
        uint32_t mask = 0;
#pragma unroll
        for (int j = 0; j < U; j++) {
            const uint64_t i = c.base + (uint64_t)j * BLOCK + tid;
            const bool in = c.ord < own && i < c.hi;
            // (issued by hand: the compiler does not know that the registers are in flight, so it places no wait of its own —
            //  its waits drained both chunks in flight at the head of every turn of the loop; arrive() below is the wait)
            const u64x2 v = __builtin_nontemporal_load((const u64x2*)(rec + (in ? i : 0)));
            r[j].k = v.x;
            r[j].v = v.y;
            mask |= in ? 1u << j : 0u;
        }
        return mask;
    };
    AggSpec one = P.aggs[0];  // (KIND >= 0: the plan's only aggregate, its kind a compile-time constant)
    if (KIND >= 0) one.kind = (uint32_t)KIND;
    constexpr bool kSplit = KIND == (int)AGG_COUNT || KIND == (int)AGG_COUNTN || KIND == (int)AGG_SUM || KIND == (int)AGG_AVG;
    const bool one_has = A.agg_src[0] < kRecOperands;
    constexpr uint32_t kNoSlot = 1u << 31;
    lds_u32* const fillp = (lds_u32*)&lds_fill;

    // one chunk: its records into the table; `last`: the bin ends with it
    auto process = [&](const Rec16 (&r)[U], uint32_t mask, bool last) {
        uint64_t key[U], p[U];
        uint32_t t[U], hs[U], out[U];  // out: 0, kNoSlot, or the aggregates (bit a) whose contribution leaves on its own
        uint32_t live = 0;             // bit j: record j has not found its slot yet
        // (decode and hash without a branch: the U records' arithmetic interleaves, and a record beyond the bin just is not live)
#pragma unroll
        for (int j = 0; j < U; j++) {
            out[j] = 0;
            const bool is_int = (r[j].k >> 63) != 0, boxed = !is_int && (r[j].v >> 48) == 0xFFF8ull;  // (rec16_decode, n1k_tables.h)
            key[j] = r[j].k & ~kRecIntFlag;
            t[j] = is_int ? (uint32_t)T_INT : (boxed ? (uint32_t)(r[j].v >> 40) & 0xFFu : (uint32_t)T_FLOAT);
            p[j] = boxed ? (r[j].v & 0xFFFFFFFFFFull) : r[j].v;
            hs[j] = bin_slot(key[j], S);
            const bool is_live = (mask >> j & 1u) && r[j].k != kEmptyKey && !(A.pad1 & 1u);  // beyond the bin / padding  (pad1: timing experiments only)
            live |= is_live ? 1u << j : 0u;
        }
        const uint32_t valid = live;
        uint32_t found = live;  // bit j: hs[j] is record j's slot (cleared again where the table has no room)
#pragma unroll 1
        for (int probe = 0; probe < 32 && live; probe++) {
            unsigned long long seen[U];
#pragma unroll
            for (int j = 0; j < U; j++) seen[j] = (live >> j & 1u) ? lds_peek(lds_word(lds, hs[j])) : 0ull;
#pragma unroll
            for (int j = 0; j < U; j++) {
                if (!(live >> j & 1u)) continue;
                if (seen[j] == key[j]) {
                    live &= ~(1u << j);
                    continue;
                }
                if (seen[j] == kEmptyKey) {
                    if (*(volatile lds_u32*)fillp >= A.lds_max_fill) {  // the table is kept sparse: the record leaves on its own
                        live &= ~(1u << j);
                        found &= ~(1u << j);
                        continue;
                    }
                    unsigned long long expected = kEmptyKey;
                    if (__hip_atomic_compare_exchange_strong(lds_word(lds, hs[j]), &expected, (unsigned long long)key[j], __ATOMIC_RELAXED,
                                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                        // a new group: its slot joins the list the emit step walks (no scan of the whole table per bin)
                        const uint32_t at = __hip_atomic_fetch_add(fillp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        slot_list[at] = (uint16_t)hs[j];
                        live &= ~(1u << j);
                        continue;
                    }
                    if (expected == key[j]) {
                        live &= ~(1u << j);
                        continue;
                    }
                }
                hs[j] = hs[j] + 1 == S ? 0 : hs[j] + 1;
            }
        }
        found &= ~live;  // (32 slots in a row taken by other keys)
        uint32_t any_out = 0;
        if (kSplit) {
            uint32_t bit[U];
#pragma unroll
            for (int j = 0; j < U; j++) {
                bit[j] = 0;
                if (!(valid >> j & 1u)) continue;
                if (!(found >> j & 1u)) { out[j] = kNoSlot; continue; }
                bit[j] = acc_lds_value(one, lds, S, hs[j], one_has ? t[j] : (uint32_t)T_NULL, one_has ? p[j] : 0ull);
                if (bit[j] == ~0u) { out[j] = 1u; bit[j] = 0; }
            }
            if (KIND == (int)AGG_SUM || KIND == (int)AGG_AVG) {
                // the flags: a fire-and-forget OR per record (as the scan kernels do it) — reading them first to skip the OR
                // where the bit is set already is a returning LDS operation, i.e. a wait, per chunk
#pragma unroll
                for (int j = 0; j < U; j++)
                    if (bit[j]) lds_or_u64(lds_word(lds, (one.lds_off + 2) * S + hs[j]), (unsigned long long)bit[j]);
            }
#pragma unroll
            for (int j = 0; j < U; j++) any_out |= out[j];
        } else {
#pragma unroll
            for (int j = 0; j < U; j++) {
                if (!(valid >> j & 1u)) continue;
                if (!(found >> j & 1u)) {
                    out[j] = kNoSlot;
                } else if (KIND >= 0) {
                    if (!acc_lds(P, one, lds, S, hs[j], one_has ? t[j] : (uint32_t)T_NULL, one_has ? p[j] : 0ull)) out[j] = 1u;
                } else {
                    for (uint32_t a = 0; a < P.naggs; a++) {
                        const bool has = A.agg_src[a] < kRecOperands;
                        if (!acc_lds(P, P.aggs[a], lds, S, hs[j], has ? t[j] : (uint32_t)T_NULL, has ? p[j] : 0ull))
                            out[j] |= 1u << a;  // |int| >= 2^40: leaves with just this contribution
                    }
                }
                any_out |= out[j];
            }
        }
        if (any_out) {
#pragma unroll 1
            for (int j = 0; j < U; j++) {
                uint64_t k1 = key[0], p1 = p[0];
                uint32_t t1 = t[0], o = out[0];
#pragma unroll
                for (int i = 1; i < U; i++)
                    if (i == j) {
                        k1 = key[i];
                        p1 = p[i];
                        t1 = t[i];
                        o = out[i];
                    }
                if (!o) continue;
                if (o & kNoSlot) {
                    emit_single(P, A, k1, nullptr, t1, p1);
                    continue;
                }
                for (uint32_t a = 0; a < P.naggs; a++)
                    if (o >> a & 1u) emit_single(P, A, k1, &P.aggs[a], t1, p1);
            }
        }
        if (last) {
            // end of the bin: its groups join the compact region, every slot they used is reset
            lds_barrier();  // (LDS only: the chunks in flight stay in flight)
            const uint32_t n = *(volatile lds_u32*)fillp;
            if (tid == 0) emit_base = n && !(A.pad1 & 2u) ? atomicAdd((unsigned long long*)&A.emit[0], (unsigned long long)n) : 0ull;
            lds_barrier();
            if (tid == 0) lds_fill = 0;  // (every thread has read n; the next chunk is processed behind the barrier below)
            const unsigned long long q0 = emit_base;
            for (uint32_t i = tid; i < n; i += BLOCK) {
                const uint32_t s = slot_list[i];
                const unsigned long long q = q0 + i;
                if (A.pad1 & 2u) {
                } else if (q < A.emit_cap) {
                    A.emit[2 + q] = lds[s];
                    store_slot(P, lds, S, s, A.emit + 2 + A.emit_cap + q * P.glob_words);
                } else
                    atomicOr(A.err_flags, (uint32_t)ERR_TABLE_FULL);
                lds_slot_reset(P, lds, S, s);
            }
            lds_barrier();
        }
    };

    Chunk cur;
    cur.ord = 0;
    cur.base = cur.hi = 0;
    enter(cur);
    Rec16 r[U], rn[U];
    uint32_t mn = load(cur, rn);
    while (cur.ord < own) {
        // The only place that waits for global loads is this copy: rn is in flight when the loop comes round, r never is.
#pragma unroll
        for (int j = 0; j < U; j++) r[j] = rn[j];
        const uint32_t m = mn;
        const Chunk nxt = after(cur);
        mn = load(nxt, rn);
        process(r, m, nxt.ord != cur.ord);
        cur = nxt;
    }
}

// the per-bin table and, behind it, the list of the slots in use (16 bits each: at most 65536 slots)
size_t agg_bins16_lds_bytes(const Program& P, uint32_t slots) { return (size_t)slots * P.lds_words * 8 + (((size_t)slots * 2 + 15) & ~(size_t)15); }

hipError_t launch_agg_bins16(const Program& P, const BinAggArgs& A, uint32_t grid, uint32_t block, uint32_t per_thread, hipStream_t st,
                             bool specialise) {
    const size_t shmem = agg_bins16_lds_bytes(P, A.lds_slots);
    grid = std::max(grid, (A.nbins + kAggOwnBins - 1) / kAggOwnBins);  // (the counts of a workgroup's bins live in LDS)
#define N1K_AGG16(BB, UU, KK)                                                                                     \
    {                                                                                                             \
        auto k = agg_bins16_kernel<BB, UU, KK>;                                                                   \
        if (shmem > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); \
        hipLaunchKernelGGL(k, dim3(grid), dim3(BB), shmem, st, P, A);                                              \
    }
    // (8 records in flight per thread spilled to scratch)
    const bool one = specialise && P.naggs == 1 && !P.aggs[0].distinct && P.aggs[0].kind <= AGG_MAX;
#define N1K_AGG16_KINDS(BB, UU)                                              \
    if (!one) N1K_AGG16(BB, UU, -1)                                          \
    else switch (P.aggs[0].kind) {  /* the plan's one aggregate fixed at compile time */ \
        case AGG_COUNT: N1K_AGG16(BB, UU, (int)AGG_COUNT) break;             \
        case AGG_COUNTN: N1K_AGG16(BB, UU, (int)AGG_COUNTN) break;           \
        case AGG_SUM: N1K_AGG16(BB, UU, (int)AGG_SUM) break;                 \
        case AGG_AVG: N1K_AGG16(BB, UU, (int)AGG_AVG) break;                 \
        case AGG_MIN: N1K_AGG16(BB, UU, (int)AGG_MIN) break;                 \
        default: N1K_AGG16(BB, UU, (int)AGG_MAX) break;                      \
    }
    if (block <= 256) {
        if (per_thread <= 2) { N1K_AGG16_KINDS(256, 2) } else { N1K_AGG16_KINDS(256, 4) }
    } else {
        if (per_thread <= 2) { N1K_AGG16_KINDS(512, 2) } else { N1K_AGG16_KINDS(512, 4) }
    }
#undef N1K_AGG16_KINDS
#undef N1K_AGG16
    return hipGetLastError();
}

}  // namespace n1k
