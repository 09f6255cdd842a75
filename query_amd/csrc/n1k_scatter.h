// n1k_scatter.h — one tile of a radix partition through LDS, of 16-byte records (Rec16: the partitioned GROUP BY) or
// 8-byte member words (COUNT(DISTINCT)).
//
// Shared by the plan-specialised scan (n1k_spec.h: the first partition pass of both) and by the second partition pass
// (n1k_bins.hip).  A workgroup ranks the tile's records per bin with LDS counters,
// reserves the bins' output ranges with ONE global atomic per bin and tile, stages the records in LDS in bin order and
// writes them out in runs, so that a bin's records leave as consecutive 16-byte stores.  Plain header: it is compiled at
// run time too (n1k_jit.cpp).
#pragma once
#include "n1k_tables.h"

namespace n1k {

constexpr uint32_t kScatterNone = 0xFFFFFFFFu;  // "no record" in the bins array of a tile

template <class E, int BLOCK, int PER>
struct ScatterLds {
    E stage[BLOCK * PER];
    uint32_t cnt[2][256];  // records per bin in this tile (two copies: the next tile's is zeroed while this one is read)
    uint32_t pre[256];     // first staged position of the bin
    unsigned long long gbase[256];  // first output position of the bin's run
    uint8_t sbin[BLOCK * PER];      // bin of every staged element (cheaper than hashing it again on the way out)
};

// Workgroup barrier for data exchanged through LDS only: LDS traffic drained, then s_barrier.  __syncthreads() also waits
// for every global load and store of the wave (s_waitcnt vmcnt(0)): the previous tile's stores, the next tile's loads
// and the bins' cursor atomics would all be drained at each of a tile's barriers instead of staying in flight.
N1K_DEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int BLOCK>
N1K_DEV void scatter_init(uint32_t (&cnt)[2][256]) {
    for (uint32_t i = threadIdx.x; i < 512; i += BLOCK) (&cnt[0][0])[i] = 0;
}

// All threads of the workgroup call it once per tile, `par` alternating 0 / 1 (scatter_init + a barrier before the first).
//   w[j], b[j]  this thread's records and their bins (< 256; kScatterNone = no record)
//   cursor      bin i's output cursor at cursor[i * cstride]  (counts from zero)
//   dst         bin i's output at dst[i * bin_stride ..], `bin_cap` elements at most: the rest goes to over(bin, element)
template <int BLOCK, int PER, class E, class Over>
N1K_DEV void scatter_tile(ScatterLds<E, BLOCK, PER>& S, uint32_t par, const E (&w)[PER], const uint32_t (&b)[PER],
                          unsigned long long* cursor, uint32_t cstride, E* dst, uint64_t bin_stride, uint64_t bin_cap,
                          Over over) {
    static_assert(BLOCK >= 256 && BLOCK * PER <= 65536, "one thread per bin; ranks are 16 bits");
    const uint32_t tid = threadIdx.x;
    uint32_t rk[PER];
#pragma unroll
    for (int j = 0; j < PER; j++) rk[j] = b[j] == kScatterNone ? kScatterNone : ((b[j] << 16) | atomicAdd(&S.cnt[par][b[j]], 1u));
    lds_barrier();
    // wave 0: four bins per lane — the bins' output ranges (one returning global atomic each, awaited only after the
    // staging), their staged positions (wave scan), and the zeroing of the next tile's counters
    unsigned long long base[4] = {0, 0, 0, 0};
    if (tid < 64) {
        uint32_t c[4], run = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            c[k] = S.cnt[par][tid * 4 + k];
            if (c[k]) base[k] = atomicAdd(&cursor[(size_t)(tid * 4 + k) * cstride], (unsigned long long)c[k]);
            run += c[k];
        }
        uint32_t incl = run;
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t t = __shfl_up(incl, off, 64);
            if ((int)tid >= off) incl += t;
        }
        uint32_t at = incl - run;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            S.pre[tid * 4 + k] = at;
            at += c[k];
            S.cnt[par ^ 1u][tid * 4 + k] = 0;
        }
    }
    lds_barrier();
#pragma unroll
    for (int j = 0; j < PER; j++)
        if (rk[j] != kScatterNone) {
            const uint32_t at = S.pre[rk[j] >> 16] + (rk[j] & 0xFFFFu);
            S.stage[at] = w[j];
            S.sbin[at] = (uint8_t)(rk[j] >> 16);
        }
    if (tid < 64) {
#pragma unroll
        for (int k = 0; k < 4; k++) S.gbase[tid * 4 + k] = base[k];
    }
    lds_barrier();
    const uint32_t staged = S.pre[255] + S.cnt[par][255];
    for (uint32_t p = tid; p < staged; p += BLOCK) {
        const E x = S.stage[p];
        const uint32_t bin = S.sbin[p];
        const unsigned long long pos = S.gbase[bin] + (p - S.pre[bin]);
        if (pos < bin_cap) dst[(size_t)bin * bin_stride + pos] = x;
        else over(bin, x);
    }
    // (no barrier here: the next tile touches stage / pre / gbase only behind its own barriers, and counts in cnt[par ^ 1])
}

}  // namespace n1k
