// n1k_distinct.cpp — the sets of the DISTINCT aggregates at finish (value/set.go:65-110, algebra/agg_count_distinct.go:84-126).
#include "n1k_engine.h"

using namespace n1k;
using namespace n1k_eng;

namespace n1k_eng {

// where the de-duplication kernel counts the new members of each group: by the packed key itself when the plan has one
// dictionary key of a small domain, in an LDS hash table while the group table is small, else per member in HBM
static void dedupe_counters(const n1k_handle* h, DedupeArgs& D) {
    D.direct_keys = 0;
    D.lds_counters = 0;
    const Program& P = h->prog;
    if (P.nkeys == 1 && P.keys[0].mode == KEYM_DICT && P.keys[0].shift == 0 && h->dict.size() + 2 <= 8192)
        D.direct_keys = (uint32_t)h->dict.size() + 2;
    else if (P.nkeys == 0)
        D.direct_keys = 1;
    else if (h->table.capacity <= 4096)
        D.lds_counters = (uint32_t)h->table.capacity;
}

// workgroups of the de-duplication kernel: as many per CU as their LDS (set + member counters) and threads allow
static uint32_t dedupe_grid(const n1k_handle* h, const DedupeArgs& D, uint32_t nbins) {
    const size_t shmem = distinct_dedupe_lds(D) + 512 + 4096;  // (+ the kernel's static arrays: the bounds of its bins)
    const uint32_t by_threads = 2048u / std::max(256u, h->opt_dedupe_block & ~3u);
    const uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(by_threads, (160 * 1024) / shmem));
    return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(nbins, (uint64_t)h->num_cus * per_cu));
}

// COUNT(DISTINCT) over the one-word members of one aggregate: radix partition of the word log until a bin's distinct
// words fit an LDS set, per-bin LDS sets, member counts added to the groups' set sizes (see n1k_kernels.hip).
n1k_status distinct_words_finish(n1k_handle* h, const AggSpec& ag, uint64_t nwords, bool hist_counted,
                                 const uint64_t* log) {
    const uint32_t set_slots = h->opt_distinct_set_slots;
    const uint64_t per_bin = std::max<uint64_t>((uint64_t)set_slots * h->opt_distinct_fill_pct / 100, 16);  // expected distinct words per final bin
    const uint32_t levels = h->opt_distinct_levels >= 0 ? (uint32_t)h->opt_distinct_levels
                                                        : (nwords <= per_bin ? 0u : (nwords <= 256 * per_bin ? 1u : 2u));
    if (!log) log = h->d_log_word[ag.log_index].p;
    const uint64_t* words = log;
    HIP_TRY(h, h->d_seg[0].ensure(2));
    HIP_TRY(h, h->d_seg[1].ensure(257));
    HIP_TRY(h, h->d_seg[2].ensure(65537));
    HIP_TRY(h, h->d_hist.ensure(65536));
    HIP_TRY(h, h->d_cursor.ensure(65536));
    HIP_TRY(h, h->d_dcounts.ensure(h->table.capacity + 2));
    const uint64_t seg0[2] = {0, nwords};
    HIP_TRY(h, hipMemcpyAsync(h->d_seg[0].p, seg0, sizeof seg0, hipMemcpyHostToDevice, h->stream));
    const uint64_t* bin_start = h->d_seg[0].p;
    uint32_t nbins = 1;
    for (uint32_t l = 0; l < levels; l++) {
        // (the number of logged words varies a little from run to run — racing duplicates in the scan's cache — so the
        //  buffers get slack: growing them by a few words would mean a fresh 800 MB allocation each time)
        HIP_TRY(h, h->d_part[l].ensure(nwords + nwords / 8 + (1u << 20)));
        RadixArgs R{};
        R.src = words;
        R.dst = h->d_part[l].p;
        R.seg_start = h->d_seg[l].p;
        R.nseg = nbins;
        R.shift = 56 - 8 * l;
        // the scan kernels counted the first digit of every word they logged (ScanArgs::word_hist)
        const bool counted = l == 0 && hist_counted && h->d_word_hist.p != nullptr;
        R.hist = counted ? h->d_word_hist.p + (size_t)ag.log_index * 256 : h->d_hist.p;
        R.cursor = h->d_cursor.p;
        R.cursor_stride = nbins == 1 ? kCursorStride : 1u;  // one segment: its 256 cursors would share 16 lines
        R.out_start = h->d_seg[l + 1].p;
        // slices per segment: enough workgroups to fill the GPU, never less than one tile each on average
        uint64_t tiles = (nwords + 8191) / 8192;
        uint32_t slices = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)h->num_cus * 8 / nbins + (nbins > 1 ? 8 : 0), tiles));
        HIP_TRY(h, launch_radix_pass(R, slices, h->stream, counted));
        words = R.dst;
        bin_start = R.out_start;
        nbins *= 256;
    }
    uint32_t* d_overflow = (uint32_t*)(h->d_counters.p + 24);
    HIP_TRY(h, hipMemsetAsync(h->d_dcounts.p, 0, (h->table.capacity + 2) * sizeof(unsigned long long), h->stream));
    HIP_TRY(h, hipMemsetAsync(d_overflow, 0, 8, h->stream));
    DedupeArgs D{};
    D.words = words;
    D.bin_start = bin_start;
    D.nbins = nbins;
    D.set_slots = set_slots;
    D.key_shift = h->nw_val_bits + 3;
    D.glob_off = ag.glob_off;
    D.counts = h->d_dcounts.p;
    D.overflow = d_overflow;
    dedupe_counters(h, D);
    HIP_TRY(h, launch_distinct_dedupe(h->prog, h->table, D, dedupe_grid(h, D, nbins), h->opt_dedupe_block | (h->opt_dedupe_unroll == 4 ? 2u : 0u), h->stream));
    uint32_t overflow = 0;
    HIP_TRY(h, hipMemcpyAsync(&overflow, d_overflow, 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->distinct_path |= 2u;
    if (overflow) {
        // some bin holds more distinct words than an LDS set takes (more than ~65536 * set_slots / 2 distinct members
        // in all): one open-addressed set in global memory over the whole word log instead
        uint64_t cap = next_pow2(std::max<uint64_t>(nwords * 2, 1024));
        HIP_TRY(h, h->d_wtable.ensure(cap));
        HIP_TRY(h, hipMemsetAsync(h->d_wtable.p, 0xFF, cap * 8, h->stream));
        HIP_TRY(h, hipMemsetAsync(h->d_dcounts.p, 0, (h->table.capacity + 2) * sizeof(unsigned long long), h->stream));
        HIP_TRY(h, launch_distinct_words_global(h->table, log, nwords, h->d_wtable.p, cap - 1,
                                                h->nw_val_bits + 3, h->d_dcounts.p, h->d_errp, h->num_cus * 8, h->stream));
        h->distinct_path |= 4u;
    }
    HIP_TRY(h, launch_distinct_add_counts(h->prog, h->table, h->d_dcounts.p, ag.glob_off, h->stream));
    return N1K_OK;
}

// The same when the specialised scan scattered the words into its hash regions already (the first partition pass is
// done): one more pass into bins of fixed capacity — no histogram, mix64 spreads distinct words evenly — and the LDS
// sets, without a host synchronisation (nothing here depends on a count the host would have to read).  Whenever that
// optimism fails (a sub-region or a bin overflowed: many copies of few words; an LDS set too small; words of the
// interpreter kernel in the plain log as well) everything is gathered into one log and the exact path above runs
// instead.  `nover` = words in the plain log.
n1k_status distinct_regions_finish(n1k_handle* h, const AggSpec& ag, uint64_t nover, bool force_exact, bool* deferred) {
    const uint32_t li = ag.log_index;
    const uint64_t cap = h->wregion_cap;
    unsigned long long* const cursors = h->d_wcursor.p + (size_t)li * kWordSubs * kCursorStride;
    const uint32_t set_slots = h->opt_distinct_set_slots;
    const uint64_t per_bin = std::max<uint64_t>((uint64_t)set_slots * h->opt_distinct_fill_pct / 100, 16);
    const bool exact = force_exact || nover > 0 || h->opt_distinct_levels == 0;
    uint32_t* d_overflow = (uint32_t*)(h->d_counters.p + 20);  // [0] an LDS set overflowed, [1] a bin of the second pass
    if (!exact) {
        // Optimistic: the member counts are only added to the groups when neither flag came up (the kernel checks), and
        // n1k_finish reads the flags together with the results (*deferred).  The rows pushed bound the words.
        const uint64_t bound = std::max<uint64_t>(h->row_base, 1);
        uint32_t bps = 1;  // bins per region: a bin's words should fit an LDS set at a quarter of its slots
        while (bps < 256 && 256ull * bps * per_bin < bound) bps *= 2;
        if (h->opt_distinct_levels == 2) bps = 256;
        else if (h->opt_distinct_levels == 1) bps = 1;
        const uint64_t nbins = 256ull * bps, mean = bound / nbins + 1, bin_cap = mean + mean / 2 + 256;
        HIP_TRY(h, h->d_dcounts.ensure(h->table.capacity + 2));
        HIP_TRY(h, hipMemsetAsync(h->d_dcounts.p, 0, (h->table.capacity + 2) * sizeof(unsigned long long), h->stream));
        HIP_TRY(h, h->d_part[0].ensure(nbins * bin_cap));
        HIP_TRY(h, h->d_cursor.ensure(65536));
        RadixArgs R{};
        R.src = h->d_wregion[li].p;
        R.dst = h->d_part[0].p;
        R.seg_count = cursors;
        R.seg_stride = cap;
        R.nseg = (uint32_t)kWordSubs;
        R.shift = 48;
        R.cursor = h->d_cursor.p;
        R.bin_cap = bin_cap;
        R.overflow = d_overflow + 1;
        const uint64_t region_tiles = (bound / 256 + 8191) / 8192 + kRecSubs;
        // workgroups per region: about three tiles each, 8 to 32 (measured at 100 M rows: 8 / 16 / 32 / 64 workgroups per region take
        // 714 / 670 / 660 / 750 us over 16-byte records — 95 tiles a region — and 382 / 375 / 393 / 452 us over 8-byte words — 48)
        const uint64_t wpr_auto = std::min<uint64_t>(32, std::max<uint64_t>(8, region_tiles / 3));
        const uint32_t wpr = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(h->opt_rec_slices ? h->opt_rec_slices : wpr_auto, region_tiles));
        HIP_TRY(h, launch_radix_scatter_words(R, wpr, bps, h->stream));
        DedupeArgs D{};
        D.words = R.dst;
        D.bin_count = h->d_cursor.p;
        D.count_stride = 1;
        D.pad1 = h->opt_spec_debug >> 8;  // (timing experiments only)
        D.bin_stride = bin_cap;
        D.nbins = (uint32_t)nbins;
        D.set_slots = set_slots;
        D.key_shift = h->nw_val_bits + 3;
        D.glob_off = ag.glob_off;
        D.counts = h->d_dcounts.p;
        D.overflow = d_overflow;
        dedupe_counters(h, D);
        HIP_TRY(h, launch_distinct_dedupe(h->prog, h->table, D, dedupe_grid(h, D, D.nbins), h->opt_dedupe_block | (h->opt_dedupe_unroll == 4 ? 2u : 0u), h->stream));
        HIP_TRY(h, launch_distinct_add_counts(h->prog, h->table, h->d_dcounts.p, ag.glob_off, h->stream, d_overflow));
        h->distinct_path |= 2u;
        *deferred = true;
        return N1K_OK;
    }
    // exact path: the sub-regions' words join the plain log (behind its own words), then partition by histogram
    std::vector<unsigned long long> rc((size_t)kWordSubs * kCursorStride);
    HIP_TRY(h, hipMemcpyAsync(rc.data(), cursors, rc.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    std::vector<uint64_t> off(kWordSubs);
    uint64_t at = nover;
    for (uint32_t b = 0; b < kWordSubs; b++) {
        off[b] = at;
        at += std::min<uint64_t>(rc[(size_t)b * kCursorStride], cap);
    }
    if (at == 0) return N1K_OK;
    HIP_TRY(h, h->d_wgather.ensure(at));
    HIP_TRY(h, h->d_woff.ensure(kWordSubs));
    if (nover) HIP_TRY(h, hipMemcpyAsync(h->d_wgather.p, h->d_log_word[li].p, nover * 8, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->d_woff.p, off.data(), kWordSubs * 8, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, launch_compact_regions(h->d_wregion[li].p, (uint32_t)kWordSubs, cap, cursors, h->d_woff.p, h->d_wgather.p, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));  // (`off` lives on this stack frame)
    return distinct_words_finish(h, ag, at, false, h->d_wgather.p);
}

}  // namespace n1k_eng
