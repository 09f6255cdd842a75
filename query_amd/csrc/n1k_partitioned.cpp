// n1k_partitioned.cpp — GROUP BY with many groups: records -> partition passes -> per-bin LDS tables (n1k_bins.hip).
#include "n1k_engine.h"

using namespace n1k;
using namespace n1k_eng;

namespace n1k_eng {

// the partitioned path carries up to kRecOperands distinct aggregate operands per record; no DISTINCT sets, no
// representative rows
bool partition_eligible(n1k_handle* h, PartitionPlan& pp) {
    const Program& P = h->prog;
    if (!h->plan.has_group || P.nkeys == 0 || h->has_distinct || P.want_rep_row) return false;
    pp.nsrc = 0;
    for (uint32_t a = 0; a < P.naggs; a++) {
        pp.agg_src[a] = 0xFFFFFFFFu;
        if (!P.aggs[a].has_operand) continue;
        uint32_t e = 0;
        for (; e < pp.nsrc; e++)
            if (!memcmp(&pp.src[e], &P.aggs[a].src, sizeof(Operand))) break;
        if (e == pp.nsrc) {
            if (pp.nsrc == kRecOperands) return false;
            pp.src[pp.nsrc++] = P.aggs[a].src;
        }
        pp.agg_src[a] = e;
    }
    return true;
}

// all keys dictionary coded and the key domain within reach of the workgroup tables: the scan kernels are at home
bool small_key_domain(const n1k_handle* h) {
    long double dom = 1;
    for (uint32_t k = 0; k < h->prog.nkeys; k++) {
        if (h->prog.keys[k].mode != KEYM_DICT) return false;
        dom *= (long double)h->dict.size() + 2;
    }
    return dom <= 4096;
}

n1k_status flush_pending(n1k_handle* h) {
    if (!h->pending.count) return N1K_OK;
    unsigned long long have = 0;
    HIP_TRY(h, hipMemcpyAsync(&have, h->d_counters.p + 1, sizeof have, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    n1k_status st = ensure_table_groups(h, have + h->pending.count);
    if (st != N1K_OK) return st;
    const uint64_t region_words = 2 + h->pending.cap * (1 + (uint64_t)h->prog.glob_words);
    HIP_TRY(h, launch_merge_partials(h->prog, h->table, 1, h->pending.cap, h->d_emit.p, region_words, h->d_errp,
                                     h->d_counters.p + 1, h->stream, h->pending.count, true));
    h->pending.count = 0;
    return N1K_OK;
}

n1k_status run_group_partitioned(n1k_handle* h, const n1k_batch* b, const PartitionPlan& pp, uint64_t groups_est,
                                 bool may_keep_region) {
    Program& P = h->prog;
    const uint64_t n = b->nrows;
    // the table is NOT sized by this batch's rows: the bins' groups are counted first (below) and the table grows to
    // what they need — 2^24 slots instead of 2^28 for config 5, which reset and finalize then scan
    n1k_status st = ensure_table(h, 0);
    if (st != N1K_OK) return st;
    // LDS table of the per-bin aggregation, and from it the number of partition passes
    uint32_t slots = (uint32_t)std::min<uint64_t>((64u * 1024u) / (P.lds_words * 8), 1u << 13);
    if (slots < 64) return run_group_batch(h, b);
    const uint64_t per_bin = slots / 4;  // groups a bin should hold on average
    const uint32_t levels = h->opt_partition_levels >= 0 ? (uint32_t)h->opt_partition_levels
                                                         : (groups_est <= per_bin ? 0u : (groups_est <= 256 * per_bin ? 1u : 2u));
    for (uint32_t i = 0; i <= std::min(levels, 2u); i++) {  // one set of record arrays per pass, plus the projection's
        HIP_TRY(h, h->d_rec_key[i].ensure(n));
        for (uint32_t e = 0; e < pp.nsrc; e++) {
            HIP_TRY(h, h->d_rec_pay[i][e].ensure(n));
            HIP_TRY(h, h->d_rec_tag[i][e].ensure(n));
        }
    }
    HIP_TRY(h, h->d_seg[0].ensure(2));
    HIP_TRY(h, h->d_seg[1].ensure(257));
    HIP_TRY(h, h->d_seg[2].ensure(65537));
    HIP_TRY(h, h->d_hist.ensure(65536));
    HIP_TRY(h, h->d_cursor.ensure(65536));
    auto rec = [&](int i) {
        RecArrays r{};
        r.key = h->d_rec_key[i].p;
        for (uint32_t e = 0; e < pp.nsrc; e++) {
            r.pay[e] = h->d_rec_pay[i][e].p;
            r.tag[e] = h->d_rec_tag[i][e].p;
        }
        return r;
    };
    hipEvent_t e0 = get_event(h), e1 = get_event(h);
    if (e0) (void)hipEventRecord(e0, h->stream);
    // (1) Filter + key + operands -> records
    unsigned long long* d_nrec = h->d_counters.p + 21;
    HIP_TRY(h, hipMemsetAsync(d_nrec, 0, sizeof(unsigned long long), h->stream));
    ProjectArgs A{};
    A.nrows = n;
    A.capacity = n;
    A.out = rec(0);
    A.cursor = d_nrec;
    for (uint32_t e = 0; e < pp.nsrc; e++) A.src[e] = pp.src[e];
    A.nsrc = pp.nsrc;
    A.err_flags = h->d_errp;
    A.hist = levels ? h->d_hist.p : nullptr;
    if (levels) HIP_TRY(h, hipMemsetAsync(h->d_hist.p, 0, 256 * sizeof(unsigned long long), h->stream));
    {
        uint64_t tiles = (n + 2047) / 2048;
        uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)h->num_cus * 4, tiles));
        HIP_TRY(h, launch_project_records(P, A, grid, h->stream));
    }
    // the number of records (rows that passed the Filter) sizes the passes: one small read-back
    unsigned long long nrec = 0;
    HIP_TRY(h, hipMemcpyAsync(&nrec, d_nrec, sizeof nrec, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    nrec = std::min<unsigned long long>(nrec, n);
    HIP_TRY(h, launch_add_counter(h->d_counters.p + 0, nrec, h->stream));  // rows_selected
    if (nrec) {
        const uint64_t seg0[2] = {0, nrec};
        HIP_TRY(h, hipMemcpyAsync(h->d_seg[0].p, seg0, sizeof seg0, hipMemcpyHostToDevice, h->stream));
        const uint64_t* bin_start = h->d_seg[0].p;
        uint32_t nbins = 1;
        int cur = 0;
        for (uint32_t l = 0; l < levels; l++) {
            RadixArgs R{};
            RecArrays src = rec(cur), dst = rec(cur + 1);
            R.src = src.key;
            R.dst = dst.key;
            R.nextra = pp.nsrc;
            for (uint32_t e = 0; e < pp.nsrc; e++) {
                R.src_pay[e] = src.pay[e];
                R.dst_pay[e] = dst.pay[e];
                R.src_tag[e] = src.tag[e];
                R.dst_tag[e] = dst.tag[e];
            }
            R.seg_start = h->d_seg[l].p;
            R.nseg = nbins;
            R.shift = 56 - 8 * l;
            R.hist = h->d_hist.p;
            R.cursor = h->d_cursor.p;
            R.cursor_stride = nbins == 1 ? kCursorStride : 1u;  // one segment: its 256 cursors would share 16 lines
            R.out_start = h->d_seg[l + 1].p;
            uint64_t tiles = (nrec + 8191) / 8192;
            uint32_t slices = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)h->num_cus * 8 / nbins + (nbins > 1 ? 8 : 0), tiles));
            HIP_TRY(h, launch_radix_pass(R, slices, h->stream, l == 0));  // the projection counted the first digit
            bin_start = R.out_start;
            nbins *= 256;
            cur++;
        }
        BinAggArgs B{};
        B.in = rec(cur);
        B.bin_start = bin_start;
        B.nbins = nbins;
        B.nsrc = pp.nsrc;
        // the bins' tables are cleared and scanned once per bin: no larger than the groups expected there need
        uint32_t bslots = slots;
        const uint64_t per = groups_est / nbins + 1;
        while (bslots > 256 && (uint64_t)bslots / 8 >= per) bslots /= 2;
        B.lds_slots = bslots;
        B.lds_max_fill = std::max(1u, bslots * 5 / 8);
        for (uint32_t a = 0; a < kMaxAggs; a++) B.agg_src[a] = a < P.naggs ? pp.agg_src[a] : 0xFFFFFFFFu;
        B.err_flags = h->d_errp;
        // partial groups of the bins: at most one per record, plus one per record and SUM/AVG for the values the
        // narrow LDS sums do not take
        uint32_t nsum = 0;
        for (uint32_t a = 0; a < P.naggs; a++) nsum += (P.aggs[a].kind == AGG_SUM || P.aggs[a].kind == AGG_AVG) ? 1u : 0u;
        const uint64_t ecap = nrec * (1 + nsum) + 1024;
        const uint64_t region_words = 2 + ecap * (1 + (uint64_t)P.glob_words);
        HIP_TRY(h, h->d_emit.ensure(region_words));
        HIP_TRY(h, hipMemsetAsync(h->d_emit.p, 0, 16, h->stream));
        B.emit = h->d_emit.p;
        B.emit_cap = ecap;
        B.emit_singletons = h->d_counters.p + 22;
        HIP_TRY(h, hipMemsetAsync(h->d_counters.p + 22, 0, sizeof(unsigned long long), h->stream));
        const size_t shmem = (size_t)bslots * P.lds_words * 8 + 1024;
        const uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(4, (160 * 1024) / shmem));
        uint32_t grid = (uint32_t)std::min<uint64_t>(nbins, (uint64_t)h->num_cus * per_cu);
        HIP_TRY(h, launch_agg_bins(P, B, h->table, h->d_counters.p + 1, grid, h->stream));
        // how many partial groups, how many groups already: the table grows to hold both, then the merge
        unsigned long long emitted = 0, have = 0, singletons = 0;
        HIP_TRY(h, hipMemcpyAsync(&singletons, h->d_counters.p + 22, sizeof singletons, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipMemcpyAsync(&emitted, h->d_emit.p, sizeof emitted, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipMemcpyAsync(&have, h->d_counters.p + 1, sizeof have, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        emitted = std::min<unsigned long long>(emitted, ecap);
        if (may_keep_region && have == 0 && singletons == 0) {
            // nothing else in the handle and every key of the region is unique: the region is the set of groups
            h->pending.count = emitted;
            h->pending.cap = ecap;
            if (e1) (void)hipEventRecord(e1, h->stream);
            h->events.emplace_back(e0, e1);
            h->stats.agg_mode = N1K_MODE_PARTITIONED;
            h->stats.spec_kernel = 0;
            return N1K_OK;
        }
        st = ensure_table_groups(h, have + emitted);
        if (st != N1K_OK) return st;
        // every key of the region is unique unless rows left the bins on their own: new groups are then plain copies
        HIP_TRY(h, launch_merge_partials(P, h->table, 1, ecap, h->d_emit.p, region_words, h->d_errp, h->d_counters.p + 1,
                                         h->stream, emitted, singletons == 0));
    }
    if (e1) (void)hipEventRecord(e1, h->stream);
    h->events.emplace_back(e0, e1);
    h->stats.agg_mode = N1K_MODE_PARTITIONED;
    h->stats.spec_kernel = 0;
    return N1K_OK;
}

// The partitioned path with the plan-specialised front end (n1k_spec.h, records mode).  Optimistic: hash regions and
// bins have fixed capacities (mix64 spreads the keys evenly unless few keys own most rows); when either overflows —
// or the plan's shape has no specialised kernel — *done stays false, nothing of the batch has been kept, and the caller
// runs the exact path (run_group_partitioned: histogram-driven passes over three-array records).
n1k_status run_group_records(n1k_handle* h, const n1k_batch* b, const PartitionPlan& pp, uint64_t groups_est, bool may_keep_region,
                             bool* done) {
    *done = false;
    Program& P = h->prog;
    const uint64_t n = b->nrows;
    if (!h->opt_records || !h->opt_spec || pp.nsrc > 1 || n == 0 || n >= (1ull << 31)) return N1K_OK;
    FastArgs F;
    const uint32_t direct_max_slots = (uint32_t)std::min<uint64_t>((156u * 1024u) / (P.lds_words * 8), 1u << 15);
    if (!build_fast_args(h, direct_max_slots, F)) return N1K_OK;
    const SpecSig sig = make_plan_sig(h, F);
    const SpecEntry* spec = find_spec(sig);
    const JitKernel* jit = nullptr;
    if (!spec && h->opt_jit) {
        jit = jit_get(sig);
        if (jit->failed || !jit->rec_wide) {
            h->jit_log = jit->log;
            jit = nullptr;
        }
    }
    if (!spec && !jit) return N1K_OK;
    n1k_status st = ensure_table(h, 0);
    if (st != N1K_OK) return st;
    // the scan: tiles of 2048 rows, a grid that is a multiple of 8 (sub-region = workgroup label, n1k_spec.h)
    const uint64_t tiles = (n + 2047) / 2048;
    const uint32_t per_cu = h->opt_rec_scan_per_cu ? h->opt_rec_scan_per_cu : 2u;
    const uint32_t grid = (uint32_t)((std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)h->num_cus * per_cu, tiles)) + 7) / 8 * 8);
    // The per-bin LDS tables: `slots` slots each (option rec_slots), filled to 5/8 at most; the second pass splits every
    // hash region into as many bins (a power of two <= 256, option rec_bins) as it takes for a bin's expected groups —
    // the probe's estimate carries a factor of two already — to stay under half of that.
    uint32_t slots = h->opt_rec_slots ? h->opt_rec_slots : 1024u;
    while (slots > 64 && (size_t)slots * P.lds_words * 8 > 64u * 1024u) slots /= 2;
    if ((size_t)slots * P.lds_words * 8 > 64u * 1024u) return N1K_OK;
    const uint64_t groups = std::min<uint64_t>(groups_est, n);
    const uint64_t want_bins = groups / (slots / 2) + 1;
    uint32_t bps = 1;  // bins per region
    while (bps < 256 && 256ull * bps < want_bins) bps *= 2;
    if (h->opt_rec_bins) bps = h->opt_rec_bins;
    const uint64_t nbins = 256ull * bps;
    // Capacities.  A bin's (a sub-region's) record count is a sum over its groups: variance = mean x (rows per group + 1);
    // six deviations and an eighth (a quarter) on top.  Whatever overflows raises a flag and the batch takes the exact path.
    const double rows_per_group = std::min<double>((double)n, 2.0 * (double)n / (double)std::max<uint64_t>(groups, 1) + 1.0);
    const uint64_t mean = n / nbins + 1;
    const uint64_t bin_cap = mean + mean / 8 + (uint64_t)(6.0 * std::sqrt((double)mean * (rows_per_group + 1.0))) + 64;
    const uint64_t nsub = 256ull * kRecSubs, sub_mean = n / nsub + 1;
    const uint64_t cap = sub_mean + sub_mean / 4 + (uint64_t)(6.0 * std::sqrt((double)sub_mean * (rows_per_group + 1.0))) + 256;
    HIP_TRY(h, h->d_rregion.ensure(2 * nsub * cap));
    HIP_TRY(h, h->d_rcursor.ensure(nsub * kCursorStride));
    HIP_TRY(h, h->d_rbins.ensure(2 * nbins * bin_cap));
    HIP_TRY(h, h->d_cursor.ensure(65536));
    uint32_t* d_flags = (uint32_t*)(h->d_counters.p + 20);  // [0] a hash region overflowed, [1] a bin
    hipEvent_t e0 = get_event(h), e1 = get_event(h);
    if (e0) (void)hipEventRecord(e0, h->stream);
    HIP_TRY(h, hipMemsetAsync(h->d_rcursor.p, 0, nsub * kCursorStride * sizeof(unsigned long long), h->stream));
    HIP_TRY(h, hipMemsetAsync(d_flags, 0, 8, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->d_counters.p + 25, h->d_counters.p + 0, 8, hipMemcpyDeviceToDevice, h->stream));  // rows_selected, to undo
    // (1) Filter + packed key + operand -> records in the hash regions
    bool aligned = true;
    for (uint32_t c = 0; c < P.ncols; c++) {
        F.cols[c] = P.cols[c];
        aligned &= ((uintptr_t)F.cols[c].tags % 2 == 0) && ((uintptr_t)F.cols[c].payload % 16 == 0) && ((uintptr_t)F.cols[c].codes % 8 == 0);
    }
    const bool wide = aligned && h->opt_wide && n >= 2;
    F.nrows = (uint32_t)n;
    F.row_base = h->row_base;
    F.err_flags = h->d_errp;
    F.rows_selected = h->d_counters.p + 0;
    WordLogArgs L;
    memset(&L, 0, sizeof L);
    L.region[0] = h->d_rregion.p;
    L.region_cursor[0] = h->d_rcursor.p;
    L.region_cap = cap;
    L.rec_overflow = d_flags;
    L.pad = h->opt_spec_debug & 3u;  // (timing experiments only)
    if (spec) HIP_TRY(h, spec->launch_records(P, F, grid, wide, L, h->stream));
    else HIP_TRY(h, jit_launch_records(jit, P, F, grid, wide, L, h->stream));
    // (2) the second partition pass: the regions' 8 sub-regions into `bps` bins of fixed capacity per region
    BinAggArgs B{};
    B.nsrc = pp.nsrc;
    {
        RadixArgs R{};
        R.src = h->d_rregion.p;
        R.dst = h->d_rbins.p;
        R.seg_count = h->d_rcursor.p;
        R.seg_stride = cap;
        R.nseg = (uint32_t)nsub;
        R.shift = 48;
        R.cursor = h->d_cursor.p;
        R.bin_cap = bin_cap;
        R.overflow = d_flags + 1;
        const uint64_t region_tiles = (n / 256 + 4095) / 4096 + kRecSubs;
        // workgroups per region: about three tiles each, 8 to 32 (measured at 100 M rows: 8 / 16 / 32 / 64 workgroups per region take
        // 714 / 670 / 660 / 750 us over 16-byte records — 95 tiles a region — and 382 / 375 / 393 / 452 us over 8-byte words — 48)
        const uint64_t wpr_auto = std::min<uint64_t>(32, std::max<uint64_t>(8, region_tiles / 3));
        const uint32_t wpr = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(h->opt_rec_slices ? h->opt_rec_slices : wpr_auto, region_tiles));
        HIP_TRY(h, launch_radix_scatter16(R, wpr, bps, h->stream));
        B.rec = h->d_rbins.p;
        B.bin_count = h->d_cursor.p;
        B.bin_count_stride = 1;
        B.bin_stride = bin_cap;
        B.nbins = (uint32_t)nbins;
    }
    // (3) one workgroup per bin: InitialGroup in an LDS table, the bin's groups into the compact region
    B.lds_slots = slots;
    B.pad1 = h->opt_spec_debug >> 8;  // (timing experiments only)
    B.lds_max_fill = std::max(1u, slots * 5 / 8);
    for (uint32_t a = 0; a < kMaxAggs; a++) B.agg_src[a] = a < P.naggs ? pp.agg_src[a] : 0xFFFFFFFFu;
    B.err_flags = h->d_errp;
    uint32_t nsum = 0;
    for (uint32_t a = 0; a < P.naggs; a++) nsum += (P.aggs[a].kind == AGG_SUM || P.aggs[a].kind == AGG_AVG) ? 1u : 0u;
    const uint64_t ecap = n * (1 + nsum) + 1024;
    const uint64_t region_words = 2 + ecap * (1 + (uint64_t)P.glob_words);
    HIP_TRY(h, h->d_emit.ensure(region_words));
    HIP_TRY(h, hipMemsetAsync(h->d_emit.p, 0, 16, h->stream));
    B.emit = h->d_emit.p;
    B.emit_cap = ecap;
    B.emit_singletons = h->d_counters.p + 22;
    HIP_TRY(h, hipMemsetAsync(h->d_counters.p + 22, 0, sizeof(unsigned long long), h->stream));
    {
        const uint32_t block = h->opt_rec_block ? h->opt_rec_block : 256u;
        const size_t shmem = agg_bins16_lds_bytes(P, slots) + 1024;
        const uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(2048 / block / 2, (160 * 1024) / shmem));
        const uint32_t bgrid = (uint32_t)std::min<uint64_t>(B.nbins, (uint64_t)h->num_cus * per_cu);
        HIP_TRY(h, launch_agg_bins16(P, B, bgrid, block, h->opt_rec_unroll ? h->opt_rec_unroll : 2u /* (2 records per thread and chunk: 554-586 us against 566-589 with 4, three boxes) */, h->stream, h->opt_agg_spec != 0));
    }
    // one copy of the counters into pinned memory (the region's group count joins them first): one host round trip
    {
        n1k_status pst = ensure_pinned_counters(h);
        if (pst != N1K_OK) return pst;
    }
    HIP_TRY(h, hipMemcpyAsync(h->d_counters.p + 26, h->d_emit.p, 8, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->pin_counters, h->d_counters.p, kCounters * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    unsigned long long emitted = h->pin_counters[26];
    const unsigned long long have = h->pin_counters[1], singletons = h->pin_counters[22];
    uint32_t flags[2];
    memcpy(flags, &h->pin_counters[20], 8);
    if (flags[0] | flags[1]) {
        // a region or a bin overflowed: nothing was merged anywhere yet — forget the records and the survivor count
        HIP_TRY(h, hipMemcpyAsync(h->d_counters.p + 0, h->d_counters.p + 25, 8, hipMemcpyDeviceToDevice, h->stream));
        if (e0) h->event_pool.push_back(e0);
        if (e1) h->event_pool.push_back(e1);
        return N1K_OK;
    }
    *done = true;
    emitted = std::min<unsigned long long>(emitted, ecap);
    h->stats.agg_mode = N1K_MODE_PARTITIONED;
    h->stats.spec_kernel = spec ? 1u : 2u;
    if (may_keep_region && have == 0 && singletons == 0) {
        h->pending.count = emitted;
        h->pending.cap = ecap;
        if (e1) (void)hipEventRecord(e1, h->stream);
        h->events.emplace_back(e0, e1);
        return N1K_OK;
    }
    st = ensure_table_groups(h, have + emitted);
    if (st != N1K_OK) return st;
    HIP_TRY(h, launch_merge_partials(P, h->table, 1, ecap, h->d_emit.p, region_words, h->d_errp, h->d_counters.p + 1, h->stream, emitted,
                                     singletons == 0));
    if (e1) (void)hipEventRecord(e1, h->stream);
    h->events.emplace_back(e0, e1);
    return N1K_OK;
}

}  // namespace n1k_eng
