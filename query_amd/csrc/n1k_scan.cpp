// n1k_scan.cpp — one column batch through Filter + InitialGroup: which kernel family runs it (plan-specialised ahead of
// time or at run time, bounded-shape, interpreter), Filter-only batches, derived columns, staging of host batches.
#include "n1k_engine.h"

using namespace n1k;
using namespace n1k_eng;

namespace n1k_eng {

// Can this plan run on the fast kernel (bounded shape, every descriptor static)?  Fills F when it can.
// fuse: the plan's arithmetic nodes stay out of HBM — the kernel (a run-time-built plan-specialised one) evaluates them in
// registers from the input columns; otherwise they are materialised derived columns and count as inputs.
// partition_only: only columns, terms and keys matter (the row exchange's partition kernels: no aggregate runs there).
bool build_fast_args(n1k_handle* h, uint32_t max_slots, FastArgs& F, bool fuse, bool partition_only) {
    const Program& P = h->prog;
    memset(&F, 0, sizeof F);
    const uint32_t ni = (uint32_t)h->plan.paths.size(), nd = (uint32_t)h->derived.size();
    if (fuse) {
        if (nd == 0 || nd > (uint32_t)kFastDerived || ni == 0 || ni > (uint32_t)kFastCols) return false;
        for (uint32_t d = 0; d < nd; d++)
            for (uint32_t k = 0; k < h->derived[d].nops; k++) {
                const Operand& o = h->derived[d].ops[k];
                if (h->derived[d].op >= AR_GREATEST) return false;  // (collation needs the string ranks: derived columns)
                if (!o.is_const && o.col >= ni + d) return false;
                if (o.is_const) F.dconst[d][k] = o.cpayload;
            }
        F.nderived = nd;
    }
    if (h->opt_fast == 0 || P.want_rep_row || P.ncols == 0 || (!fuse && P.ncols > (uint32_t)kFastCols)) return false;
    if (P.nkeys > (uint32_t)kFastKeys || (!partition_only && (P.naggs > (uint32_t)kFastAggs || P.naggs == 0))) return false;
    // predicate: none, one term, or AND of two terms
    uint32_t term_ix[2] = {0, 0};
    if (P.nlogic == 0) F.nterms = 0;
    else if (P.nlogic == 1 && P.logic[0].op == LOGIC_PUSH) { F.nterms = 1; term_ix[0] = P.logic[0].arg; }
    else if (P.nlogic == 3 && P.logic[0].op == LOGIC_PUSH && P.logic[1].op == LOGIC_PUSH && P.logic[2].op == LOGIC_AND &&
             P.logic[2].arg == 2) { F.nterms = 2; term_ix[0] = P.logic[0].arg; term_ix[1] = P.logic[1].arg; }
    else return false;
    for (uint32_t i = 0; i < F.nterms; i++) {
        const Term& t = P.terms[term_ix[i]];
        FastTerm& ft = F.terms[i];
        if (t.op >= TERM_NUM_LT && t.op <= TERM_NUM_EQ) {
            if (t.a.is_const) return false;
            ft.op = t.op; ft.col = t.a.col; ft.ctag = t.b.ctag; ft.cpayload = t.b.cpayload;
        } else if (t.op >= TERM_IS_NULL && t.op <= TERM_IS_NOT_VALUED) {
            if (t.a.is_const) return false;
            ft.op = t.op; ft.col = t.a.col;
        } else if (t.op == TERM_EQ) {  // column = "string constant" (either side)
            const Operand *c = nullptr, *k = nullptr;
            if (!t.a.is_const && t.b.is_const && t.b.ctag == T_STRING) { c = &t.a; k = &t.b; }
            else if (!t.b.is_const && t.a.is_const && t.a.ctag == T_STRING) { c = &t.b; k = &t.a; }
            else return false;
            ft.op = TERM_STR_EQ; ft.col = c->col; ft.ctag = T_STRING; ft.cpayload = k->cpayload;
        } else return false;
    }
    // keys: dictionary columns whose domain fits the LDS table are addressed by perfect hash (DIRECT); anything else
    // (integer keys, big dictionaries) goes through an open-addressed LDS table on the packed key (hashed)
    uint64_t domain = 1;
    bool direct = true;
    F.nkeys = P.nkeys;
    for (uint32_t k = 0; k < P.nkeys; k++) {
        const KeySpec& ks = P.keys[k];
        if (ks.src.is_const) return false;
        F.keys[k].col = ks.src.col;
        F.keys[k].shift = ks.shift;
        if (ks.mode != KEYM_DICT) { direct = false; continue; }
        uint64_t radix = (uint64_t)h->dict.size() + 2;
        if (ks.bits < 64 && radix > (1ull << ks.bits)) return false;
        F.keys[k].stride = (uint32_t)std::min<uint64_t>(domain, 0xFFFFFFFFull);
        F.keys[k].radix = (uint32_t)radix;
        domain *= radix;
        if (domain > max_slots) direct = false;
    }
    if (direct) {
        F.hashed = 0;
        F.lds_slots = (uint32_t)std::max<uint64_t>(domain, 2);
    } else {
        F.hashed = 1;
        uint32_t slots = (uint32_t)std::min<uint64_t>(h->opt_lds_bytes / (P.lds_words * 8), 1u << 15);
        if (slots < 16) return false;
        F.lds_slots = slots;
        F.lds_max_fill = std::max(1u, (uint32_t)((uint64_t)slots * 5 / 8));
    }
    F.naggs = partition_only ? 0u : P.naggs;
    uint32_t ndist = 0;
    for (uint32_t a = 0; a < F.naggs; a++) {
        const AggSpec& ag = P.aggs[a];
        if (ag.distinct) {
            // COUNT(DISTINCT column) whose members leave as one word: the specialised kernels scatter them into hash
            // regions; anything else DISTINCT stays with the interpreter kernel
            if (ag.kind != AGG_COUNT || !ag.has_operand || ag.src.is_const || !h->layout_fixed || !h->distinct_words[ag.log_index] ||
                ++ndist > kSpecDistinct)
                return false;
        }
        if (ag.has_operand) {
            if (ag.src.is_const) return false;
            F.agg_col[a] = ag.src.col;
        }
    }
    F.ncols = fuse ? ni : P.ncols;
    for (uint32_t c = 0; c < F.ncols; c++) F.cols[c] = P.cols[c];
    return true;
}

// shape of the compiled plan (what a plan-specialised kernel is instantiated for)
SpecSig make_plan_sig(const n1k_handle* h, const FastArgs& F) {
    const Program& P = h->prog;
    SpecSig g{};
    g.ncols = (int)F.ncols; g.nterms = (int)F.nterms; g.nkeys = (int)F.nkeys; g.naggs = (int)F.naggs;
    g.hashed = (int)F.hashed;  // a dictionary domain beyond the LDS has no prebuilt kernel: interpreter
    for (uint32_t c = 0; c < F.ncols; c++) g.col_kind[c] = F.cols[c].kind;
    for (uint32_t t = 0; t < F.nterms; t++) {
        g.terms[t].op = F.terms[t].op;
        g.terms[t].col = F.terms[t].col;
        bool num = F.terms[t].op >= TERM_NUM_LT && F.terms[t].op <= TERM_NUM_EQ;
        g.terms[t].const_int = num && F.terms[t].ctag == T_INT ? 1u : 0u;
    }
    for (uint32_t k = 0; k < F.nkeys; k++) g.key_col[k] = F.keys[k].col;
    for (uint32_t a = 0; a < F.naggs; a++) {
        g.aggs[a].kind = P.aggs[a].kind;
        g.aggs[a].has_operand = P.aggs[a].has_operand;
        g.aggs[a].col = P.aggs[a].has_operand ? F.agg_col[a] : 0u;
        g.aggs[a].distinct = P.aggs[a].distinct ? 1u : 0u;
    }
    g.nderived = (int)F.nderived;
    for (uint32_t d = 0; d < F.nderived; d++) {
        g.derived[d].op = h->derived[d].op;
        g.derived[d].nops = h->derived[d].nops;
        for (uint32_t k = 0; k < h->derived[d].nops; k++) {
            const Operand& o = h->derived[d].ops[k];
            g.derived[d].ops[k].is_const = o.is_const ? 1u : 0u;
            g.derived[d].ops[k].v = o.is_const ? o.ctag : o.col;
        }
    }
    return g;
}

// exact-shape lookup among the ahead-of-time instantiated plan shapes (n1k_spec.h)
const SpecEntry* find_spec(const SpecSig& g) {
    for (const SpecEntry& e : spec_registry())
        if (memcmp(&e.sig, &g, sizeof g) == 0) return &e;
    return nullptr;
}

n1k_status run_group_batch(n1k_handle* h, const n1k_batch* b) {
    Program& P = h->prog;
    n1k_status st = ensure_table(h, b->nrows);
    if (st != N1K_OK) return st;
    ScanArgs A{};
    A.nrows = b->nrows;
    A.nrows_dev = h->push_nrows_dev;
    A.row_base = h->row_base;
    if (h->has_distinct) {
        // every qualifying operand appends one (group key, value, class) pair: at most one per row and aggregate
        uint64_t need = h->row_base + b->nrows;
        if (need > h->log_capacity) {
            uint64_t cap = std::max<uint64_t>(need, h->log_capacity * 2);
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            for (uint32_t d = 0; d < h->n_distinct; d++) {
                DevBuf<uint64_t> nk, nv, nw;
                DevBuf<uint8_t> nc;
                HIP_TRY(h, nk.ensure(cap));
                HIP_TRY(h, nv.ensure(cap));
                HIP_TRY(h, nc.ensure(cap));
                if (h->distinct_words[d]) HIP_TRY(h, nw.ensure(cap));
                if (h->log_capacity) {
                    HIP_TRY(h, hipMemcpy(nk.p, h->d_log_key[d].p, h->log_capacity * 8, hipMemcpyDeviceToDevice));
                    HIP_TRY(h, hipMemcpy(nv.p, h->d_log_val[d].p, h->log_capacity * 8, hipMemcpyDeviceToDevice));
                    HIP_TRY(h, hipMemcpy(nc.p, h->d_log_cls[d].p, h->log_capacity, hipMemcpyDeviceToDevice));
                    if (h->distinct_words[d])
                        HIP_TRY(h, hipMemcpy(nw.p, h->d_log_word[d].p, h->log_capacity * 8, hipMemcpyDeviceToDevice));
                }
                h->d_log_key[d].release();
                h->d_log_val[d].release();
                h->d_log_cls[d].release();
                h->d_log_word[d].release();
                h->d_log_key[d] = nk;
                h->d_log_val[d] = nv;
                h->d_log_cls[d] = nc;
                h->d_log_word[d] = nw;
            }
            h->log_capacity = cap;
        }
        for (uint32_t d = 0; d < h->n_distinct; d++) {
            A.log_key[d] = h->d_log_key[d].p;
            A.log_val[d] = h->d_log_val[d].p;
            A.log_cls[d] = h->d_log_cls[d].p;
            A.log_word[d] = h->distinct_words[d] ? h->d_log_word[d].p : nullptr;
        }
        A.log_cursor = h->d_counters.p + 8;
        A.word_cursor = h->d_counters.p + 16;
        A.log_capacity = h->log_capacity;
        A.nw_key_bits = h->nw_key_bits;
        A.nw_val_bits = h->nw_val_bits;
        if (!h->d_word_hist.p) {
            HIP_TRY(h, h->d_word_hist.ensure(kMaxDistinct * 256));
            HIP_TRY(h, hipMemsetAsync(h->d_word_hist.p, 0, kMaxDistinct * 256 * sizeof(unsigned long long), h->stream));
        }
        A.word_hist = h->d_word_hist.p;
        bool any_words = false;
        for (uint32_t d = 0; d < h->n_distinct; d++) any_words |= h->distinct_words[d];
        A.dcache_aggs = any_words ? h->n_distinct : 0;
        A.dcache_slots = any_words ? 4096u / (h->n_distinct > 2 ? 4u : h->n_distinct) : 0;  // 32 KB of LDS in all
    }
    uint32_t block = h->opt_block ? h->opt_block : 1024;
    uint32_t rpl = block == 1024 ? h->opt_rows_per_lane : 4;
    uint32_t max_slots = h->opt_lds_bytes / (P.lds_words * 8);
    max_slots = (uint32_t)std::min<uint64_t>(max_slots, 1u << 15);
    if (max_slots < 2) return fail(h, N1K_UNSUPPORTED, "accumulator row too wide for LDS");
    FastArgs F;
    // DIRECT tables may take (almost) the whole 160 KiB LDS of a CU: occupancy is chosen from the table size
    const uint32_t direct_max_slots = (uint32_t)std::min<uint64_t>((156u * 1024u) / (P.lds_words * 8), 1u << 15);
    // Arithmetic nodes not materialised yet: is there a run-time-built kernel of this shape that evaluates them in
    // registers (same conditions as the kernel choice below)?  If not they become derived columns now.
    bool fuse = false;
    if (!h->derived_ready) {
        if (h->opt_fuse_arith && h->opt_agg_mode != N1K_MODE_LDS_HASH && h->opt_spec && h->opt_jit &&
            (h->opt_jit == 2 || b->nrows >= h->opt_jit_min_rows) && build_fast_args(h, direct_max_slots, F, true)) {
            bool kh = false;
            for (uint32_t k = 0; k < F.nkeys; k++) kh |= F.keys[k].col >= F.ncols || F.cols[F.keys[k].col].kind != COLK_DICT32;
            if ((size_t)F.lds_slots * P.lds_words * 8 <= 64 * 1024 && kh == (F.hashed != 0)) {
                SpecSig fs = make_plan_sig(h, F);
                fs.seg = h->push_nseg > 1 ? 1 : 0;
                const JitKernel* k = jit_get(fs);
                if (k->failed) h->jit_log = k->log;
                else fuse = true;
            }
        }
        if (!fuse) {
            st = materialize_derived(h, b);
            if (st != N1K_OK) return st;
        }
    }
    if (h->opt_agg_mode != N1K_MODE_LDS_HASH && build_fast_args(h, direct_max_slots, F, fuse)) {
        // Shapes with COUNT(DISTINCT): the specialised kernels keep nothing of a DISTINCT aggregate in the workgroup
        // table (its member words go to the hash regions), so they run on a copy of the program with a compact LDS layout
        Program Pc;
        uint32_t ndist = 0;
        for (uint32_t a = 0; a < P.naggs; a++) ndist += P.aggs[a].distinct ? 1u : 0u;
        if (ndist) {
            Pc = P;
            uint32_t w = 1;
            for (uint32_t a = 0; a < Pc.naggs; a++) {
                AggSpec& ag = Pc.aggs[a];
                ag.lds_off = w;
                if (ag.distinct) ag.lds_n = 0;
                else w += (ag.kind == AGG_COUNT || ag.kind == AGG_COUNTN) ? 1u : (ag.kind == AGG_SUM ? kLdsWordsSum : (ag.kind == AGG_AVG ? kLdsWordsAvg : kWordsMinMax));
            }
            Pc.lds_words = w;
        }
        const Program& P = ndist ? Pc : h->prog;  // (shadows the handle's program for the launches below)
        const uint32_t table_bytes = F.lds_slots * P.lds_words * 8;
        // the word scatter's LDS (per DISTINCT aggregate one ScatterLds<uint64_t, 512, 4>, n1k_scatter.h: 2048 staged words,
        // counters, run starts) and, in what is left of the workgroup's share of the CU, its "already logged" caches
        const uint32_t scatter_bytes = ndist * (2048u * 8u + 2u * 256u * 4u + 256u * 4u + 256u * 8u + 2048u) + (ndist ? 64u : 0u);
        uint32_t dcache_slots = 0;
        if (ndist) {
            const uint32_t without = table_bytes + scatter_bytes;
            const uint32_t share = 160u * 1024u / std::max(1u, std::min(3u, 160u * 1024u / (without + 512u)));
            for (uint32_t sl = 4096; sl >= 64; sl >>= 1)
                if (without + ndist * sl * 8u + 512u <= share) { dcache_slots = sl; break; }
        }
        const uint32_t lds_total = table_bytes + scatter_bytes + ndist * dcache_slots * 8u;
        // workgroups per CU that fit: 512 threads x 3 (<= 48 KiB each), x 2 (<= 72 KiB), else 1024 threads x 1
        uint32_t fblock = h->opt_block == 1024 || h->opt_block == 512 ? h->opt_block : (table_bytes <= 72 * 1024 ? 512u : 1024u);
        if (ndist) fblock = 512;
        uint32_t per_cu = fblock == 512 ? (lds_total <= 48 * 1024 ? 3u : (lds_total <= 72 * 1024 ? 2u : 1u))
                                        : (lds_total <= 72 * 1024 ? 2u : 1u);
        if (ndist) per_cu = std::max(1u, std::min(3u, 160u * 1024u / (lds_total + 512u)));  // (two workgroups of 80 KiB fit a CU)
        uint32_t frpl = h->opt_rows_per_lane;
        uint32_t fgrid = h->opt_grid_blocks ? h->opt_grid_blocks : (uint32_t)(h->num_cus * per_cu);
        // slabs + merge kernel pay off once the table is more than a few KiB
        const bool use_slabs = !F.hashed && (h->opt_slabs == 1 ? table_bytes >= 4096 : h->opt_slabs == 2);
        F.err_flags = h->d_errp;
        F.rows_selected = h->d_counters.p + 0;
        // a prebuilt plan-specialised kernel of exactly this shape?
        // (segmented batches — a received row region — run on the run-time-built variant of the shape only: the prebuilt
        //  kernels carry none of the segment bookkeeping)
        SpecSig sig = make_plan_sig(h, F);
        sig.seg = h->push_nseg > 1 ? 1 : 0;
        const SpecEntry* spec = h->opt_spec && !sig.seg ? find_spec(sig) : nullptr;
        // no prebuilt kernel of this shape: instantiate the same template at run time (large batches, or forced)
        const JitKernel* jit = nullptr;
        bool key_kinds_hashed = false;
        for (uint32_t k = 0; k < F.nkeys; k++) key_kinds_hashed |= F.keys[k].col >= F.ncols || F.cols[F.keys[k].col].kind != COLK_DICT32;  // (a fused node is a TAGGED64 value)
        if (!spec && h->opt_spec && h->opt_jit && (h->opt_jit == 2 || b->nrows >= h->opt_jit_min_rows) &&
            table_bytes <= 64 * 1024 && key_kinds_hashed == (F.hashed != 0)) {
            jit = jit_get(sig);
            if (jit->failed) {
                h->jit_log = jit->log;
                jit = nullptr;
            }
        }
        if (jit && fblock != 512) {  // run-time instantiations are built for 512-thread workgroups
            fblock = 512;
            per_cu = table_bytes <= 48 * 1024 ? 3u : 2u;
            fgrid = h->opt_grid_blocks ? h->opt_grid_blocks : (uint32_t)(h->num_cus * per_cu);
        }
        h->stats.spec_kernel = spec ? 1u : (jit ? (F.nderived ? 3u : 2u) : 0u);
        if (F.nderived && !jit) return fail(h, N1K_DEVICE_ERROR, "fused arithmetic without its kernel");  // (decided above)
        if ((F.hashed || ndist || h->push_nrows_dev || h->push_nseg) && !spec && !jit) goto interpreter;  // the bounded-shape kernel is DIRECT only, no DISTINCT
        F.nrows_dev = h->push_nrows_dev;
        if (h->push_nseg > 1) {
            if (h->push_nseg > kMaxSegments || b->nrows >= (1ull << 31)) return fail(h, N1K_INVALID, "segmented batch too large");
            F.nseg = h->push_nseg;
            F.seg_rows = (uint32_t)h->push_seg_rows;
            F.seg_count_stride = kCursorStride;
            F.seg_counts = h->push_seg_counts;
        }
        WordLogArgs L;
        memset(&L, 0, sizeof L);
        if (ndist) {
            // hash regions: 256 x kRecSubs sub-regions per DISTINCT aggregate, each with room for its share of all rows
            // pushed so far plus a quarter (mix64 spreads distinct words evenly; many copies of few words overflow into the
            // plain word log)
            const uint64_t rows_total = h->row_base + b->nrows;
            uint64_t need = h->opt_region_cap ? h->opt_region_cap : (rows_total + rows_total / 4) / kWordSubs + 4096;
            need = (need + 15) / 16 * 16;  // whole 128-byte lines
            if (!h->d_wcursor.p) {
                HIP_TRY(h, h->d_wcursor.ensure(kMaxDistinct * kWordSubs * kCursorStride));
                HIP_TRY(h, hipMemsetAsync(h->d_wcursor.p, 0, kMaxDistinct * kWordSubs * kCursorStride * sizeof(unsigned long long), h->stream));
            }
            if (need > h->wregion_cap) {
                const uint64_t ncap = (std::max<uint64_t>(need, h->wregion_cap * 2) + 15) / 16 * 16;
                if (ncap >= 0xFFFFFF00ull) return fail(h, N1K_OOM, "COUNT(DISTINCT): more than 2^32 words per hash region");
                HIP_TRY(h, hipStreamSynchronize(h->stream));
                for (uint32_t a = 0; a < P.naggs; a++) {
                    if (!P.aggs[a].distinct) continue;
                    const uint32_t li = P.aggs[a].log_index;
                    DevBuf<uint64_t> nb;
                    HIP_TRY(h, nb.ensure(kWordSubs * ncap));
                    if (h->wregion_cap && h->wregion_used)
                        HIP_TRY(h, launch_regrow_regions(h->d_wregion[li].p, kWordSubs, h->wregion_cap, nb.p, ncap,
                                                         h->d_wcursor.p + (size_t)li * kWordSubs * kCursorStride, h->stream));  // (clamps the cursors of regions that had overflowed)
                    HIP_TRY(h, hipStreamSynchronize(h->stream));
                    h->d_wregion[li].release();
                    h->d_wregion[li] = nb;
                }
                h->wregion_cap = ncap;
            }
            uint32_t d = 0;
            for (uint32_t a = 0; a < P.naggs; a++) {
                if (!P.aggs[a].distinct) continue;
                const uint32_t li = P.aggs[a].log_index;
                L.region[d] = h->d_wregion[li].p;
                L.region_cursor[d] = h->d_wcursor.p + (size_t)li * kWordSubs * kCursorStride;
                L.over_word[d] = A.log_word[li];
                L.log_key[d] = A.log_key[li];
                L.log_val[d] = A.log_val[li];
                L.log_cls[d] = A.log_cls[li];
                L.log_index[d] = li;
                d++;
            }
            L.region_cap = h->wregion_cap;
            L.over_cursor = A.word_cursor;
            L.over_hist = A.word_hist;
            L.over_capacity = A.log_capacity;
            L.log_cursor = A.log_cursor;
            L.log_capacity = A.log_capacity;
            L.nw_key_bits = h->nw_key_bits;
            L.nw_val_bits = h->nw_val_bits;
            L.dcache_slots = (h->opt_spec_debug & 4u) ? 0u : dcache_slots;
            L.pad = h->opt_spec_debug;
            h->wregion_used = true;
        }
        hipEvent_t e0 = get_event(h), e1 = get_event(h);
        if (e0) (void)hipEventRecord(e0, h->stream);
        const uint64_t chunk = 1ull << 31;  // 32-bit row indices inside one launch
        for (uint64_t off = 0; off < b->nrows; off += chunk) {
            uint64_t n = std::min<uint64_t>(chunk, b->nrows - off);
            F.row_base = h->row_base + off;
            bool aligned = true;
            for (uint32_t c = 0; c < F.ncols; c++) {
                F.cols[c] = P.cols[c];
                if (F.cols[c].tags) F.cols[c].tags += off;
                if (F.cols[c].payload) F.cols[c].payload += off;
                if (F.cols[c].codes) F.cols[c].codes += off;
                aligned &= ((uintptr_t)F.cols[c].tags % 2 == 0) && ((uintptr_t)F.cols[c].payload % 16 == 0) &&
                           ((uintptr_t)F.cols[c].codes % 8 == 0);
            }
            if (spec || jit) {
                // WIDE launch over the even prefix (2 adjacent rows per lane and load), scalar launch for an odd last row
                bool wide = aligned && h->opt_wide && n >= 2;
                // (a row count that lives on the device may be odd: the kernel masks the last item's second row itself)
                uint64_t n_main = wide && !h->push_nrows_dev && !h->push_nseg ? (n & ~1ull) : n;
                F.nrows = (uint32_t)n_main;
                uint64_t items = wide ? (n_main + 1) / 2 : n_main;
                uint32_t rpl = wide ? 2 : 4;
                uint64_t tiles = (items + (uint64_t)fblock * rpl - 1) / ((uint64_t)fblock * rpl);
                if (ndist) tiles = (tiles + 3) / 4;  // a workgroup reserves chunks in every hash region: give it a few tiles to fill them
                uint32_t g = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(fgrid, tiles));
                F.slabs = nullptr;
                if (use_slabs && g > 1) {
                    HIP_TRY(h, h->d_slabs.ensure((size_t)g * P.lds_words * F.lds_slots));
                    HIP_TRY(h, h->d_block_sel.ensure(g));
                    F.slabs = h->d_slabs.p;
                    F.block_selected = h->d_block_sel.p;
                }
                if (spec) HIP_TRY(h, spec->launch(P, F, h->table, h->d_counters.p + 1, g, fblock, wide, L, h->stream));
                else HIP_TRY(h, jit_launch(jit, P, F, h->table, h->d_counters.p + 1, g, wide, L, ndist, h->stream));
                if (F.slabs) {
                    // One-call execution of a small-table plan, and this merge is the query's last device work: its last
                    // workgroup runs the tail too (FinalGroup into pinned memory, the counters, the next execution's reopen).
                    TailArgs T{};
                    SmallTail tl;
                    const bool last_work = h->one_call && h->opt_tail_in_merge && n_main == n && off + n == b->nrows && !ndist &&
                                           !h->has_array_agg && h->push_nseg <= 1 && !h->push_nrows_dev;
                    if (last_work && small_tail_layout(h, tl) && tl.fused && ensure_pinned_counters(h) == N1K_OK &&
                        small_tail_pinned(h, tl) == N1K_OK) {
                        if (!h->d_merge_done.p) {
                            HIP_TRY(h, h->d_merge_done.ensure(4));
                            HIP_TRY(h, hipMemsetAsync(h->d_merge_done.p, 0, 16, h->stream));
                        }
                        char* d = h->pin_out;
                        tl.clear = !h->prog.wide_int;
                        T.out_keys = (OutValue*)d;
                        T.out_aggs = (OutValue*)(d + tl.off_aggs);
                        T.out_parts = (OutPartial*)(d + tl.off_parts);
                        T.out_rep = (uint64_t*)(d + tl.off_rep);
                        T.counters = h->d_counters.p;
                        T.host_counters = (unsigned long long*)(h->pin_out + tl.total);
                        T.max_out = tl.spec_groups;
                        T.done = h->d_merge_done.p;
                        T.clear = tl.clear ? 1u : 0u;
                        T.enabled = 1;
                        h->tail_done = tl;
                        h->tail_in_merge = true;
                    }
                    HIP_TRY(h, launch_merge_slabs(P, F, h->table, g, h->d_counters.p + 1, h->stream, h->opt_merge_chunks, T.enabled ? &T : nullptr));
                }
                F.slabs = nullptr;
                if (n_main < n) {
                    for (uint32_t c = 0; c < F.ncols; c++) {
                        if (F.cols[c].tags) F.cols[c].tags += n_main;
                        if (F.cols[c].payload) F.cols[c].payload += n_main;
                        if (F.cols[c].codes) F.cols[c].codes += n_main;
                    }
                    F.nrows = (uint32_t)(n - n_main);
                    F.row_base += n_main;
                    if (spec) HIP_TRY(h, spec->launch(P, F, h->table, h->d_counters.p + 1, 1, fblock, false, L, h->stream));
                    else HIP_TRY(h, jit_launch(jit, P, F, h->table, h->d_counters.p + 1, 1, false, L, ndist, h->stream));
                }
                continue;
            }
            F.nrows = (uint32_t)n;
            uint64_t tiles = (n + (uint64_t)fblock * frpl - 1) / ((uint64_t)fblock * frpl);
            uint32_t g = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(fgrid, tiles));
            F.slabs = nullptr;
            if (use_slabs && g > 1) {
                HIP_TRY(h, h->d_slabs.ensure((size_t)g * P.lds_words * F.lds_slots));
                HIP_TRY(h, h->d_block_sel.ensure(g));
                F.slabs = h->d_slabs.p;
                F.block_selected = h->d_block_sel.p;
            }
            HIP_TRY(h, launch_scan_fast(P, F, h->table, h->d_counters.p + 1, g, fblock, frpl, h->stream));
            if (F.slabs) HIP_TRY(h, launch_merge_slabs(P, F, h->table, g, h->d_counters.p + 1, h->stream, h->opt_merge_chunks));
            F.slabs = nullptr;
        }
        if (e1) (void)hipEventRecord(e1, h->stream);
        h->events.emplace_back(e0, e1);
        h->stats.agg_mode = F.hashed ? N1K_MODE_LDS_HASH : N1K_MODE_LDS_DIRECT;
        return N1K_OK;
    }
interpreter:
    // DIRECT: every key is dictionary coded and the whole key domain fits the LDS table -> perfect hash
    bool direct = h->opt_agg_mode != N1K_MODE_LDS_HASH;
    uint64_t domain = 1;
    for (uint32_t k = 0; k < P.nkeys && direct; k++) {
        if (P.keys[k].mode != KEYM_DICT) direct = false;
        uint64_t radix = (uint64_t)h->dict.size() + 2;
        A.direct_stride[k] = (uint32_t)domain;
        A.direct_radix[k] = (uint32_t)std::min<uint64_t>(radix, 0xFFFFFFFFull);
        domain *= radix;
        if (domain > max_slots) direct = false;
    }
    uint32_t slots = direct ? (uint32_t)std::max<uint64_t>(domain, 2) : max_slots;
    if (P.nkeys == 0) slots = 2;
    A.lds_slots = slots;
    A.lds_max_fill = std::max(1u, (uint32_t)((uint64_t)slots * 5 / 8));
    A.err_flags = h->d_errp;
    A.rows_selected = h->d_counters.p + 0;
    uint64_t tile_rows = (uint64_t)block * rpl;
    uint64_t ntiles = (b->nrows + tile_rows - 1) / tile_rows;
    uint32_t per_cu = block == 1024 ? 1 : (block == 512 ? 2 : 4);
    uint32_t grid = h->opt_grid_blocks ? h->opt_grid_blocks : (uint32_t)(h->num_cus * per_cu);
    grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(grid, ntiles));
    hipEvent_t e0 = get_event(h), e1 = get_event(h);
    if (e0) (void)hipEventRecord(e0, h->stream);
    if (h->push_nseg > 1) {
        // a segmented batch on the interpreter: one launch per segment, its row count read on the device
        Program Ps = P;
        for (uint32_t sg = 0; sg < h->push_nseg; sg++) {
            const uint64_t off = (uint64_t)sg * h->push_seg_rows;
            for (uint32_t c = 0; c < P.ncols; c++) {
                Ps.cols[c] = P.cols[c];  // (derived columns too: they were evaluated over the whole capacity, row for row)
                if (Ps.cols[c].tags) Ps.cols[c].tags += off;
                if (Ps.cols[c].payload) Ps.cols[c].payload += off;
                if (Ps.cols[c].codes) Ps.cols[c].codes += off;
            }
            ScanArgs As = A;
            As.nrows = h->push_seg_rows;
            As.nrows_dev = h->push_seg_counts + (size_t)sg * kCursorStride;
            const uint64_t nt = (h->push_seg_rows + tile_rows - 1) / tile_rows;
            const uint32_t g = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(grid, nt));
            HIP_TRY(h, launch_scan_group(Ps, As, h->table, h->d_counters.p + 1, g, block, rpl, direct, h->stream));
        }
    } else
        HIP_TRY(h, launch_scan_group(P, A, h->table, h->d_counters.p + 1, grid, block, rpl, direct, h->stream));
    if (e1) (void)hipEventRecord(e1, h->stream);
    h->events.emplace_back(e0, e1);
    h->stats.agg_mode = direct ? N1K_MODE_LDS_DIRECT : N1K_MODE_LDS_HASH;
    return N1K_OK;
}

n1k_status run_filter_batch(n1k_handle* h, const n1k_batch* b) {
    Program& P = h->prog;
    if (b->nrows == 0) return N1K_OK;
    hipEvent_t e0 = get_event(h), e1 = get_event(h);
    unsigned long long total = 0;
    if (h->opt_filter_stream) {
        // ONE pass (filter_stream_kernel): predicate, ordered compaction and the tiles' offsets by a chained scan; the ordinals
        // land in a buffer sized for every row, the host reads the count and copies that many
        const uint64_t ntiles = (b->nrows + kFilterStreamTile - 1) / kFilterStreamTile;
        HIP_TRY(h, h->d_tile_off.ensure(ntiles + 1));
        HIP_TRY(h, h->d_sel.ensure(b->nrows));
        uint32_t grid = (uint32_t)std::min<uint64_t>(ntiles, (uint64_t)h->num_cus * 4);  // (124 VGPRs: four 256-thread workgroups per CU)
        // one comparison of a TAGGED64 column with a NUMBER constant, arrays aligned for two rows per load: the wide variant
        bool fast = h->opt_wide && P.nlogic == 1 && P.logic[0].op == LOGIC_PUSH;
        if (fast) {
            const Term& t = P.terms[P.logic[0].arg];
            fast = t.op >= TERM_NUM_LT && t.op <= TERM_NUM_EQ && !t.a.is_const && t.a.col < (uint32_t)h->plan.paths.size() &&
                   P.cols[t.a.col].kind == COLK_TAGGED64 && (uintptr_t)P.cols[t.a.col].payload % 16 == 0 && (uintptr_t)P.cols[t.a.col].tags % 2 == 0;
        }
        if (fast) grid = (uint32_t)std::min<uint64_t>(ntiles, (uint64_t)h->num_cus * (h->opt_grid_blocks ? h->opt_grid_blocks : 5));  // (84 VGPRs; measured 4: 0.43, 5: 0.39, 6: 0.40, 8: 0.40 ms per 100 M rows)
        if (e0) (void)hipEventRecord(e0, h->stream);
        HIP_TRY(h, launch_filter_stream(P, b->nrows, h->row_base, h->d_sel.p, (unsigned long long*)h->d_tile_off.p,
                                        (unsigned long long*)h->d_tile_off.p + ntiles, h->d_counters.p + 3, h->d_errp, grid, h->stream, fast));
        h->stats.spec_kernel = fast ? 1u : 0u;
        if (e1) (void)hipEventRecord(e1, h->stream);  // device time excludes the PCIe copy of the ordinals
        HIP_TRY(h, hipMemcpyAsync(&total, h->d_counters.p + 3, sizeof total, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        if (total) {
            size_t old = h->selected.size();
            h->selected.resize(old + total);
            HIP_TRY(h, hipMemcpyAsync(h->selected.data() + old, h->d_sel.p, total * 8, hipMemcpyDeviceToHost, h->stream));
        }
        h->events.emplace_back(e0, e1);
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        h->stats.rows_selected += total;
        return N1K_OK;
    }
    // three kernels (ablation, option filter_stream = 0): ballot mask + per-tile counts, single-workgroup scan, compaction
    uint64_t ntiles = (b->nrows + kFilterTile - 1) / kFilterTile;
    HIP_TRY(h, h->d_mask.ensure(ntiles * (kFilterTile / 64)));
    HIP_TRY(h, h->d_tile_cnt.ensure(ntiles));
    HIP_TRY(h, h->d_tile_off.ensure(ntiles));
    uint64_t nchunks = (b->nrows + 1023) / 1024;
    uint32_t grid = (uint32_t)std::min<uint64_t>(nchunks, (uint64_t)h->num_cus * 8);
    if (e0) (void)hipEventRecord(e0, h->stream);
    HIP_TRY(h, hipMemsetAsync(h->d_tile_cnt.p, 0, ntiles * sizeof(uint32_t), h->stream));
    HIP_TRY(h, launch_filter_mask(P, b->nrows, h->d_mask.p, h->d_tile_cnt.p, h->d_errp, grid, h->stream));
    HIP_TRY(h, launch_tile_scan(h->d_tile_cnt.p, h->d_tile_off.p, ntiles, h->d_counters.p + 3, h->stream));
    HIP_TRY(h, hipMemcpyAsync(&total, h->d_counters.p + 3, sizeof total, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (total) {
        HIP_TRY(h, h->d_sel.ensure(total));
        uint32_t cgrid = (uint32_t)std::min<uint64_t>(ntiles, (uint64_t)h->num_cus * 8);
        HIP_TRY(h, launch_filter_compact(h->d_mask.p, h->d_tile_off.p, b->nrows, h->row_base, h->d_sel.p, cgrid, h->stream));
        if (e1) (void)hipEventRecord(e1, h->stream);  // device time excludes the PCIe copy of the ordinals
        size_t old = h->selected.size();
        h->selected.resize(old + total);
        HIP_TRY(h, hipMemcpyAsync(h->selected.data() + old, h->d_sel.p, total * 8, hipMemcpyDeviceToHost, h->stream));
    } else if (e1) {
        (void)hipEventRecord(e1, h->stream);
    }
    h->events.emplace_back(e0, e1);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->stats.rows_selected += total;
    return N1K_OK;
}

// Point the program at this batch's input columns and evaluate the arithmetic nodes into derived columns (defer: not
// yet — run_group_batch first looks for a kernel that evaluates them in registers, and materialises them otherwise).
n1k_status bind_columns(n1k_handle* h, const n1k_batch* b, bool defer) {
    Program& P = h->prog;
    // first push: give the plan's string constants their dictionary codes
    auto resolve = [&](Operand& o) {
        if (o.is_const && o.ctag == T_STRING && o.pad == 1) {
            o.cpayload = intern(h, h->const_strings[(size_t)o.cpayload]);
            o.pad = 0;
        }
    };
    for (uint32_t t = 0; t < P.nterms; t++) { resolve(P.terms[t].a); resolve(P.terms[t].b); resolve(P.terms[t].c); }
    for (uint32_t k = 0; k < P.nkeys; k++) resolve(P.keys[k].src);
    for (uint32_t a = 0; a < P.naggs; a++) resolve(P.aggs[a].src);
    for (auto& d : h->derived)
        for (uint32_t k = 0; k < d.nops; k++) resolve(d.ops[k]);
    const uint32_t ni = (uint32_t)h->plan.paths.size();
    for (uint32_t c = 0; c < ni; c++) {
        P.cols[c].kind = b->cols[c].kind == N1K_COL_DICT32 ? COLK_DICT32 : COLK_TAGGED64;
        P.cols[c].tags = b->cols[c].tags;
        P.cols[c].payload = b->cols[c].payload;
        P.cols[c].codes = b->cols[c].codes;
    }
    bool tagged_key = false;
    for (uint32_t k = 0; k < P.nkeys; k++) tagged_key |= P.keys[k].mode == KEYM_TAGGED;  // (fix_layout came first)
    if (tagged_key) {  // FLOAT group keys that are NaN / +-Inf are the strings they marshal to (n1k_device.h pack_key_field)
        P.nan_code = intern(h, "NaN");
        P.pinf_code = intern(h, "+Infinity");
        P.ninf_code = intern(h, "-Infinity");
    }
    P.dict_size = (uint32_t)h->dict.size();
    P.empty_str_code = lookup_code(h, "");
    P.empty_arr_code = lookup_code(h, "[]");
    P.empty_obj_code = lookup_code(h, "{}");
    h->derived_ready = h->derived.empty();
    if (h->derived_ready) return N1K_OK;
    for (size_t i = 0; i < h->derived.size(); i++) {
        DevCol& d = P.cols[ni + i];
        d.kind = COLK_TAGGED64;
        d.tags = nullptr;
        d.payload = nullptr;
        d.codes = nullptr;
    }
    return defer ? N1K_OK : materialize_derived(h, b);
}

// one element-wise arith_kernel launch per arithmetic node: the node's values as a TAGGED64 column in HBM
n1k_status materialize_derived(n1k_handle* h, const n1k_batch* b) {
    Program& P = h->prog;
    if (h->derived_ready) return N1K_OK;
    const uint32_t ni = (uint32_t)h->plan.paths.size();
    for (const auto& d : h->derived)
        if (d.op >= AR_GREATEST) {  // GREATEST / LEAST collate strings by rank: the table must cover this batch's dictionary
            n1k_status rst = ensure_rank(h);
            if (rst != N1K_OK) return rst;
            break;
        }
    h->dv_tags.resize(h->derived.size());
    h->dv_payload.resize(h->derived.size());
    for (size_t i = 0; i < h->derived.size(); i++) {
        // an earlier launch may still read the previous batch's derived columns
        if (h->dv_tags[i].n < b->nrows) HIP_TRY(h, hipStreamSynchronize(h->stream));
        HIP_TRY(h, h->dv_tags[i].ensure(std::max<uint64_t>(b->nrows, 1)));
        HIP_TRY(h, h->dv_payload[i].ensure(std::max<uint64_t>(b->nrows, 1)));
        ArithArgs A{};
        A.op = h->derived[i].op;
        A.nops = h->derived[i].nops;
        for (uint32_t k = 0; k < A.nops; k++) A.ops[k] = h->derived[i].ops[k];
        for (uint32_t c = 0; c < ni + i; c++) A.cols[c] = P.cols[c];
        A.nrows = b->nrows;
        A.out_tags = h->dv_tags[i].p;
        A.out_payload = h->dv_payload[i].p;
        A.str_rank = h->d_rank.p;
        A.err_flags = h->d_errp;
        HIP_TRY(h, launch_arith(A, h->stream));
        DevCol& d = P.cols[ni + i];
        d.kind = COLK_TAGGED64;
        d.tags = A.out_tags;
        d.payload = A.out_payload;
        d.codes = nullptr;
    }
    h->derived_ready = true;
    return N1K_OK;
}

n1k_status push_device(n1k_handle* h, const n1k_batch* b) {
    if (h->stop_flag.load()) return fail(h, N1K_STOPPED, "operator was stopped");
    h->device_clean = false;
    n1k_status st = ensure_device(h);
    if (st != N1K_OK) return st;
    st = validate_batch(h, b);
    if (st != N1K_OK) return st;
    if (!h->layout_fixed) {
        st = fix_layout(h, b);
        if (st != N1K_OK) return st;
    }
    // High-cardinality GROUP BY: beyond a few ten thousand groups the scan's LDS stage absorbs nothing and every row
    // costs atomics on a table in HBM.  Whether a batch is like that is learnt from the data: the first rows of a
    // large batch run through the scan kernels; if they bring many new groups, the rest is partitioned (below).
    st = flush_pending(h);  // a region kept from the previous batch joins the table before more rows arrive
    if (st != N1K_OK) return st;
    const bool first_rows = h->row_base == 0 && h->merged_groups_bound == 0;  // nothing in the handle yet
    PartitionPlan pp;
    const bool can_partition = h->plan.has_group && !h->push_nrows_dev && !h->push_nseg && partition_eligible(h, pp);
    uint64_t head = b->nrows;
    bool decide = false;
    if (can_partition && h->opt_agg_mode == N1K_MODE_PARTITIONED) head = 0;
    else if (can_partition && h->opt_agg_mode == N1K_MODE_AUTO && b->nrows >= h->opt_partition_min_rows && !small_key_domain(h)) {
        head = std::min<uint64_t>(b->nrows, h->opt_partition_probe_rows);
        decide = true;
    }
    auto view = [&](uint64_t off, uint64_t n, std::vector<n1k_col>& cols, n1k_batch& v) {
        cols.assign(b->cols, b->cols + b->ncols);
        for (auto& c : cols) {
            if (c.tags) c.tags += off;
            if (c.payload) c.payload += off;
            if (c.codes) c.codes += off;
        }
        v.nrows = n;
        v.ncols = b->ncols;
        v.cols = cols.data();
    };
    std::vector<n1k_col> cols;
    n1k_batch v{};
    view(0, b->nrows, cols, v);
    // (the probe, the partitioned path and the Filter-only kernels read derived columns; run_group_batch decides itself)
    const bool defer = h->plan.has_group && !decide && !(can_partition && head == 0);
    st = bind_columns(h, &v, defer);
    if (st != N1K_OK) return st;
    st = ensure_rank(h);
    if (st != N1K_OK) return st;
    bool partition = can_partition && head == 0 && b->nrows > 0;
    uint64_t groups_est = b->nrows;
    if (decide && first_rows && h->opt_partition_sticky && h->sticky.valid && b->nrows >= h->sticky.rows / 2 && b->nrows <= h->sticky.rows * 2) {
        decide = false;  // as the last execution over a batch of this size went: no probe (see n1k_handle::sticky)
        partition = true;
        groups_est = std::min<uint64_t>(b->nrows, h->sticky.groups_est * b->nrows / std::max<uint64_t>(h->sticky.rows, 1) + 1024);
    }
    if (decide) {
        // probe: Filter + group key of the first `head` rows into the table (keys only), counted before and after
        st = ensure_table(h, head);
        if (st != N1K_OK) return st;
        unsigned long long before = 0, after = 0;
        HIP_TRY(h, hipMemcpyAsync(&before, h->d_counters.p + 1, sizeof before, hipMemcpyDeviceToHost, h->stream));
        const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)h->num_cus * 4, (head + 1023) / 1024));
        HIP_TRY(h, launch_probe_keys(h->prog, head, h->table, h->d_errp, h->d_counters.p + 1, grid, h->stream));
        HIP_TRY(h, hipMemcpyAsync(&after, h->d_counters.p + 1, sizeof after, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        const uint64_t fresh = after > before ? after - before : 0;
        partition = fresh >= h->opt_partition_min_groups;
        // how many groups will the batch bring?  If the keys are draws from a universe of U values, m draws show
        // d = U (1 - e^(-m/U)) of them: solve for U from the probe (m = head rows, d = fresh groups) and evaluate at
        // the batch.  (All-distinct probes have no finite U: the row count stays the bound.)  Only the number of
        // partition passes and the size of the per-bin LDS tables hang on it; a low guess costs speed, not results.
        const double m = (double)head, d = (double)std::max<uint64_t>(1, fresh);
        if (d < 0.98 * m) {
            double lo = d, hi = 1e18;
            for (int it = 0; it < 200; it++) {
                const double U = std::sqrt(lo * hi);
                if (U * (1.0 - std::exp(-m / U)) < d) lo = U; else hi = U;
            }
            const double U = lo, nn = (double)b->nrows;
            groups_est = std::min<uint64_t>(groups_est, (uint64_t)(2.0 * U * (1.0 - std::exp(-nn / U))) + 1024);
        }
        if (partition && first_rows && h->table.capacity) {
            // the probe's keys are all the handle holds: drop them, so that the partitioned path's groups can stay in
            // their compact region (no table at all for this query)
            HIP_TRY(h, launch_init_table(h->prog, h->table, 0, h->table.capacity, nullptr, h->stream));
            HIP_TRY(h, hipMemsetAsync(h->d_counters.p + 1, 0, sizeof(unsigned long long), h->stream));
        }
    }
    if (b->nrows) {
        if (!h->plan.has_group) st = run_filter_batch(h, &v);
        else if (partition) {
            bool done = false;
            st = run_group_records(h, &v, pp, groups_est, first_rows, &done);
            h->sticky.valid = st == N1K_OK && done && first_rows && h->opt_agg_mode == N1K_MODE_AUTO;
            h->sticky.rows = b->nrows;
            h->sticky.groups_est = groups_est;
            if (st == N1K_OK && !done) st = run_group_partitioned(h, &v, pp, groups_est, first_rows);
        } else
            st = run_group_batch(h, &v);
        if (st != N1K_OK) return st;
    }
    h->row_base += b->nrows;
    h->stats.rows_in += b->nrows;
    h->stats.batches += 1;
    h->stats.bytes_scanned += b->nrows * batch_bytes_per_row(h);
    return N1K_OK;
}

// host columns -> the handle's staging buffers on the device (the caller's memory is not retained after return: cgo rule)
n1k_status stage_host_batch(n1k_handle* h, const n1k_batch* batch, std::vector<n1k_col>& dcols) {
    uint32_t nc = batch->ncols;
    const int set = h->st_cur;
    auto& s_tags = h->st_tags[set];
    auto& s_payload = h->st_payload[set];
    auto& s_codes = h->st_codes[set];
    s_tags.resize(std::max<size_t>(s_tags.size(), nc));
    s_payload.resize(std::max<size_t>(s_payload.size(), nc));
    s_codes.resize(std::max<size_t>(s_codes.size(), nc));
    if (!h->copy_stream) {
        HIP_TRY(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
        HIP_TRY(h, hipEventCreateWithFlags(&h->st_copied, hipEventDisableTiming));
        for (int i = 0; i < 2; i++) HIP_TRY(h, hipEventCreateWithFlags(&h->st_free[i], hipEventDisableTiming));
    }
    // the kernels of the batch before last may still read this set: the COPIES wait for them on the device, the host
    // does not (a buffer that has to grow is freed by hipFree, which waits for the device itself)
    if (h->st_busy[set]) HIP_TRY(h, hipStreamWaitEvent(h->copy_stream, h->st_free[set], 0));
    dcols.assign(nc, n1k_col{});
    uint64_t n = batch->nrows;
    for (uint32_t c = 0; c < nc; c++) {
        const n1k_col& col = batch->cols[c];
        dcols[c] = col;
        if (col.kind == N1K_COL_DICT32) {
            HIP_TRY(h, s_codes[c].ensure(n));
            if (n) HIP_TRY(h, hipMemcpyAsync(s_codes[c].p, col.codes, n * 4, hipMemcpyHostToDevice, h->copy_stream));
            dcols[c].codes = s_codes[c].p;
        } else {
            HIP_TRY(h, s_tags[c].ensure(n));
            HIP_TRY(h, s_payload[c].ensure(n));
            if (n) {
                HIP_TRY(h, hipMemcpyAsync(s_tags[c].p, col.tags, n, hipMemcpyHostToDevice, h->copy_stream));
                HIP_TRY(h, hipMemcpyAsync(s_payload[c].p, col.payload, n * 8, hipMemcpyHostToDevice, h->copy_stream));
            }
            dcols[c].tags = s_tags[c].p;
            dcols[c].payload = s_payload[c].p;
        }
    }
    // the caller's memory is not retained after return (cgo rule): wait for the copies — not for the compute stream
    HIP_TRY(h, hipEventRecord(h->st_copied, h->copy_stream));
    HIP_TRY(h, hipStreamWaitEvent(h->stream, h->st_copied, 0));
    HIP_TRY(h, hipEventSynchronize(h->st_copied));
    return N1K_OK;
}

// behind the kernels of a staged batch: its set may be overwritten once this event has passed
n1k_status staged_batch_issued(n1k_handle* h) {
    const int set = h->st_cur;
    HIP_TRY(h, hipEventRecord(h->st_free[set], h->stream));
    h->st_busy[set] = true;
    h->st_cur ^= 1;
    return N1K_OK;
}

}  // namespace n1k_eng
