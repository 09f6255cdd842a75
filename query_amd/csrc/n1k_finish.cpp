// n1k_finish.cpp — afterItems: IntermediateGroup / FinalGroup over what the batches left on the device, then the grouped tail
// (execution/group_intermediate.go:56-104, group_final.go:55-118).
#include "n1k_engine.h"

using namespace n1k;
using namespace n1k_eng;

// stats.query_ms: the last device work of the query is behind this point of the stream (the wait that follows completes it)
static void mark_query_end(n1k_handle* h) {
    if (h->q0_recorded && h->ev_q1 && hipEventRecord(h->ev_q1, h->stream) == hipSuccess) h->q1_recorded = true;
}

namespace n1k_eng {

// The speculative FinalGroup of a small table (n1k_finish below; with the one-call path the merge kernel's last workgroup runs
// it, n1k_scan.cpp): whether the plan and the handle's state allow it, and where its pieces land in the pinned buffer.
bool small_tail_layout(n1k_handle* h, SmallTail& t) {
    const ParsedPlan& pl = h->plan;
    const uint32_t nk = (uint32_t)pl.keys.size(), na = (uint32_t)pl.aggs.size();
    const bool topk_forced = pl.has_order && pl.limit >= 0 && !pl.has_having && h->opt_topk_min_groups < 4096;  // tests
    t.ok = pl.has_group && !h->has_distinct && h->table.capacity && h->table.capacity <= (1u << 20) && !topk_forced && !h->pending.count;
    if (!t.ok) return false;
    t.spec_groups = std::min<uint64_t>(h->table.capacity, 4096);
    t.off_aggs = t.spec_groups * nk * sizeof(OutValue);
    t.off_parts = t.off_aggs + t.spec_groups * na * sizeof(OutValue);
    t.off_rep = t.off_parts + t.spec_groups * na * sizeof(OutPartial);
    t.total = t.off_rep + t.spec_groups * 8;
    t.fused = h->opt_pinned_out && h->opt_fused_tail && h->table.capacity <= 8192;
    return true;
}

n1k_status small_tail_pinned(n1k_handle* h, const SmallTail& t) {
    if (h->pin_cap < t.total + kCounters * sizeof(unsigned long long)) {
        if (h->pin_out) (void)hipHostFree(h->pin_out);
        h->pin_out = nullptr;
        h->pin_cap = 0;
        HIP_TRY(h, hipHostMalloc((void**)&h->pin_out, t.total + kCounters * sizeof(unsigned long long), hipHostMallocDefault));
        h->pin_cap = t.total + kCounters * sizeof(unsigned long long);
    }
    return N1K_OK;
}

}  // namespace n1k_eng

extern "C" {

n1k_status n1k_finish(n1k_handle* h, n1k_result* out) {
    return guarded(h, [&]() -> n1k_status {
    if (!h || !out) return N1K_INVALID;
    memset(out, 0, sizeof *out);
    h->failure_global = false;
    if (h->stop_flag.load()) return fail(h, N1K_STOPPED, "operator was stopped");
    std::chrono::steady_clock::time_point wait_end;
    bool waited = false;
    struct AfterWait {  // (host trace: what n1k_finish does behind its wait)
        n1k_handle* h; const std::chrono::steady_clock::time_point* t; const bool* on;
        ~AfterWait() { if (*on) h->host_us[4] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - *t).count(); }
    } after_wait{h, &wait_end, &waited};
    const ParsedPlan& pl = h->plan;
    uint32_t nk = (uint32_t)pl.keys.size(), na = (uint32_t)pl.aggs.size();
    out->nkeys = nk;
    out->naggs = na;
    uint32_t err_flags = 0;
    unsigned long long counters[kCounters] = {0};
    const size_t rec_keys = (size_t)nk * sizeof(OutValue), rec_aggs = (size_t)na * sizeof(OutValue),
                 rec_parts = (size_t)na * sizeof(OutPartial);
    // Speculative FinalGroup: when the plan has no DISTINCT step the finalize kernel does not depend on anything the
    // host has to read first, so it is launched for up to `spec_groups` groups together with the copy of the
    // counters: ONE host synchronisation per query when the result fits (else the sized pass below runs as well).
    uint64_t spec_groups = 0;
    if (h->device_ready) {
        HIP_TRY(h, hipSetDevice(h->device));
        {
            n1k_status pst = ensure_pinned_counters(h);
            if (pst != N1K_OK) return pst;
        }
        // (a table of millions of slots is not worth scanning twice: the sized pass alone then)
        SmallTail tl;
        if (h->tail_in_merge ? (tl = h->tail_done, true) : small_tail_layout(h, tl)) {
            spec_groups = tl.spec_groups;
            const size_t off_aggs = tl.off_aggs, off_parts = tl.off_parts, off_rep = tl.off_rep, total = tl.total;
            const bool fused_tail = tl.fused;
            if (!h->tail_in_merge) {
                HIP_TRY(h, h->d_out.ensure(total + 16));
                n1k_status ps = small_tail_pinned(h, tl);
                if (ps != N1K_OK) return ps;
            }
            if (h->tail_in_merge) {
                // the merge kernel's last workgroup has run the tail already (n1k_scan.cpp): nothing to launch
                h->tail_in_merge = false;
                h->out_count_dirty = !tl.clear;
                h->device_clean = tl.clear;
            } else if (fused_tail) {
                // small tables: FinalGroup, the counters behind it and — for a one-call execution, when nothing else on the
                // device needs a reset (no wide-value tables) — the state the next execution starts from, in ONE last kernel
                const bool clear = h->clear_on_finish && !h->prog.wide_int;
                char* d = h->pin_out;
                HIP_TRY(h, launch_finalize_small(h->prog, h->table, (OutValue*)d, (OutValue*)(d + off_aggs), (OutPartial*)(d + off_parts),
                                                 (uint64_t*)(d + off_rep), h->d_counters.p, (unsigned long long*)(h->pin_out + total), spec_groups,
                                                 h->d_errp, clear, h->stream));
                h->out_count_dirty = !clear;
                h->device_clean = clear;
            } else {
            if (h->out_count_dirty) HIP_TRY(h, hipMemsetAsync(h->d_counters.p + 2, 0, sizeof(unsigned long long), h->stream));
            h->out_count_dirty = true;  // (reopen zeroes every counter in its one launch)
            if (h->opt_pinned_out) {
                // The few groups of a speculative FinalGroup are written by the kernel straight into the pinned host buffer
                // (posted stores over PCIe) and a one-wave kernel publishes the counters behind them: no copy engine in the
                // query's critical path (two hipMemcpyAsync D2H cost ~ 21 us of a 0.33 ms query: 2 x 4.7 us + a 12 us gap).
                char* d = h->pin_out;
                HIP_TRY(h, launch_finalize(h->prog, h->table, (OutValue*)d, (OutValue*)(d + off_aggs), (OutPartial*)(d + off_parts),
                                           (uint64_t*)(d + off_rep), h->d_counters.p + 2, spec_groups, h->d_errp, h->stream));
                HIP_TRY(h, launch_publish_counters(h->d_counters.p, (unsigned long long*)(h->pin_out + total), kCounters, h->stream));
            } else {
                char* d = h->d_out.p;
                HIP_TRY(h, launch_finalize(h->prog, h->table, (OutValue*)d, (OutValue*)(d + off_aggs), (OutPartial*)(d + off_parts),
                                           (uint64_t*)(d + off_rep), h->d_counters.p + 2, spec_groups, h->d_errp, h->stream));
                HIP_TRY(h, hipMemcpyAsync(h->pin_out, d, total, hipMemcpyDeviceToHost, h->stream));
                HIP_TRY(h, hipMemcpyAsync(h->pin_out + total, h->d_counters.p, sizeof counters, hipMemcpyDeviceToHost, h->stream));
            }
            }
            mark_query_end(h);
            {
                const auto w0 = std::chrono::steady_clock::now();
                HIP_TRY(h, hipStreamSynchronize(h->stream));
                wait_end = std::chrono::steady_clock::now();
                h->host_us[3] += std::chrono::duration<double, std::micro>(wait_end - w0).count();
                waited = true;
            }
            memcpy(counters, h->pin_out + total, sizeof counters);
        } else {
            HIP_TRY(h, hipMemcpyAsync(h->pin_counters, h->d_counters.p, sizeof counters, hipMemcpyDeviceToHost, h->stream));
            mark_query_end(h);
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            memcpy(counters, h->pin_counters, sizeof counters);
        }
        err_flags = (uint32_t)counters[12];
        drain_events(h);
    } else if (h->stats.rows_in == 0) {
        // no batch was ever pushed: nothing ran on the device; only the empty-input row can be produced
        n1k_status st = ensure_device(h);
        if (st != N1K_OK) return st;
    }
    if (!pl.has_group) {
        if (err_flags & ERR_UNSUPPORTED_VALUE)
            return fail(h, N1K_UNSUPPORTED_DATA, "a value outside the device subset was met (ordering of arrays/objects)");
        out->nselected = h->selected.size();
        out->selected = h->selected.data();
        h->stats.groups_out = 0;
        return N1K_OK;
    }
    h->stats.rows_selected = counters[0];
    h->stats.wide_key_values = counters[13];
    h->stats.distinct_path = 0;
    uint64_t ng = h->pending.count ? h->pending.count : counters[1];  // (a kept region: the table is empty)
    h->r_keys.clear();
    h->r_aggs.clear();
    h->r_parts.clear();
    h->r_rep.clear();
    bool sets_exact = false, sets_deferred = false;  // the optimistic COUNT(DISTINCT) path reports failure with the results
redo_sets:
    sets_deferred = false;
    if (ng > 0 && h->has_distinct) {
        // K6: de-duplicate the logged (group, value) pairs of every DISTINCT aggregate (≙ Set.Len(), value/set.go:198-215)
        h->distinct_path = 0;
        if (h->wregion_used) HIP_TRY(h, hipMemsetAsync(h->d_counters.p + 20, 0, 8, h->stream));
        for (uint32_t a = 0; a < na; a++) {
            const AggSpec& ag = h->prog.aggs[a];
            if (!ag.distinct || ag.kind == AGG_ARRAY) continue;  // (ARRAY_AGG: after FinalGroup, array_agg_groups)
            const uint64_t npairs = std::min<uint64_t>(counters[8 + ag.log_index], h->log_capacity);
            const uint64_t nwords = h->distinct_words[ag.log_index] ? std::min<uint64_t>(counters[16 + ag.log_index], h->log_capacity) : 0;
            DistinctArgs D{};
            D.log_key = h->d_log_key[ag.log_index].p;
            D.log_val = h->d_log_val[ag.log_index].p;
            D.log_cls = h->d_log_cls[ag.log_index].p;
            D.npairs = npairs;
            D.glob_off = ag.glob_off;
            D.kind = ag.kind;
            D.total_words = h->d_counters.p + 5;
            HIP_TRY(h, h->d_regions.ensure(h->table.capacity * 6));
            D.regions = h->d_regions.p;
            HIP_TRY(h, hipMemsetAsync(h->d_counters.p + 5, 0, sizeof(unsigned long long), h->stream));
            HIP_TRY(h, launch_distinct_layout(h->prog, h->table, D, h->stream));  // also zeroes the set sizes
            if (npairs) {
                // pairs of two words (floats, wide values, SUM/AVG DISTINCT): per-(group, class) sets in global memory
                unsigned long long words = 0;
                HIP_TRY(h, hipMemcpyAsync(&words, h->d_counters.p + 5, sizeof words, hipMemcpyDeviceToHost, h->stream));
                HIP_TRY(h, hipStreamSynchronize(h->stream));
                HIP_TRY(h, h->d_set_table.ensure(std::max<uint64_t>(words, 1)));
                D.set_table = h->d_set_table.p;
                if (words) HIP_TRY(h, hipMemsetAsync(h->d_set_table.p, 0xFF, words * 8, h->stream));
                HIP_TRY(h, launch_distinct_insert(h->prog, h->table, D, h->d_errp, h->stream));
                h->distinct_path |= 1u;
            }
            if (h->opt_spec_debug & 8u) continue;
            if (h->wregion_used && h->distinct_words[ag.log_index]) {
                n1k_status st = distinct_regions_finish(h, ag, nwords, sets_exact, &sets_deferred);
                if (st != N1K_OK) return st;
            } else if (nwords) {
                n1k_status st = distinct_words_finish(h, ag, nwords);
                if (st != N1K_OK) return st;
            }
        }
        h->stats.distinct_path = h->distinct_path;
    }
    if (ng > 0) {
        const bool spec_hit = spec_groups && ng <= spec_groups;
        const uint64_t lay = spec_hit ? spec_groups : ng;  // the arrays are laid out for `lay` groups
        const size_t off_aggs = lay * rec_keys, off_parts = off_aggs + lay * rec_aggs, off_rep = off_parts + lay * rec_parts;
        const size_t total = off_rep + lay * 8;
        const char* hp = h->pin_out;
        size_t o_aggs = off_aggs, o_parts = off_parts, o_rep = off_rep;  // layout of the host copy
        h->stats.topk_candidates = 0;
        if (!spec_hit) {
            const uint64_t keep = pl.limit >= 0 ? (uint64_t)pl.offset + (uint64_t)pl.limit : ng;
            const bool topk = pl.has_order && pl.limit >= 0 && !pl.has_having && pl.order[0].proj_index < 0 && keep > 0 && keep < ng &&
                              ng >= h->opt_topk_min_groups && ng < (1ull << 32);
            // groups kept in their compact region + a top-k filter: only the first ORDER BY term's value of every group is
            // written (16 B per group, not the whole output row), the candidates' rows are finalised after the selection
            const bool lean = topk && h->pending.count && h->opt_lean_topk;
            HIP_TRY(h, h->d_out.ensure((lean ? ng * sizeof(OutValue) : total) + 16));
            char* d = h->d_out.p;
            if (h->out_count_dirty) HIP_TRY(h, hipMemsetAsync(h->d_counters.p + 2, 0, sizeof(unsigned long long), h->stream));
            h->out_count_dirty = true;  // (reopen zeroes every counter in its one launch)
            if (lean) {
                // (the order image of the first ORDER BY term straight from FinalGroup: 8 B per group written, one kernel less)
                HIP_TRY(h, h->d_images.ensure(ng));
                n1k_status rst = ensure_rank(h);
                if (rst != N1K_OK) return rst;
                HIP_TRY(h, launch_finalize_region(h->prog, h->d_emit.p, h->pending.cap, ng, nullptr, nullptr, nullptr, nullptr, h->d_errp,
                                                  h->stream, nullptr, nullptr, pl.order[0].key_index >= 0,
                                                  (uint32_t)(pl.order[0].key_index >= 0 ? pl.order[0].key_index : pl.order[0].agg_index),
                                                  h->d_images.p, pl.order[0].desc));
            }
            else if (h->pending.count)
                HIP_TRY(h, launch_finalize_region(h->prog, h->d_emit.p, h->pending.cap, ng, (OutValue*)d, (OutValue*)(d + off_aggs),
                                                  (OutPartial*)(d + off_parts), (uint64_t*)(d + off_rep), h->d_errp, h->stream));
            else
                HIP_TRY(h, launch_finalize(h->prog, h->table, (OutValue*)d, (OutValue*)(d + off_aggs), (OutPartial*)(d + off_parts),
                                           (uint64_t*)(d + off_rep), h->d_counters.p + 2, ng, h->d_errp, h->stream));
            size_t copy_bytes = total;
            const char* src = d;
            if (topk) {
                // ORDER BY ... LIMIT: only the groups that can be among the first offset+limit rows leave the device
                const OrderTerm& t0 = pl.order[0];
                HIP_TRY(h, h->d_images.ensure(ng));
                HIP_TRY(h, h->d_cand.ensure(topk_cand_entries(ng)));
                HIP_TRY(h, h->d_topk.ensure(topk_state_bytes()));
                n1k_status rst = ensure_rank(h);
                if (rst != N1K_OK) return rst;
                const OutValue* vals = lean || t0.key_index >= 0 ? (const OutValue*)d : (const OutValue*)(d + off_aggs);
                // the threshold from a sample of the groups first (one small kernel instead of eight histogram passes); exact
                // whenever at least `keep` candidates come out, else the radix select over all images
                const bool sampled = h->opt_topk_sample && topk_can_sample(ng, keep);
                unsigned long long ncand = 0;
                for (int attempt = sampled ? 0 : 1; attempt < 2; attempt++) {
                    HIP_TRY(h, launch_topk_select(h->prog, vals, lean ? 1u : (t0.key_index >= 0 ? nk : na),
                                                  lean ? 0u : (uint32_t)(t0.key_index >= 0 ? t0.key_index : t0.agg_index), ng, t0.desc, keep,
                                                  h->d_images.p, h->d_topk.p, h->d_cand.p, h->stream, attempt == 0, lean || (attempt == 1 && sampled)));
                    HIP_TRY(h, hipMemcpyAsync(h->pin_counters + kCounters, h->d_topk.p + topk_ncand_offset(), sizeof ncand, hipMemcpyDeviceToHost, h->stream));
                    HIP_TRY(h, hipStreamSynchronize(h->stream));
                    ncand = h->pin_counters[kCounters];
                    if (ncand >= keep) break;
                }
                o_aggs = ncand * rec_keys;
                o_parts = o_aggs + ncand * rec_aggs;
                o_rep = o_parts + ncand * rec_parts;
                copy_bytes = o_rep + ncand * 8;
                HIP_TRY(h, h->d_out2.ensure(copy_bytes + 16));
                char* c = h->d_out2.p;
                if (lean)
                    HIP_TRY(h, launch_finalize_region(h->prog, h->d_emit.p, h->pending.cap, ncand, (OutValue*)c, (OutValue*)(c + o_aggs),
                                                      (OutPartial*)(c + o_parts), (uint64_t*)(c + o_rep), h->d_errp, h->stream, h->d_cand.p));
                else
                    HIP_TRY(h, launch_topk_compact(h->d_cand.p, ncand, nk, na, (const OutValue*)d, (const OutValue*)(d + off_aggs),
                                                   (const OutPartial*)(d + off_parts), (const uint64_t*)(d + off_rep), (OutValue*)c,
                                                   (OutValue*)(c + o_aggs), (OutPartial*)(c + o_parts), (uint64_t*)(c + o_rep), h->stream));
                src = c;
                h->stats.topk_candidates = ncand;
                ng = ncand;
            }
            uint32_t veto[2] = {0, 0};
            // results up to a few MB land in pinned memory (grown on demand); larger ones — millions of groups without a LIMIT —
            // in the pageable vector, whose copy the link's time dominates anyway
            const bool pinned_rows = copy_bytes <= (8u << 20);
            if (pinned_rows && h->pin_rows_cap < copy_bytes) {
                if (h->pin_rows) (void)hipHostFree(h->pin_rows);
                h->pin_rows = nullptr;
                h->pin_rows_cap = 0;
                const size_t want = std::max<size_t>(copy_bytes, 64u << 10);
                HIP_TRY(h, hipHostMalloc((void**)&h->pin_rows, want, hipHostMallocDefault));
                h->pin_rows_cap = want;
            }
            if (!pinned_rows) h->out_host.resize(copy_bytes);
            char* const landing = pinned_rows ? h->pin_rows : h->out_host.data();
            HIP_TRY(h, hipMemcpyAsync(landing, src, copy_bytes, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(h, hipMemcpyAsync(h->pin_counters + kCounters + 1, h->d_errp, 4, hipMemcpyDeviceToHost, h->stream));
            if (sets_deferred) HIP_TRY(h, hipMemcpyAsync(h->pin_counters + kCounters + 2, h->d_counters.p + 20, 8, hipMemcpyDeviceToHost, h->stream));
            mark_query_end(h);
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            memcpy(&err_flags, h->pin_counters + kCounters + 1, 4);
            if (sets_deferred) memcpy(veto, h->pin_counters + kCounters + 2, 8);
            if (sets_deferred && (veto[0] | veto[1])) {
                // a set or a bin overflowed on the optimistic path: no counts were added; once more, exactly
                sets_exact = true;
                ng = counters[1];
                goto redo_sets;
            }
            hp = landing;
        }
        h->r_keys.assign((const n1k_value*)hp, (const n1k_value*)hp + ng * nk);
        h->r_aggs.assign((const n1k_value*)(hp + o_aggs), (const n1k_value*)(hp + o_aggs) + ng * na);
        h->r_rep.assign((const uint64_t*)(hp + o_rep), (const uint64_t*)(hp + o_rep) + ng);
        const OutPartial* parts = (const OutPartial*)(hp + o_parts);
        h->r_parts.resize(ng * na);
        for (size_t i = 0; i < ng * na; i++) {
            n1k_partial& p = h->r_parts[i];
            memset(&p, 0, sizeof p);
            p.count = parts[i].count;
            p.isum = parts[i].isum;
            p.fsum = parts[i].fsum;
            p.int_exact = parts[i].flags & 1u;
            p.has_float = (parts[i].flags >> 1) & 1u;
            p.extreme.tag = (uint8_t)parts[i].ext_tag;
            p.extreme.v.code = parts[i].ext_payload;
            p.distinct = parts[i].distinct;
        }
    }
    if (ng > 0 && h->has_array_agg && !(err_flags & ERR_TABLE_FULL)) {
        n1k_status ast = array_agg_groups(h, ng, counters);
        if (ast != N1K_OK) return ast;
    }
    // What the verdict words of a multi-GPU exchange said comes first: every rank read the same verdicts, so every rank
    // returns from here with the same status (n1k_failure_is_global) and none of them goes on to the gather.
    if (err_flags & kErrFromVerdict) {
        h->failure_global = true;
        if (err_flags & ERR_PEER_FAILED) {
            const int ps = (int)(counters[27] & 0xFF);
            return fail(h, ps > 0 && ps <= N1K_REGION_FULL ? (n1k_status)ps : N1K_DEVICE_ERROR,
                        "a rank failed before the exchange (its status: %d); the step is void on every rank", ps);
        }
        if (err_flags & ERR_EXCHANGE_WIDE)
            return fail(h, N1K_UNSUPPORTED, "a sender's group keys hold float / wide integer values: use the row exchange");
        if (err_flags & ERR_EXCHANGE_OVERFLOW)
            return fail(h, N1K_REGION_FULL, "a sender's region overflowed: raise the region capacity (on every rank)");
        if (err_flags & ERR_PEER_UNPACKABLE)
            return fail(h, N1K_UNSUPPORTED_DATA, "a sender met a group key value that does not fit the packed key (option wide_values, or a key layout too narrow)");
        return fail(h, N1K_UNSUPPORTED_DATA, "a sender met a value outside the device subset (ordering of arrays/objects)");
    }
    if (err_flags & ERR_TABLE_FULL)
        return fail(h, N1K_OOM, "group table capacity exceeded: raise the max_groups option (now %llu)",
                    (unsigned long long)h->opt_max_groups);
    if (err_flags & ERR_UNPACKABLE_KEY)
        return fail(h, N1K_UNSUPPORTED_DATA,
                    "a group key value does not fit the packed key: more than %llu distinct float / wide integer key "
                    "values (option wide_values), or a key layout too narrow for them",
                    (unsigned long long)((1ull << h->prog.wide_bits) / 2));
    if (err_flags & ERR_UNSUPPORTED_VALUE)
        return fail(h, N1K_UNSUPPORTED_DATA, "a value outside the device subset was met (ordering of arrays/objects)");
    if (ng == 0 && nk == 0) {
        // FinalGroup.afterItems: no keys and no input -> one row of Default() values (execution/group_final.go:108-117)
        h->r_aggs.resize(na);
        h->r_parts.resize(na);
        h->r_rep.assign(1, ~0ull);
        for (uint32_t a = 0; a < na; a++) default_value(pl.aggs[a], h->r_aggs[a], h->r_parts[a]);
        ng = 1;
    }
    if (pl.has_having) {
        n1k_status st = having_groups(h, ng);
        if (st != N1K_OK) return st;
    }
    h->r_proj.clear();
    if (pl.has_project) {
        n1k_status st = project_groups(h, ng);
        if (st != N1K_OK) return st;
    }
    if (pl.has_order || pl.limit >= 0 || pl.offset > 0) {
        n1k_status st = order_groups(h, ng);
        if (st != N1K_OK) return st;
    }
    out->nproj = pl.has_project ? (uint32_t)h->project_ops.size() : 0;
    out->proj = out->nproj ? h->r_proj.data() : nullptr;
    out->ngroups = ng;
    out->keys = h->r_keys.data();
    out->aggs = h->r_aggs.data();
    out->partials = h->r_parts.data();
    out->rep_row = h->r_rep.data();
    h->stats.groups_out = ng;
    return N1K_OK;
    });
}

}  // extern "C"
