// n1k_engine.h — what the translation units of the host engine share: the handle, its helpers and the stages of a
// query (scan, partitioned GROUP BY, DISTINCT sets, grouped tail, multi-GPU exchange).  Internal: nothing here is part of
// the C ABI (include/n1k.h).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <numeric>
#include <string>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/n1k.h"
#include "n1k_jit.h"
#include "n1k_json.h"
#include "n1k_kernels.h"
#include "n1k_plan.h"


using namespace n1k;

static_assert(sizeof(n1k_value) == 16, "n1k_value layout");
static_assert(sizeof(OutValue) == sizeof(n1k_value), "OutValue must alias n1k_value");
static_assert(sizeof(Program) + sizeof(ScanArgs) + sizeof(GlobalTable) + 64 <= 4096, "kernel arguments exceed 4 KiB");

namespace n1k_eng {

extern thread_local std::string g_create_error;

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    hipError_t ensure(size_t count) {
        if (count <= n) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
        size_t want = std::max(count, (size_t)16);
        hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
        if (e == hipSuccess) n = want;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
};

inline uint64_t next_pow2(uint64_t x) {
    uint64_t p = 1;
    while (p < x) p <<= 1;
    return p;
}
inline uint32_t ceil_log2(uint64_t x) {
    uint32_t b = 0;
    while ((1ull << b) < x) b++;
    return b;
}

}  // namespace n1k_eng

using namespace n1k_eng;

struct n1k_handle {
    ParsedPlan plan;
    std::string last_error;
    std::atomic<int> stop_flag{0};

    // options
    int64_t opt_agg_mode = N1K_MODE_AUTO;
    uint64_t opt_max_groups = 1ull << 26;
    uint32_t opt_grid_blocks = 0;
    uint32_t opt_rep_row = 0;
    uint32_t opt_lds_bytes = 64 * 1024;   // HASH mode: LDS table bytes per workgroup
    uint32_t opt_block = 0, opt_rows_per_lane = 4, opt_fast = 1, opt_spec = 1, opt_wide = 1, opt_slabs = 1;
    uint32_t opt_jit = 1;                 // 0 off, 1 auto (large batches only), 2 always
    uint64_t opt_jit_min_rows = 4u << 20;
    uint32_t opt_topk_sample = 1;  // ORDER BY ... LIMIT: threshold of the device top-k filter from a sample first (0: always the exact radix select)
    uint32_t opt_lean_topk = 1;    // ORDER BY ... LIMIT over a kept region: order values first, rows for the candidates only
    uint32_t opt_part_block = 256; // workgroup size of the run-time-built partition kernel (256 | 512; measured 0.43 vs 0.58 ms per 100 M rows)
    uint32_t opt_part_subs = 1;    // row exchange: sub-regions per destination with their own counters (0: one dense run)
    uint32_t opt_part_per_cu = 0;  // workgroups per CU of the run-time-built partition kernel (0 = 2)
    uint32_t opt_pinned_out = 1;   // speculative FinalGroup writes its (few) groups straight into pinned host memory
    uint32_t opt_fuse_arith = 1;   // arithmetic nodes evaluated in registers by the run-time-built scan (no derived columns)
    bool derived_ready = true;     // the derived columns of the batch being pushed are materialised (or there are none)
    uint64_t opt_wide_values = 1u << 20;  // capacity of the wide key value tables (distinct big ints / floats)
    DevBuf<uint64_t> d_wide_int, d_wide_flt;
    // high-cardinality GROUP BY: record arrays (ping-pong per partition pass) and its tuning
    DevBuf<uint64_t> d_rec_key[3], d_rec_pay[3][kRecOperands];
    DevBuf<uint8_t> d_rec_tag[3][kRecOperands];
    DevBuf<uint64_t> d_emit;  // the bins' partial groups before they are merged into the table
    // the same path with the plan-specialised front end: 16-byte records (Rec16) written straight into 256 hash regions
    // by the scan (projection + first partition pass in one kernel), then into bins of fixed capacity
    DevBuf<uint64_t> d_rregion, d_rbins;
    DevBuf<unsigned long long> d_rcursor;
    uint32_t opt_records = 1;  // 0: always the three-array records of the interpreter front end (ablation, tests)
    uint32_t opt_rec_slots = 0, opt_rec_bins = 0, opt_rec_slices = 0, opt_rec_unroll = 0, opt_rec_block = 0, opt_rec_scan_per_cu = 0;  // tuning (0 = chosen from the data)
    // ... or instead of it: while the table is empty and their keys are unique, the region IS the set of groups;
    // n1k_finish finalizes it directly, anything else that needs the table merges it first (flush_pending)
    struct { uint64_t count = 0, cap = 0; } pending;
    // (measured, 100 M rows, GROUP BY cat, region_id: 6 400 groups 11.3 ms scan kernels vs 6.6 ms partitioned; 64 000 groups
    //  14.4 vs 9.8 ms: the LDS hash stage holds about a thousand groups, beyond that rows turn into global atomics)
    uint64_t opt_partition_min_rows = 8u << 20, opt_partition_probe_rows = 512u << 10, opt_partition_min_groups = 4096;
    int32_t opt_partition_levels = -1;
    // A handle that has just run a batch of about this size through the partitioned path (groups estimated from a probe of its
    // first rows) takes the next execution's batch the same way without probing again (a prepared statement executed again over
    // the same keyspace): the path checks itself (fixed-capacity regions and bins raise flags: exact path), and a batch that
    // would have been better off on the scan kernels is only slower, never wrong.  Forgotten when the path falls back.
    struct { bool valid = false; uint64_t rows = 0, groups_est = 0; } sticky;
    uint32_t opt_partition_sticky = 1;
    uint64_t groups_seen = 0;
    DevBuf<uint64_t> d_images;   // ORDER BY ... LIMIT: order images, candidate indices, select state, compacted records
    DevBuf<uint32_t> d_cand;
    DevBuf<char> d_topk, d_out2;
    uint64_t opt_topk_min_groups = 65536;  // device top-k filter from this many groups on
    // InitialProject over the final groups: an inner operator that only carries the derived columns of the terms'
    // expressions (its input columns are group keys / aggregates, like HAVING's)
    n1k_handle* project = nullptr;
    std::vector<int> project_cols;        // per inner column: key index k (>= 0) or -(aggregate index) - 1
    std::vector<Operand> project_ops;     // one per result term, in the inner operator's column space
    std::vector<n1k_value> r_proj;        // [ngroups][nterms]
    // HAVING: an inner Filter-only operator over the final groups (its columns are group keys / aggregates)
    n1k_handle* having = nullptr;
    std::vector<int> having_cols;        // per inner column: key index k (>= 0) or -(aggregate index) - 1
    std::vector<uint32_t> having_codes;  // dictionary code of this handle -> code of the inner handle (lazy)
    // raw documents -> columns (n1k_extract_json): leaf paths as field chains, the extracted batch
    std::vector<JsonPath> json_paths;
    int json_paths_state = 0;  // 0 not parsed, 1 ok, -1 some path is not a field chain
    std::vector<std::vector<uint8_t>> js_tags;
    std::vector<std::vector<uint64_t>> js_payload;
    std::vector<n1k_col> js_cols;
    uint32_t opt_json_threads = 0;  // 0 = hardware concurrency (at most 16)
    // n1k_push_json through the device extractor (n1k_jsondev.hip): the batch's bytes, offsets, status, columns, string table
    uint32_t opt_json_device = 1;
    uint64_t opt_json_device_min_docs = 4096;
    uint32_t opt_json_device_left_pct = 12;  // more documents than this left to the host: the host path takes the whole batch
    DevBuf<char> jd_bytes;
    DevBuf<uint64_t> jd_offsets, jd_new_first, jd_patch_docs, jd_patch_pay, jd_tab;
    DevBuf<uint8_t> jd_status, jd_patch_tags;
    DevBuf<uint32_t> jd_new_list, jd_code_of, jd_codes;
    std::vector<DevBuf<uint8_t>> jd_tags;
    std::vector<DevBuf<uint64_t>> jd_payload;
    std::vector<uint8_t> jd_host_status;
    bool failure_global = false;  // the last failure reported on this handle was learnt from (or told through) the verdict words of an
                                  // exchange: every rank's step fails alike, nobody enters the gather (n1k_failure_is_global)
    uint32_t opt_inject_failure = 0;  // tests: the exchange pretends that its site 1 (buffers) / 2 (partition, export) / 3 (receiving part) failed, once
    // One-call executions (n1k_run_device_batch): the query's last kernel (finalize_small_kernel) leaves table and counters as
    // n1k_reset would, so the next execution starts with its scan — device_clean says that the device state is what a reset
    // produces (any push / merge / partition clears it), clear_on_finish asks n1k_finish for that last kernel
    bool device_clean = false, clear_on_finish = false;
    uint32_t opt_filter_stream = 1; // Filter-only plans: the one-pass kernel (0: mask + scan + compaction, the ablation)
    uint32_t opt_fused_tail = 1;    // 0: finalize_kernel + publish_counters_kernel as separate launches (ablation)
    // the speculative FinalGroup of a small table: where its pieces land (n1k_finish.cpp small_tail_layout) ...
    struct SmallTailState { bool ok = false, fused = false, clear = false; uint64_t spec_groups = 0; size_t off_aggs = 0, off_parts = 0, off_rep = 0, total = 0; };
    SmallTailState tail_done;     // ... and, when the merge kernel's last workgroup has run it already (tail_in_merge), what it used
    bool tail_in_merge = false;
    bool one_call = false;        // inside n1k_run_device_batch: the batch is the whole query and its result leaves the device next
    // 1: the merge kernel's last workgroup runs the tail (scan -> merge+tail: two launches per query).  Measured, three
    // alternations on one box: 0.311-0.319 vs 0.305-0.309 ms per step at 100 M rows, 0.107-0.112 vs 0.103 at 10 M — the fences
    // and the wait for the last workgroup cost what the saved launch gains: off.
    uint32_t opt_tail_in_merge = 0;
    DevBuf<unsigned int> d_merge_done;
    uint32_t opt_agg_spec = 1;      // agg_bins16_kernel: the plan's one aggregate fixed at compile time (0: the generic kernel, A/B)
    uint32_t opt_merge_chunks = 0;  // merge_slabs_kernel: block rows (0 = from the grid)
    bool out_count_dirty = true;  // the finalize position counter holds a previous finish's count
    char* pin_out = nullptr;  // pinned host copy of a speculative FinalGroup (n1k_finish)
    unsigned long long* pin_counters = nullptr;  // pinned host copy of the device counters (one D2H per decision point): kCounters words,
                                                 // then kPinScratch words for the small reads of n1k_finish (candidate count, flags)
    char* pin_rows = nullptr;                    // pinned landing place of n1k_finish's sized output copy (pageable D2H copies are staged
    size_t pin_rows_cap = 0;                     //  by the runtime: ~ 35 us per copy + wait where the pinned one takes ~ 10)
    size_t pin_cap = 0;
    std::string jit_log;
    int device = -1;
    bool device_ready = false;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cus = 256;

    // dictionary (all STRING/ARRAY/OBJECT payloads are codes into it)
    std::vector<std::string> dict;
    std::unordered_map<std::string, uint32_t> dict_index;
    bool need_rank = false;
    size_t rank_built_for = (size_t)-1;
    DevBuf<uint32_t> d_rank;

    // compiled program (column pointers are patched per batch)
    Program prog{};
    bool layout_fixed = false;
    uint32_t col_kinds[kMaxCols]{};
    std::vector<std::string> agg_names;
    bool has_distinct = false, has_minmax = false, has_array_agg = false;
    uint32_t n_distinct = 0;
    // arithmetic operands -> derived columns (input columns first, then one per arithmetic node)
    struct Derived { uint32_t op, nops; Operand ops[4]; };
    std::vector<Derived> derived;
    std::vector<std::string> const_strings;  // string constants of the plan, interned lazily (see to_operand)
    std::vector<DevBuf<uint8_t>> dv_tags;
    std::vector<DevBuf<uint64_t>> dv_payload;
    DevBuf<uint64_t> d_log_key[kMaxDistinct], d_log_val[kMaxDistinct], d_regions, d_set_table;
    DevBuf<uint8_t> d_log_cls[kMaxDistinct];
    // COUNT(DISTINCT) member words (ScanArgs::log_word) and the scratch of their partition / de-duplication at finish
    DevBuf<uint64_t> d_log_word[kMaxDistinct], d_part[2], d_seg[3], d_wtable;
    DevBuf<unsigned long long> d_hist, d_cursor, d_dcounts, d_word_hist;
    // hash regions of the specialised scan's COUNT(DISTINCT) (WordLogArgs): per aggregate 256 regions x kRecSubs sub-regions
    // (kWordSubs in all) of wregion_cap words each
    DevBuf<uint64_t> d_wregion[kMaxDistinct], d_woff, d_wgather;
    DevBuf<unsigned long long> d_wcursor;  // kMaxDistinct x kWordSubs counters, kCursorStride apart
    uint64_t wregion_cap = 0;
    bool wregion_used = false;             // some batch of this query went through the regions
    uint32_t opt_dedupe_block = 1025;      // workgroup size of the de-duplication kernel, +1: probe word by word (tuning)
    uint32_t opt_spec_debug = 0;           // timing experiments: 1 words not stored, 2 word scatter skipped, 4 no workgroup cache, 8 finish skips the sets
    uint64_t opt_region_cap = 0;           // forced capacity of a hash region (tests: overflow into the plain log), 0 = from the rows
    uint32_t opt_distinct_words = 1;      // 0: every pair takes the (key, value, class) log and the global sets
    uint32_t opt_distinct_fill_pct = 25;     // a final bin's expected words, in % of the LDS set's slots (tuning)
    uint32_t opt_dedupe_unroll = 0;          // words per thread and chunk of the de-duplication kernel at 1024 threads: 2 (0) or 4 (tuning)
    uint32_t opt_distinct_set_slots = 8192;  // LDS set size of the de-duplication kernel (power of two; 64 KB: two workgroups per CU)
    int32_t opt_distinct_levels = -1;        // partition passes before the LDS sets: -1 = by log size, 0..2 forced (tests)
    uint32_t nw_key_bits = 0, nw_val_bits = 0;
    bool distinct_words[kMaxDistinct] = {false, false, false, false};
    uint32_t distinct_path = 0;  // how the last finish built the sets: bit 0 global pair sets, bit 1 LDS word sets, bit 2 global word set
    uint64_t log_capacity = 0;

    // device state
    GlobalTable table{};
    DevBuf<uint64_t> d_keys, d_acc, d_rep, d_slabs;
    DevBuf<unsigned long long> d_block_sel;
    uint32_t* d_errp = nullptr;  // lives inside d_counters ([12]) so one copy reads counters and flags
    DevBuf<unsigned long long> d_counters;  // [0] rows_selected [1] ngroups [2] out_count [3] filter total [4] rehash scratch
                                            // [5] distinct region words [8..11] pair-log cursors
    uint64_t row_base = 0;
    const unsigned long long* push_nrows_dev = nullptr;  // the batch being pushed holds min(nrows, *this) rows (n1k_exchange_rows)
    // the batch being pushed is segmented (a row region received from another GPU: kRowSubs sub-regions of push_seg_rows rows
    // capacity, their row counts on the device kCursorStride words apart)
    const unsigned long long* push_seg_counts = nullptr;
    uint32_t push_nseg = 0;
    uint64_t push_seg_rows = 0;
    uint64_t merged_groups_bound = 0;  // groups that may have arrived through merges (bounds the table like rows do)

    // staging for host batches
    // Two sets, used in turn: the H2D copies of batch k + 1 run on their own stream while the kernels of batch k still read
    // the other set; n1k_push_batch waits for its copies only (the caller's memory is free on return), never for kernels.
    std::vector<DevBuf<uint8_t>> st_tags[2];
    std::vector<DevBuf<uint64_t>> st_payload[2];
    std::vector<DevBuf<uint32_t>> st_codes[2];
    hipStream_t copy_stream = nullptr;
    hipEvent_t st_free[2] = {nullptr, nullptr};  // recorded on the compute stream behind the kernels that read the set
    bool st_busy[2] = {false, false};
    hipEvent_t st_copied = nullptr;
    int st_cur = 0;

    // filter-only path
    DevBuf<uint64_t> d_mask, d_tile_off, d_sel;
    DevBuf<uint32_t> d_tile_cnt;
    std::vector<uint64_t> selected;

    // results
    std::vector<n1k_value> r_keys, r_aggs;
    std::vector<n1k_partial> r_parts;
    std::vector<uint64_t> r_rep;
    DevBuf<char> d_out;            // finalize output: [keys][aggs][partials][rep rows], copied to the host at once
    std::vector<char> out_host;
    std::vector<char> export_blob;

    // host-side timing of the one-call path (N1K_HOST_TRACE=1: printed at destroy): [0] reset [1] push (launches) [2] finish up to
    // the wait [3] the wait [4] finish after the wait, in microseconds, and the number of calls
    double host_us[6] = {0, 0, 0, 0, 0, 0};
    // stats
    hipEvent_t ev_q0 = nullptr, ev_q1 = nullptr;  // the whole query on the stream: recorded by n1k_reset / before n1k_finish's last wait
    bool q0_recorded = false, q1_recorded = false;
    n1k_stats stats{};
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    std::vector<hipEvent_t> event_pool;
};

namespace n1k_eng {

n1k_status fail(n1k_handle* h, n1k_status st, const char* fmt, ...);

#define HIP_TRY(h, expr)                                                                                  \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess)                                                                             \
            return fail(h, _e == hipErrorOutOfMemory ? N1K_OOM : N1K_DEVICE_ERROR, "%s failed: %s", #expr, \
                        hipGetErrorString(_e));                                                           \
    } while (0)

constexpr uint32_t kPinScratch = 16;
constexpr uint64_t kWordSubs = 256ull * kRecSubs;  // sub-regions of a DISTINCT aggregate's member words

// n1k_engine.cpp: plan binding, device and table management
uint32_t intern(n1k_handle* h, const std::string& s);
uint32_t lookup_code(const n1k_handle* h, const char* s);
bool to_operand(n1k_handle* h, const Expr* e, Operand& o, PlanError& err);
bool compile_plan(n1k_handle* h, PlanError& err);
n1k_status ensure_device(n1k_handle* h);
n1k_status ensure_rank(n1k_handle* h);
n1k_status fix_layout(n1k_handle* h, const n1k_batch* b);
n1k_status ensure_table(n1k_handle* h, uint64_t incoming_rows);
n1k_status ensure_table_groups(n1k_handle* h, uint64_t groups);
hipEvent_t get_event(n1k_handle* h);
n1k_status ensure_pinned_counters(n1k_handle* h);
typedef n1k_handle::SmallTailState SmallTail;
bool small_tail_layout(n1k_handle* h, SmallTail& t);       // n1k_finish.cpp
n1k_status small_tail_pinned(n1k_handle* h, const SmallTail& t);
void drain_events(n1k_handle* h);
n1k_status validate_batch(n1k_handle* h, const n1k_batch* b);
uint64_t batch_bytes_per_row(const n1k_handle* h);
void default_value(const AggDef& d, n1k_value& v, n1k_partial& p);

// n1k_scan.cpp: one batch through Filter + InitialGroup (kernel choice), Filter-only batches, staging of host batches
bool build_fast_args(n1k_handle* h, uint32_t max_slots, FastArgs& F, bool fuse = false, bool partition_only = false);
SpecSig make_plan_sig(const n1k_handle* h, const FastArgs& F);
const SpecEntry* find_spec(const SpecSig& g);
n1k_status run_group_batch(n1k_handle* h, const n1k_batch* b);
n1k_status run_filter_batch(n1k_handle* h, const n1k_batch* b);
n1k_status bind_columns(n1k_handle* h, const n1k_batch* b, bool defer = false);
n1k_status materialize_derived(n1k_handle* h, const n1k_batch* b);
n1k_status push_device(n1k_handle* h, const n1k_batch* b);
n1k_status stage_host_batch(n1k_handle* h, const n1k_batch* batch, std::vector<n1k_col>& dcols);
n1k_status staged_batch_issued(n1k_handle* h);

// n1k_partitioned.cpp: GROUP BY with many groups (records -> partition passes -> per-bin LDS tables)
struct PartitionPlan {
    Operand src[kRecOperands];
    uint32_t nsrc = 0;
    uint32_t agg_src[kMaxAggs];
};
bool partition_eligible(n1k_handle* h, PartitionPlan& pp);
bool small_key_domain(const n1k_handle* h);
n1k_status flush_pending(n1k_handle* h);
n1k_status run_group_partitioned(n1k_handle* h, const n1k_batch* b, const PartitionPlan& pp, uint64_t groups_est, bool may_keep_region);
n1k_status run_group_records(n1k_handle* h, const n1k_batch* b, const PartitionPlan& pp, uint64_t groups_est, bool may_keep_region, bool* done);

// n1k_distinct.cpp: the sets of the DISTINCT aggregates at finish
n1k_status distinct_words_finish(n1k_handle* h, const AggSpec& ag, uint64_t nwords, bool hist_counted = true, const uint64_t* log = nullptr);
n1k_status distinct_regions_finish(n1k_handle* h, const AggSpec& ag, uint64_t nover, bool force_exact, bool* deferred);

// n1k_jsonpush.cpp: n1k_push_json through the device extractor (done = false: the host path takes the batch)
n1k_status push_json_device(n1k_handle* h, uint64_t ndocs, const uint64_t* offsets, const char* bytes, bool* done);

// n1k_tail.cpp: what follows FinalGroup (HAVING, projection, ORDER BY / OFFSET / LIMIT, ARRAY_AGG assembly)
n1k_status build_projection(n1k_handle* h);
n1k_status having_groups(n1k_handle* h, uint64_t& ng);
n1k_status project_groups(n1k_handle* h, uint64_t ng);
n1k_status order_groups(n1k_handle* h, uint64_t& ng);
n1k_status array_agg_groups(n1k_handle* h, uint64_t ng, const unsigned long long* counters);

// No C++ exception leaves the library ("no C++ exceptions or abort() across the ABI"; a Go caller cannot
// unwind through cgo): allocation failures of the host containers become N1K_OOM, anything else N1K_DEVICE_ERROR.
template <class F>
n1k_status guarded(const n1k_handle* ch, F&& f) noexcept {
    n1k_handle* h = const_cast<n1k_handle*>(ch);
    try {
        return f();
    } catch (const std::bad_alloc&) {
        try { if (h) h->last_error = "out of host memory"; else g_create_error = "out of host memory"; } catch (...) {}
        return N1K_OOM;
    } catch (const std::exception& e) {
        try { if (h) h->last_error = std::string("internal error: ") + e.what(); else g_create_error = e.what(); } catch (...) {}
        return N1K_DEVICE_ERROR;
    } catch (...) {
        return N1K_DEVICE_ERROR;
    }
}

}  // namespace n1k_eng
