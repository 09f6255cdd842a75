// n1k_engine.cpp — host engine behind the C ABI (include/n1k.h).
//
// Host-side mirror of the reference operators for this path: one Handle plays the role of one
// Parallel copy of Sequence[Filter, InitialGroup] plus the serial IntermediateGroup / FinalGroup
// (execution/parallel.go:52-83, filter.go, group_initial.go, group_intermediate.go, group_final.go),
// with the consumer life cycle of execution/base.go:492-545:
//     create = beforeItems, push_batch = processItem*, finish = afterItems, reset = reopen,
//     stop = SendStop, destroy = Done.
// There is no CPU fallback: without a HIP device every compute call fails with N1K_DEVICE_ERROR.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
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

#include <rccl/rccl.h>

using namespace n1k;

static_assert(sizeof(n1k_value) == 16, "n1k_value layout");
static_assert(sizeof(OutValue) == sizeof(n1k_value), "OutValue must alias n1k_value");
static_assert(sizeof(Program) + sizeof(ScanArgs) + sizeof(GlobalTable) + 64 <= 4096, "kernel arguments exceed 4 KiB");

namespace {

thread_local std::string g_create_error;

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

uint64_t next_pow2(uint64_t x) {
    uint64_t p = 1;
    while (p < x) p <<= 1;
    return p;
}
uint32_t ceil_log2(uint64_t x) {
    uint32_t b = 0;
    while ((1ull << b) < x) b++;
    return b;
}

}  // namespace

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
    bool out_count_dirty = true;  // the finalize position counter holds a previous finish's count
    char* pin_out = nullptr;  // pinned host copy of a speculative FinalGroup (n1k_finish)
    unsigned long long* pin_counters = nullptr;  // pinned host copy of the device counters (one D2H per decision point)
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

    // stats
    n1k_stats stats{};
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    std::vector<hipEvent_t> event_pool;
};

namespace {

n1k_status fail(n1k_handle* h, n1k_status st, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    h->last_error = buf;
    return st;
}

#define HIP_TRY(h, expr)                                                                                  \
    do {                                                                                                  \
        hipError_t _e = (expr);                                                                           \
        if (_e != hipSuccess)                                                                             \
            return fail(h, _e == hipErrorOutOfMemory ? N1K_OOM : N1K_DEVICE_ERROR, "%s failed: %s", #expr, \
                        hipGetErrorString(_e));                                                           \
    } while (0)

n1k_status materialize_derived(n1k_handle* h, const n1k_batch* b);

uint32_t intern(n1k_handle* h, const std::string& s) {
    auto it = h->dict_index.find(s);
    if (it != h->dict_index.end()) return it->second;
    uint32_t code = (uint32_t)h->dict.size();
    h->dict.push_back(s);
    h->dict_index.emplace(s, code);
    return code;
}

uint32_t lookup_code(const n1k_handle* h, const char* s) {
    auto it = h->dict_index.find(s);
    return it == h->dict_index.end() ? 0xFFFFFFFFu : it->second;
}

bool to_operand(n1k_handle* h, const Expr* e, Operand& o, PlanError& err) {
    memset(&o, 0, sizeof o);
    if (e->kind == EK::Path) {
        for (size_t i = 0; i < h->plan.paths.size(); i++)
            if (h->plan.paths[i] == e->text) {
                o.is_const = 0;
                o.col = (uint32_t)i;
                return true;
            }
        err.msg = "unbound path " + e->text;
        return false;
    }
    if (e->kind == EK::Const) {
        o.is_const = 1;
        o.ctag = e->ctag;
        o.cpayload = e->cpayload;
        if (e->ctag == T_STRING) {
            // string constants are interned at the first push, AFTER whatever dictionary the caller interned, so that
            // a caller who interns its dictionary right after n1k_create keeps its own codes (code == index)
            o.pad = 1;
            o.cpayload = h->const_strings.size();
            h->const_strings.push_back(e->cstr);
        }
        return true;
    }
    // arithmetic node -> derived column evaluated once per batch (expression/arith_*.go, func_num.go)
    uint32_t op;
    if (e->kind == EK::Func) {
        const std::string& f = e->fname;
        op = f == "round" ? AR_ROUND : f == "trunc" ? AR_TRUNC : f == "abs" ? AR_ABS : f == "ceil" ? AR_CEIL
             : f == "floor" ? AR_FLOOR : f == "sign" ? AR_SIGN : AR_SQRT;
        n1k_handle::Derived d{};
        d.op = op;
        for (auto& c : e->ch) {
            Operand x;
            if (!to_operand(h, c.get(), x, err)) return false;
            d.ops[d.nops++] = x;
        }
        if (h->plan.paths.size() + h->derived.size() >= (size_t)kMaxCols) {
            err.unsupported = true;
            err.msg = "too many columns (inputs + arithmetic nodes > 16)";
            return false;
        }
        memset(&o, 0, sizeof o);
        o.is_const = 0;
        o.col = (uint32_t)(h->plan.paths.size() + h->derived.size());
        h->derived.push_back(d);
        return true;
    }
    switch (e->kind) {
        case EK::Add: op = AR_ADD; break;
        case EK::Mult: op = AR_MULT; break;
        case EK::Sub: op = AR_SUB; break;
        case EK::Div: op = AR_DIV; break;
        case EK::Mod: op = AR_MOD; break;
        case EK::Neg: op = AR_NEG; break;
        case EK::IDiv: op = AR_IDIV; break;
        case EK::IMod: op = AR_IMOD; break;
        default:
            err.unsupported = true;
            err.msg = "a predicate used as a value is not on the device path";
            return false;
    }
    auto emit = [&](const n1k_handle::Derived& d, Operand& out) -> bool {
        if (h->plan.paths.size() + h->derived.size() >= (size_t)kMaxCols) {
            err.unsupported = true;
            err.msg = "too many columns (inputs + arithmetic nodes > 16)";
            return false;
        }
        memset(&out, 0, sizeof out);
        out.is_const = 0;
        out.col = (uint32_t)(h->plan.paths.size() + h->derived.size());
        h->derived.push_back(d);
        return true;
    };
    std::vector<Operand> ops;
    for (auto& c : e->ch) {
        Operand x;
        if (!to_operand(h, c.get(), x, err)) return false;
        // (a string operand needs no dictionary rank here: arithmetic over a non-number is NULL, expression/arith_add.go:51-70)
        ops.push_back(x);
    }
    // n-ary Add / Mult fold left to right, at most 4 operands per kernel: ((a+b+c+d) + e + ...)
    size_t i = 0;
    Operand acc{};
    bool have_acc = false;
    do {
        n1k_handle::Derived d{};
        d.op = op;
        if (have_acc) d.ops[d.nops++] = acc;
        while (i < ops.size() && d.nops < 4) d.ops[d.nops++] = ops[i++];
        if (!emit(d, acc)) return false;
        have_acc = true;
    } while (i < ops.size());
    o = acc;
    return true;
}

// condition tree -> postfix over predicate terms
bool compile_cond(n1k_handle* h, const Expr* e, PlanError& err) {
    Program& P = h->prog;
    auto push_term = [&](uint32_t op, const Expr* a, const Expr* b, const Expr* c) -> bool {
        if (P.nterms >= (uint32_t)kMaxTerms || P.nlogic >= (uint32_t)kMaxLogic) {
            err.unsupported = true;
            err.msg = "condition too large for the device program";
            return false;
        }
        Term& t = P.terms[P.nterms];
        memset(&t, 0, sizeof t);
        t.op = op;
        if (a && !to_operand(h, a, t.a, err)) return false;
        if (b && !to_operand(h, b, t.b, err)) return false;
        if (c && !to_operand(h, c, t.c, err)) return false;
        P.logic[P.nlogic++] = LogicOp{LOGIC_PUSH, (uint8_t)P.nterms};
        P.nterms++;
        return true;
    };
    auto push_logic = [&](uint8_t op, uint8_t arg) -> bool {
        if (P.nlogic >= (uint32_t)kMaxLogic) {
            err.unsupported = true;
            err.msg = "condition too large for the device program";
            return false;
        }
        P.logic[P.nlogic++] = LogicOp{op, arg};
        return true;
    };
    switch (e->kind) {
        case EK::And:
        case EK::Or:
            if (e->ch.size() > 16) { err.unsupported = true; err.msg = "AND/OR arity > 16"; return false; }
            for (auto& c : e->ch)
                if (!compile_cond(h, c.get(), err)) return false;
            return push_logic(e->kind == EK::And ? LOGIC_AND : LOGIC_OR, (uint8_t)e->ch.size());
        case EK::Not:
            if (!compile_cond(h, e->ch[0].get(), err)) return false;
            return push_logic(LOGIC_NOT, 0);
        case EK::Eq:
        case EK::LT:
        case EK::LE: {
            const Expr* a = e->ch[0].get();
            const Expr* b = e->ch[1].get();
            auto is_num = [](const Expr* x) { return x->kind == EK::Const && (x->ctag == T_INT || x->ctag == T_FLOAT); };
            // x <op> NUMBER constant (either side): the cheap term form, same semantics
            if (is_num(b) && !is_num(a)) {
                uint32_t op = e->kind == EK::Eq ? TERM_NUM_EQ : (e->kind == EK::LT ? TERM_NUM_LT : TERM_NUM_LE);
                return push_term(op, a, b, nullptr);
            }
            if (is_num(a) && !is_num(b)) {  // (c < x) == (x > c)
                uint32_t op = e->kind == EK::Eq ? TERM_NUM_EQ : (e->kind == EK::LT ? TERM_NUM_GT : TERM_NUM_GE);
                return push_term(op, b, a, nullptr);
            }
            h->need_rank = true;
            return push_term(e->kind == EK::Eq ? TERM_EQ : (e->kind == EK::LT ? TERM_LT : TERM_LE), a, b, nullptr);
        }
        case EK::Between:
            h->need_rank = true;
            return push_term(TERM_BETWEEN, e->ch[0].get(), e->ch[1].get(), e->ch[2].get());
        case EK::IsNull: return push_term(TERM_IS_NULL, e->ch[0].get(), nullptr, nullptr);
        case EK::IsNotNull: return push_term(TERM_IS_NOT_NULL, e->ch[0].get(), nullptr, nullptr);
        case EK::IsMissing: return push_term(TERM_IS_MISSING, e->ch[0].get(), nullptr, nullptr);
        case EK::IsNotMissing: return push_term(TERM_IS_NOT_MISSING, e->ch[0].get(), nullptr, nullptr);
        case EK::IsValued: return push_term(TERM_IS_VALUED, e->ch[0].get(), nullptr, nullptr);
        case EK::IsNotValued: return push_term(TERM_IS_NOT_VALUED, e->ch[0].get(), nullptr, nullptr);
        default: return push_term(TERM_TRUTH, e, nullptr, nullptr);  // a value used as a condition (incl. arithmetic)
    }
}

bool compile_plan(n1k_handle* h, PlanError& err) {
    Program& P = h->prog;
    memset(&P, 0, sizeof P);
    h->derived.clear();
    h->const_strings.clear();
    const ParsedPlan& pl = h->plan;
    if (pl.paths.size() > (size_t)kMaxCols) { err.unsupported = true; err.msg = "more than 16 leaf paths"; return false; }
    if (pl.keys.size() > (size_t)kMaxKeys) { err.unsupported = true; err.msg = "more than 4 group keys"; return false; }
    if (pl.aggs.size() > (size_t)kMaxAggs) { err.unsupported = true; err.msg = "more than 8 aggregates"; return false; }
    P.ncols = (uint32_t)pl.paths.size();
    if (pl.condition && !compile_cond(h, pl.condition.get(), err)) return false;
    P.nkeys = (uint32_t)pl.keys.size();
    for (uint32_t k = 0; k < P.nkeys; k++)
        if (!to_operand(h, pl.keys[k].get(), P.keys[k].src, err)) return false;
    P.naggs = (uint32_t)pl.aggs.size();
    uint32_t lds_w = 1, glob_w = 0;
    if (h->opt_rep_row) {
        P.want_rep_row = 1;
        P.rep_lds_word = lds_w++;
    }
    for (uint32_t a = 0; a < P.naggs; a++) {
        const AggDef& d = pl.aggs[a];
        AggSpec& s = P.aggs[a];
        memset(&s, 0, sizeof s);
        s.kind = d.kind;
        s.distinct = d.distinct ? 1 : 0;
        s.has_operand = d.operand ? 1 : 0;
        if (d.operand && !to_operand(h, d.operand.get(), s.src, err)) return false;
        h->agg_names.push_back(d.text);
        s.lds_off = lds_w;
        s.glob_off = glob_w;
        if (d.kind == AGG_ARRAY && !d.operand) { err.msg = "array_agg needs an operand"; return false; }
        if (d.kind == AGG_ARRAY) h->has_array_agg = true;
        if (d.distinct || d.kind == AGG_ARRAY) {  // (ARRAY_AGG logs its operands the way the DISTINCT aggregates do)
            s.distinct = 1;
            if (h->n_distinct >= kMaxDistinct) {
                err.unsupported = true;
                err.msg = "more than 4 DISTINCT aggregates";
                return false;
            }
            h->has_distinct = true;
            s.log_index = h->n_distinct++;
            s.lds_n = d.kind == AGG_ARRAY ? 0 : kLdsWordsDistinct;
            lds_w += s.lds_n;
            glob_w += (d.kind == AGG_SUM || d.kind == AGG_AVG) ? kGlobWordsDistinctSum : kGlobWordsDistinct;
        } else if (d.kind == AGG_COUNT || d.kind == AGG_COUNTN) {
            lds_w += 1;
            glob_w += 1;
        } else if (d.kind == AGG_SUM) {
            lds_w += kLdsWordsSum;
            glob_w += kGlobWordsSum;
        } else if (d.kind == AGG_AVG) {
            lds_w += kLdsWordsAvg;
            glob_w += kGlobWordsAvg;
        } else {
            h->need_rank = true;
            h->has_minmax = true;
            lds_w += kWordsMinMax;
            glob_w += kWordsMinMax;
        }
        if (pl.has_order) h->need_rank = true;  // order images of string values (top-k filter)
    }
    P.lds_words = lds_w;
    P.glob_words = glob_w ? glob_w : 1;
    if (h->has_array_agg) {
        if (h->opt_rep_row) { err.unsupported = true; err.msg = "array_agg with representative rows"; return false; }
        P.emit_packed_key = 1;
    }
    P.ncols = (uint32_t)(pl.paths.size() + h->derived.size());  // inputs, then derived columns
    return true;
}

n1k_status ensure_device(n1k_handle* h) {
    if (h->device_ready) {
        HIP_TRY(h, hipSetDevice(h->device));
        return N1K_OK;
    }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(h, N1K_DEVICE_ERROR, "no HIP device available (this library has no CPU fallback)");
    if (h->device < 0) {
        int cur = 0;
        HIP_TRY(h, hipGetDevice(&cur));
        h->device = cur;
    }
    if (h->device >= n) return fail(h, N1K_DEVICE_ERROR, "device %d out of range (%d visible)", h->device, n);
    HIP_TRY(h, hipSetDevice(h->device));
    hipDeviceProp_t prop;
    HIP_TRY(h, hipGetDeviceProperties(&prop, h->device));
    h->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (!h->stream) {
        HIP_TRY(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        h->own_stream = true;
    }
    HIP_TRY(h, h->d_counters.ensure(kCounters));
    h->d_errp = (uint32_t*)(h->d_counters.p + 12);
    HIP_TRY(h, hipMemsetAsync(h->d_counters.p, 0, kCounters * sizeof(unsigned long long), h->stream));
    h->device_ready = true;
    return N1K_OK;
}

// bytewise rank of every dictionary string (value/string.go:116-130 compares Go strings bytewise)
n1k_status ensure_rank(n1k_handle* h) {
    if (!h->need_rank || h->rank_built_for == h->dict.size()) {
        h->prog.str_rank = h->d_rank.p;
        return N1K_OK;
    }
    size_t n = h->dict.size();
    std::vector<uint32_t> order(n), rank(std::max(n, (size_t)1));
    std::iota(order.begin(), order.end(), 0u);
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return h->dict[a] < h->dict[b]; });
    for (size_t i = 0; i < n; i++) rank[order[i]] = (uint32_t)i;
    HIP_TRY(h, hipStreamSynchronize(h->stream));  // earlier launches may still read the old table
    HIP_TRY(h, h->d_rank.ensure(std::max(n, (size_t)1)));
    if (n) HIP_TRY(h, hipMemcpy(h->d_rank.p, rank.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
    const bool rebuilt = h->rank_built_for != (size_t)-1;
    h->rank_built_for = n;
    h->prog.str_rank = h->d_rank.p;
    // groups that hold string MIN / MAX winners carry ranks of the old order: re-stamp them (n1k_kernels.hip)
    if (rebuilt && h->has_minmax && h->layout_fixed && h->table.capacity && (h->row_base || h->merged_groups_bound))
        HIP_TRY(h, launch_restamp_ranks(h->prog, h->table, h->stream));
    return N1K_OK;
}

// Decide the key bit fields once the column kinds are known (first batch).
n1k_status fix_layout(n1k_handle* h, const n1k_batch* b) {
    Program& P = h->prog;
    for (uint32_t c = 0; c < (uint32_t)h->plan.paths.size(); c++) h->col_kinds[c] = b->cols[c].kind;
    for (uint32_t c = (uint32_t)h->plan.paths.size(); c < P.ncols; c++) h->col_kinds[c] = N1K_COL_TAGGED64;  // derived columns
    uint32_t n_dict = 0, n_tag = 0;
    for (uint32_t k = 0; k < P.nkeys; k++) {
        KeySpec& ks = P.keys[k];
        bool dict = !ks.src.is_const && h->col_kinds[ks.src.col] == N1K_COL_DICT32;
        ks.mode = dict ? KEYM_DICT : KEYM_TAGGED;
        if (dict) n_dict++; else n_tag++;
    }
    uint32_t dbits = 0, tbits = 0;
    if (n_tag == 0 && n_dict) dbits = std::min(32u, 63u / n_dict);
    else if (n_tag) {
        dbits = n_dict ? 24u : 0u;
        if (n_dict * dbits + n_tag * 8 > 63) dbits = (63 - n_tag * 8) / std::max(n_dict, 1u);
        tbits = (63 - n_dict * dbits) / n_tag;
    }
    uint32_t shift = 0;
    for (uint32_t k = 0; k < P.nkeys; k++) {
        KeySpec& ks = P.keys[k];
        ks.bits = ks.mode == KEYM_DICT ? dbits : tbits;
        ks.shift = shift;
        shift += ks.bits;
        if (ks.bits < (ks.mode == KEYM_DICT ? 4u : 8u)) return fail(h, N1K_UNSUPPORTED, "group key layout does not fit 63 bits");
    }
    // value tables for the numbers a TAGGED field cannot hold itself (n1k_device.h: wide_code)
    P.wide_int = P.wide_flt = nullptr;
    P.wide_bits = 0;
    P.wide_count = h->d_counters.p + 13;
    if (n_tag && h->opt_wide_values) {
        uint32_t wb = 4;
        while ((1ull << wb) < h->opt_wide_values * 2 && wb < 30) wb++;
        wb = std::min(wb, tbits - 4);
        const size_t n = (size_t)1 << wb;
        HIP_TRY(h, h->d_wide_int.ensure(n));
        HIP_TRY(h, h->d_wide_flt.ensure(n));
        HIP_TRY(h, hipMemsetAsync(h->d_wide_int.p, 0xFF, n * 8, h->stream));
        HIP_TRY(h, hipMemsetAsync(h->d_wide_flt.p, 0xFF, n * 8, h->stream));
        P.wide_int = h->d_wide_int.p;
        P.wide_flt = h->d_wide_flt.p;
        P.wide_bits = wb;
    }
    // COUNT(DISTINCT) member words: [packed key : K1][class : 3][value : 61 - K1].  A layout of few bits keeps all
    // of them; wider layouts (integer keys, several keys) only qualify row by row when the packed key is small.
    {
        uint32_t total = 0;
        for (uint32_t k = 0; k < P.nkeys; k++) total += P.keys[k].bits;
        h->nw_key_bits = P.nkeys == 0 ? 1u : (total <= 34 ? total : 24u);
        h->nw_val_bits = 61 - h->nw_key_bits;
        for (uint32_t a = 0; a < P.naggs; a++)
            if (P.aggs[a].distinct)
                h->distinct_words[P.aggs[a].log_index] = h->opt_distinct_words && P.aggs[a].kind == AGG_COUNT;
    }
    h->layout_fixed = true;
    return N1K_OK;
}

}  // namespace

// ---- table management -------------------------------------------------------------------------

namespace {

constexpr uint64_t kWordSubs = 256ull * kRecSubs;  // sub-regions of a DISTINCT aggregate's member words

n1k_status alloc_table(n1k_handle* h, uint64_t capacity, GlobalTable& t, DevBuf<uint64_t>& keys, DevBuf<uint64_t>& acc,
                       DevBuf<uint64_t>& rep) {
    HIP_TRY(h, keys.ensure(capacity));
    HIP_TRY(h, acc.ensure(capacity * h->prog.glob_words));
    if (h->prog.want_rep_row) HIP_TRY(h, rep.ensure(capacity));
    t.keys = keys.p;
    t.acc = acc.p;
    t.rep_row = h->prog.want_rep_row ? rep.p : nullptr;
    t.capacity = capacity;
    HIP_TRY(h, launch_init_table(h->prog, t, 0, capacity, nullptr, h->stream));
    return N1K_OK;
}

// Make sure the global table can take `incoming_rows` more rows worth of new groups (bounded by max_groups).
n1k_status ensure_table(n1k_handle* h, uint64_t incoming_rows) {
    // groups <= rows pushed so far: an upper bound that needs no device round trip
    uint64_t want_groups = std::min<uint64_t>(h->opt_max_groups, h->row_base + h->merged_groups_bound + incoming_rows);
    if (h->prog.nkeys == 0) want_groups = 1;
    else {
        // all keys dictionary coded: the key domain bounds the number of groups (|dict| + MISSING + NULL per key)
        bool all_dict = true;
        long double dom = 1;
        for (uint32_t k = 0; k < h->prog.nkeys; k++) {
            all_dict &= h->prog.keys[k].mode == KEYM_DICT;
            dom *= (long double)h->dict.size() + 2;
        }
        if (all_dict && dom < (long double)want_groups) want_groups = (uint64_t)dom;
    }
    uint64_t cap = next_pow2(std::max<uint64_t>(want_groups * 2, 1024));
    if (cap <= h->table.capacity) return N1K_OK;
    if (!h->table.capacity) return alloc_table(h, cap, h->table, h->d_keys, h->d_acc, h->d_rep);
    // grow: rehash every occupied slot into a bigger table (keys keep their packed form)
    GlobalTable nt{};
    DevBuf<uint64_t> nk, na, nr;
    n1k_status st = alloc_table(h, cap, nt, nk, na, nr);
    if (st != N1K_OK) return st;
    HIP_TRY(h, launch_rehash(h->prog, h->table, nt, h->d_errp, h->d_counters.p + 4, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->d_keys.release();
    h->d_acc.release();
    h->d_rep.release();
    h->d_keys = nk;
    h->d_acc = na;
    h->d_rep = nr;
    h->table = nt;
    return N1K_OK;
}

// the same for a known number of groups (the partitioned path counts its groups before it inserts them)
n1k_status ensure_table_groups(n1k_handle* h, uint64_t groups) {
    if (groups > h->opt_max_groups)
        return fail(h, N1K_OOM, "group table capacity exceeded: raise the max_groups option (now %llu)", (unsigned long long)h->opt_max_groups);
    uint64_t cap = next_pow2(std::max<uint64_t>(groups * 2, 1024));
    if (cap <= h->table.capacity) return N1K_OK;
    if (!h->table.capacity) return alloc_table(h, cap, h->table, h->d_keys, h->d_acc, h->d_rep);
    GlobalTable nt{};
    DevBuf<uint64_t> nk, na, nr;
    n1k_status st = alloc_table(h, cap, nt, nk, na, nr);
    if (st != N1K_OK) return st;
    HIP_TRY(h, launch_rehash(h->prog, h->table, nt, h->d_errp, h->d_counters.p + 4, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->d_keys.release();
    h->d_acc.release();
    h->d_rep.release();
    h->d_keys = nk;
    h->d_acc = na;
    h->d_rep = nr;
    h->table = nt;
    return N1K_OK;
}

hipEvent_t get_event(n1k_handle* h) {
    if (!h->event_pool.empty()) {
        hipEvent_t e = h->event_pool.back();
        h->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

void drain_events(n1k_handle* h) {
    for (auto& pr : h->events) {
        float ms = 0.f;
        if (pr.first && pr.second && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) h->stats.device_ms += ms;
        if (pr.first) h->event_pool.push_back(pr.first);
        if (pr.second) h->event_pool.push_back(pr.second);
    }
    h->events.clear();
}

n1k_status validate_batch(n1k_handle* h, const n1k_batch* b) {
    if (!b) return fail(h, N1K_INVALID, "null batch");
    if (b->ncols != (uint32_t)h->plan.paths.size()) return fail(h, N1K_INVALID, "batch has %u columns, plan needs %u", b->ncols, (uint32_t)h->plan.paths.size());
    for (uint32_t c = 0; c < b->ncols; c++) {
        const n1k_col& col = b->cols[c];
        if (col.kind == N1K_COL_DICT32) {
            if (b->nrows && !col.codes) return fail(h, N1K_INVALID, "column %u: null codes", c);
        } else if (col.kind == N1K_COL_TAGGED64) {
            if (b->nrows && (!col.tags || !col.payload)) return fail(h, N1K_INVALID, "column %u: null tags/payload", c);
        } else
            return fail(h, N1K_INVALID, "column %u: unknown kind %u", c, col.kind);
        if (h->layout_fixed && col.kind != h->col_kinds[c])
            return fail(h, N1K_INVALID, "column %u changed kind between batches", c);
    }
    return N1K_OK;
}

uint64_t batch_bytes_per_row(const n1k_handle* h) {
    uint64_t b = 0;
    for (uint32_t c = 0; c < (uint32_t)h->plan.paths.size(); c++) b += h->col_kinds[c] == N1K_COL_DICT32 ? 4 : 9;
    return b;
}

// Can this plan run on the fast kernel (bounded shape, every descriptor static)?  Fills F when it can.
// fuse: the plan's arithmetic nodes stay out of HBM — the kernel (a run-time-built plan-specialised one) evaluates them in
// registers from the input columns; otherwise they are materialised derived columns and count as inputs.
// partition_only: only columns, terms and keys matter (the row exchange's partition kernels: no aggregate runs there).
bool build_fast_args(n1k_handle* h, uint32_t max_slots, FastArgs& F, bool fuse = false, bool partition_only = false) {
    const Program& P = h->prog;
    memset(&F, 0, sizeof F);
    const uint32_t ni = (uint32_t)h->plan.paths.size(), nd = (uint32_t)h->derived.size();
    if (fuse) {
        if (nd == 0 || nd > (uint32_t)kFastDerived || ni == 0 || ni > (uint32_t)kFastCols) return false;
        for (uint32_t d = 0; d < nd; d++)
            for (uint32_t k = 0; k < h->derived[d].nops; k++) {
                const Operand& o = h->derived[d].ops[k];
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
                if (F.slabs) HIP_TRY(h, launch_merge_slabs(P, F, h->table, g, h->d_counters.p + 1, h->stream));
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
            if (F.slabs) HIP_TRY(h, launch_merge_slabs(P, F, h->table, g, h->d_counters.p + 1, h->stream));
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
    uint64_t ntiles = (b->nrows + kFilterTile - 1) / kFilterTile;
    if (ntiles == 0) return N1K_OK;
    HIP_TRY(h, h->d_mask.ensure(ntiles * (kFilterTile / 64)));
    HIP_TRY(h, h->d_tile_cnt.ensure(ntiles));
    HIP_TRY(h, h->d_tile_off.ensure(ntiles));
    uint64_t nchunks = (b->nrows + 1023) / 1024;
    uint32_t grid = (uint32_t)std::min<uint64_t>(nchunks, (uint64_t)h->num_cus * 8);
    hipEvent_t e0 = get_event(h), e1 = get_event(h);
    if (e0) (void)hipEventRecord(e0, h->stream);
    HIP_TRY(h, hipMemsetAsync(h->d_tile_cnt.p, 0, ntiles * sizeof(uint32_t), h->stream));
    HIP_TRY(h, launch_filter_mask(P, b->nrows, h->d_mask.p, h->d_tile_cnt.p, h->d_errp, grid, h->stream));
    HIP_TRY(h, launch_tile_scan(h->d_tile_cnt.p, h->d_tile_off.p, ntiles, h->d_counters.p + 3, h->stream));
    unsigned long long total = 0;
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
n1k_status bind_columns(n1k_handle* h, const n1k_batch* b, bool defer = false) {
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

// ---- high-cardinality GROUP BY: records -> radix partition -> per-bin LDS aggregation (n1k_kernels.hip) ---------

struct PartitionPlan {
    Operand src[kRecOperands];
    uint32_t nsrc = 0;
    uint32_t agg_src[kMaxAggs];
};

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
        const uint32_t wpr = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(h->opt_rec_slices ? h->opt_rec_slices : 8u, region_tiles));
        HIP_TRY(h, launch_radix_scatter16(R, wpr, bps, h->stream));
        B.rec = h->d_rbins.p;
        B.bin_count = h->d_cursor.p;
        B.bin_count_stride = 1;
        B.bin_stride = bin_cap;
        B.nbins = (uint32_t)nbins;
    }
    // (3) one workgroup per bin: InitialGroup in an LDS table, the bin's groups into the compact region
    B.lds_slots = slots;
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
        const size_t shmem = (size_t)slots * P.lds_words * 8 + 1024;
        const uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(2048 / block / 2, (160 * 1024) / shmem));
        const uint32_t bgrid = (uint32_t)std::min<uint64_t>(B.nbins, (uint64_t)h->num_cus * per_cu);
        const uint64_t per_thread = mean / block + 1;
        HIP_TRY(h, launch_agg_bins16(P, B, bgrid, block, h->opt_rec_unroll ? h->opt_rec_unroll : (uint32_t)std::min<uint64_t>(per_thread, 4), h->stream));
    }
    // one copy of the counters into pinned memory (the region's group count joins them first): one host round trip
    if (!h->pin_counters) HIP_TRY(h, hipHostMalloc((void**)&h->pin_counters, kCounters * sizeof(unsigned long long), hipHostMallocDefault));
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

n1k_status push_device(n1k_handle* h, const n1k_batch* b) {
    if (h->stop_flag.load()) return fail(h, N1K_STOPPED, "operator was stopped");
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

void default_value(const AggDef& d, n1k_value& v, n1k_partial& p) {
    memset(&v, 0, sizeof v);
    memset(&p, 0, sizeof p);
    p.extreme.tag = N1K_T_NULL;
    // Default(): COUNT/COUNTN (also DISTINCT) 0, everything else NULL
    if (d.kind == AGG_COUNT || d.kind == AGG_COUNTN) {
        v.tag = N1K_T_INT;
        v.v.i = 0;
    } else
        v.tag = N1K_T_NULL;
}

}  // namespace

// Every aggregate and key text of the plan inside `expr` (longest first) becomes a synthetic leaf path (`$g`.`aN`) /
// (`$g`.`kN`): an expression over the final groups then reads columns, like any other (HAVING, projection terms).
static std::string group_paths(const n1k_handle* h, std::string expr) {
    std::vector<std::pair<std::string, std::string>> subst;
    for (size_t a = 0; a < h->plan.aggs.size(); a++) subst.emplace_back(h->plan.aggs[a].text, "(`$g`.`a" + std::to_string(a) + "`)");
    for (size_t k = 0; k < h->plan.key_texts.size(); k++) subst.emplace_back(h->plan.key_texts[k], "(`$g`.`k" + std::to_string(k) + "`)");
    std::stable_sort(subst.begin(), subst.end(), [](const auto& x, const auto& y) { return x.first.size() > y.first.size(); });
    for (auto& sb : subst) {
        if (sb.first.empty()) continue;
        for (size_t pos = 0; (pos = expr.find(sb.first, pos)) != std::string::npos; pos += sb.second.size())
            expr.replace(pos, sb.first.size(), sb.second);
    }
    return expr;
}

// (`$g`.`kN`) / (`$g`.`aN`) -> N (keys) or -N - 1 (aggregates); false for any other path
static bool group_path_index(const n1k_handle* h, const std::string& p, int& out) {
    int idx = -1;
    char kind = 0;
    if (sscanf(p.c_str(), "(`$g`.`%c%d`)", &kind, &idx) != 2 || (kind != 'k' && kind != 'a') || idx < 0 ||
        (size_t)idx >= (kind == 'k' ? h->plan.key_texts.size() : h->plan.aggs.size()))
        return false;
    out = kind == 'k' ? idx : -idx - 1;
    return true;
}

// InitialProject over the final groups (execution/project_initial.go:52-144): every result term's expression is
// compiled over the groups' keys and aggregates; arithmetic and numeric functions become derived columns of an inner
// operator and are evaluated on the device by the same element-wise kernel as the arithmetic of WHERE / GROUP BY.
static n1k_status build_projection(n1k_handle* h) {
    auto* f = new n1k_handle();
    h->project = f;
    std::vector<std::unique_ptr<Expr>> trees;
    PlanError err;
    for (const ProjectTerm& t : h->plan.project) {
        auto e = parse_expression(group_paths(h, t.text), err);
        if (!e) {
            g_create_error = "projection: " + err.msg;
            return err.unsupported ? N1K_UNSUPPORTED : N1K_INVALID;
        }
        std::vector<std::string> paths;
        std::function<void(const Expr*)> walk = [&](const Expr* x) {
            if (x->kind == EK::Path) {
                if (std::find(f->plan.paths.begin(), f->plan.paths.end(), x->text) == f->plan.paths.end()) f->plan.paths.push_back(x->text);
                return;
            }
            for (auto& c : x->ch) walk(c.get());
        };
        walk(e.get());
        trees.push_back(std::move(e));
    }
    for (const std::string& p : f->plan.paths) {
        int idx;
        if (!group_path_index(h, p, idx)) {
            g_create_error = "a projection term refers to " + p + ", which is neither a group key nor an aggregate of the plan";
            return N1K_UNSUPPORTED;
        }
        h->project_cols.push_back(idx);
    }
    if (f->plan.paths.size() > (size_t)kMaxCols) {
        g_create_error = "projection over more than 16 keys / aggregates";
        return N1K_UNSUPPORTED;
    }
    for (auto& e : trees) {
        Operand o;
        if (!to_operand(f, e.get(), o, err)) {
            g_create_error = "projection: " + err.msg;
            return err.unsupported ? N1K_UNSUPPORTED : N1K_INVALID;
        }
        if (o.is_const && o.ctag == T_STRING) {
            g_create_error = "a string constant as a projection term does not run on the device";
            return N1K_UNSUPPORTED;
        }
        h->project_ops.push_back(o);
    }
    f->prog.ncols = (uint32_t)(f->plan.paths.size() + f->derived.size());
    return N1K_OK;
}

// ================================================================== C ABI

// No C++ exception leaves the library ("no C++ exceptions or abort() across the ABI"; a Go caller cannot
// unwind through cgo): allocation failures of the host containers become N1K_OOM, anything else N1K_DEVICE_ERROR.
template <class F>
static n1k_status guarded(const n1k_handle* ch, F&& f) noexcept {
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


extern "C" {

int n1k_abi_version(void) { return N1K_ABI_VERSION; }

int n1k_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* n1k_create_error(void) { return g_create_error.c_str(); }

n1k_status n1k_create(const char* plan_json, size_t len, n1k_handle** out) {
    return guarded(nullptr, [&]() -> n1k_status {
    if (out) *out = nullptr;
    if (!plan_json || !out) {
        g_create_error = "null argument";
        return N1K_INVALID;
    }
    auto* h = new n1k_handle();
    PlanError err;
    if (!parse_plan_json(plan_json, len, h->plan, err) || !compile_plan(h, err)) {
        g_create_error = err.msg;
        n1k_status st = err.unsupported ? N1K_UNSUPPORTED : N1K_INVALID;
        delete h;
        return st;
    }
    if (h->plan.has_having) {
        // The HAVING condition speaks of group keys and aggregates: replace each (longest text first) by a synthetic
        // leaf path and hand the result to an inner Filter-only operator, so that the device's predicate evaluator —
        // comparisons, arithmetic, 4-valued logic — is the one that decides, as it does for WHERE.
        std::string cond = h->plan.having_text;
        std::vector<std::pair<std::string, std::string>> subst;
        for (size_t a = 0; a < h->plan.aggs.size(); a++) subst.emplace_back(h->plan.aggs[a].text, "(`$g`.`a" + std::to_string(a) + "`)");
        for (size_t k = 0; k < h->plan.key_texts.size(); k++) subst.emplace_back(h->plan.key_texts[k], "(`$g`.`k" + std::to_string(k) + "`)");
        std::stable_sort(subst.begin(), subst.end(), [](const auto& x, const auto& y) { return x.first.size() > y.first.size(); });
        for (auto& sb : subst) {
            if (sb.first.empty()) continue;
            for (size_t pos = 0; (pos = cond.find(sb.first, pos)) != std::string::npos; pos += sb.second.size())
                cond.replace(pos, sb.first.size(), sb.second);
        }
        std::string js = "{\"#operator\":\"Filter\",\"condition\":\"";
        for (char c : cond) {
            if (c == '"' || c == '\\') js.push_back('\\');
            if ((unsigned char)c < 0x20) {
                char b[8];
                snprintf(b, sizeof b, "\\u%04x", (unsigned)c);
                js += b;
            } else
                js.push_back(c);
        }
        js += "\"}";
        n1k_status hst = n1k_create(js.c_str(), js.size(), &h->having);
        if (hst != N1K_OK) {
            g_create_error = "HAVING: " + g_create_error;
            delete h;
            return hst;
        }
        for (const std::string& p : h->having->plan.paths) {
            int idx = -1;
            char kind = 0;
            if (sscanf(p.c_str(), "(`$g`.`%c%d`)", &kind, &idx) != 2 || (kind != 'k' && kind != 'a') || idx < 0 ||
                (size_t)idx >= (kind == 'k' ? h->plan.key_texts.size() : h->plan.aggs.size())) {
                g_create_error = "HAVING refers to " + p + ", which is neither a group key nor an aggregate of the plan";
                n1k_destroy(h->having);
                delete h;
                return N1K_UNSUPPORTED;
            }
            h->having_cols.push_back(kind == 'k' ? idx : -idx - 1);
        }
    }
    if (h->plan.has_project) {
        n1k_status pst = build_projection(h);
        if (pst != N1K_OK) {
            if (h->having) n1k_destroy(h->having);
            h->having = nullptr;
            delete h;
            return pst;
        }
    }
    g_create_error.clear();
    *out = h;
    return N1K_OK;
    });
}

static void destroy_handle(n1k_handle* h);
void n1k_destroy(n1k_handle* h) {
    if (!h) return;
    try {
        destroy_handle(h);
    } catch (...) {
    }
}
static void destroy_handle(n1k_handle* h) {
    if (h->having) n1k_destroy(h->having);
    h->having = nullptr;
    if (h->project) n1k_destroy(h->project);
    h->project = nullptr;
    if (h->device_ready) {
        (void)hipSetDevice(h->device);
        if (h->stream) (void)hipStreamSynchronize(h->stream);
        drain_events(h);
        for (auto e : h->event_pool) (void)hipEventDestroy(e);
        h->d_rank.release();
        h->d_keys.release();
        h->d_acc.release();
        h->d_rep.release();
        h->d_slabs.release();
        h->d_block_sel.release();
        for (auto& b : h->dv_tags) b.release();
        for (auto& b : h->dv_payload) b.release();
        h->d_regions.release();
        h->d_set_table.release();
        for (uint32_t d = 0; d < kMaxDistinct; d++) {
            h->d_log_key[d].release();
            h->d_log_val[d].release();
            h->d_log_cls[d].release();
        }
        h->d_counters.release();
        for (uint32_t d = 0; d < kMaxDistinct; d++) h->d_log_word[d].release();
        h->d_part[0].release();
        h->d_part[1].release();
        for (auto& b : h->d_seg) b.release();
        h->d_wtable.release();
        h->d_hist.release();
        h->d_cursor.release();
        h->d_dcounts.release();
        h->d_word_hist.release();
        for (uint32_t d = 0; d < kMaxDistinct; d++) h->d_wregion[d].release();
        h->d_wcursor.release();
        h->d_woff.release();
        h->d_wgather.release();
        h->d_wide_int.release();
        h->d_wide_flt.release();
        if (h->pin_out) (void)hipHostFree(h->pin_out);
        if (h->pin_counters) (void)hipHostFree(h->pin_counters);
        h->d_emit.release();
        h->d_rregion.release();
        h->d_rbins.release();
        h->d_rcursor.release();
        for (int i = 0; i < 3; i++) {
            h->d_rec_key[i].release();
            for (uint32_t e = 0; e < kRecOperands; e++) {
                h->d_rec_pay[i][e].release();
                h->d_rec_tag[i][e].release();
            }
        }
        h->d_images.release();
        h->d_cand.release();
        h->d_topk.release();
        h->d_out2.release();
        for (int i = 0; i < 2; i++) {
            for (auto& b : h->st_tags[i]) b.release();
            for (auto& b : h->st_payload[i]) b.release();
            for (auto& b : h->st_codes[i]) b.release();
            if (h->st_free[i]) (void)hipEventDestroy(h->st_free[i]);
        }
        if (h->st_copied) (void)hipEventDestroy(h->st_copied);
        if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
        h->d_mask.release();
        h->d_tile_off.release();
        h->d_sel.release();
        h->d_tile_cnt.release();
        h->d_out.release();
        if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    }
    delete h;
}

n1k_status n1k_reset(n1k_handle* h) {
    return guarded(h, [&]() -> n1k_status {
    if (!h) return N1K_INVALID;
    h->stop_flag.store(0);
    h->row_base = 0;
    h->merged_groups_bound = 0;
    h->selected.clear();
    h->r_keys.clear();
    h->r_aggs.clear();
    h->r_parts.clear();
    h->r_rep.clear();
    memset(&h->stats, 0, sizeof h->stats);
    if (h->device_ready) {
        HIP_TRY(h, hipSetDevice(h->device));
        if (!h->events.empty()) {  // pushes that were never finished: their events must complete before reuse
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            drain_events(h);
        }
        h->stats.device_ms = 0;
        h->groups_seen = 0;
        h->out_count_dirty = false;
        h->pending.count = 0;
        // one launch, no host synchronisation: table back to empty and all counters / error flags to zero
        if (h->table.capacity) HIP_TRY(h, launch_init_table(h->prog, h->table, 0, h->table.capacity, h->d_counters.p, h->stream));
        else HIP_TRY(h, hipMemsetAsync(h->d_counters.p, 0, kCounters * sizeof(unsigned long long), h->stream));
        if (h->d_word_hist.p) HIP_TRY(h, hipMemsetAsync(h->d_word_hist.p, 0, kMaxDistinct * 256 * sizeof(unsigned long long), h->stream));
        if (h->d_wcursor.p) HIP_TRY(h, hipMemsetAsync(h->d_wcursor.p, 0, kMaxDistinct * kWordSubs * kCursorStride * sizeof(unsigned long long), h->stream));
        h->wregion_used = false;
        if (h->prog.wide_int) {
            const size_t n = (size_t)1 << h->prog.wide_bits;
            HIP_TRY(h, hipMemsetAsync(h->d_wide_int.p, 0xFF, n * 8, h->stream));
            HIP_TRY(h, hipMemsetAsync(h->d_wide_flt.p, 0xFF, n * 8, h->stream));
        }
    }
    return N1K_OK;
    });
}

void n1k_stop(n1k_handle* h) {
    if (h) h->stop_flag.store(1);
}

const char* n1k_last_error(const n1k_handle* h) { return h ? h->last_error.c_str() : "null handle"; }

uint32_t n1k_num_columns(const n1k_handle* h) { return h ? (uint32_t)h->plan.paths.size() : 0; }
const char* n1k_column_path(const n1k_handle* h, uint32_t i) {
    return (h && i < h->plan.paths.size()) ? h->plan.paths[i].c_str() : nullptr;
}
uint32_t n1k_num_keys(const n1k_handle* h) { return h ? (uint32_t)h->plan.keys.size() : 0; }
uint32_t n1k_num_aggregates(const n1k_handle* h) { return h ? (uint32_t)h->plan.aggs.size() : 0; }
const char* n1k_aggregate_name(const n1k_handle* h, uint32_t i) {
    return (h && i < h->agg_names.size()) ? h->agg_names[i].c_str() : nullptr;
}

uint32_t n1k_num_projection_terms(const n1k_handle* h) { return h ? (uint32_t)h->plan.project.size() : 0; }
const char* n1k_projection_expr(const n1k_handle* h, uint32_t i) {
    return (h && i < h->plan.project.size()) ? h->plan.project[i].text.c_str() : nullptr;
}
const char* n1k_projection_alias(const n1k_handle* h, uint32_t i) {
    return (h && i < h->plan.project.size()) ? h->plan.project[i].as.c_str() : nullptr;
}

n1k_status n1k_dict_intern(n1k_handle* h, uint32_t n, const uint64_t* offsets, const char* bytes, uint32_t* out_codes) {
    return guarded(h, [&]() -> n1k_status {
    if (!h || (n && (!offsets || !bytes || !out_codes))) return N1K_INVALID;
    for (uint32_t i = 0; i < n; i++) {
        if (offsets[i + 1] < offsets[i]) return fail(h, N1K_INVALID, "dictionary offsets are not monotone");
        out_codes[i] = intern(h, std::string(bytes + offsets[i], (size_t)(offsets[i + 1] - offsets[i])));
    }
    if (h->dict.size() >= 0xFFFFFFF0ull) return fail(h, N1K_OOM, "dictionary too large");
    return N1K_OK;
    });
}

uint32_t n1k_dict_size(const n1k_handle* h) { return h ? (uint32_t)h->dict.size() : 0; }

n1k_status n1k_dict_get(const n1k_handle* h, uint32_t code, const char** ptr, size_t* len) {
    return guarded(h, [&]() -> n1k_status {
    if (!h || !ptr || !len || code >= h->dict.size()) return N1K_INVALID;
    *ptr = h->dict[code].data();
    *len = h->dict[code].size();
    return N1K_OK;
    });
}

n1k_status n1k_set_option(n1k_handle* h, const char* name, int64_t value) {
    return guarded(h, [&]() -> n1k_status {
    if (!h || !name) return N1K_INVALID;
    std::string n = name;
    if (n == "agg_mode") h->opt_agg_mode = value;
    else if (n == "max_groups") h->opt_max_groups = value > 0 ? (uint64_t)value : 1;
    else if (n == "grid_blocks") h->opt_grid_blocks = (uint32_t)std::max<int64_t>(0, value);
    else if (n == "fast") h->opt_fast = value ? 1 : 0;
    else if (n == "spec") h->opt_spec = value ? 1 : 0;
    else if (n == "wide") h->opt_wide = value ? 1 : 0;
    else if (n == "fuse_arith") h->opt_fuse_arith = value ? 1 : 0;
    else if (n == "pinned_out") h->opt_pinned_out = value ? 1 : 0;
    else if (n == "lean_topk") h->opt_lean_topk = value ? 1 : 0;
    else if (n == "part_block") h->opt_part_block = value == 256 ? 256 : 512;
    else if (n == "part_subs") h->opt_part_subs = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 0), 2);  // 0 off, 1 large batches, 2 always (tests)
    else if (n == "part_per_cu") h->opt_part_per_cu = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 0), 8);
    else if (n == "jit") h->opt_jit = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 0), 2);
    else if (n == "jit_min_rows") h->opt_jit_min_rows = (uint64_t)std::max<int64_t>(value, 0);
    else if (n == "distinct_words") {
        if (h->layout_fixed) return fail(h, N1K_INVALID, "distinct_words must be set before the first batch");
        h->opt_distinct_words = value ? 1 : 0;
    } else if (n == "distinct_set_slots") {
        uint32_t v = 64;
        while (v < (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 64), 8192)) v <<= 1;
        h->opt_distinct_set_slots = v;
    } else if (n == "records") {
        h->opt_records = value ? 1 : 0;
    } else if (n == "rec_slots") {
        h->opt_rec_slots = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 0), 8192);
    } else if (n == "rec_bins") {
        uint32_t v = 0;
        if (value > 0) for (v = 1; v < (uint32_t)std::min<int64_t>(value, 256); v <<= 1) {}
        h->opt_rec_bins = v;
    } else if (n == "rec_slices") {
        h->opt_rec_slices = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 0), 64);
    } else if (n == "rec_scan_per_cu") {
        h->opt_rec_scan_per_cu = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 0), 8);
    } else if (n == "rec_block") {
        h->opt_rec_block = value <= 0 ? 0u : (value <= 256 ? 256u : 512u);
    } else if (n == "rec_unroll") {
        h->opt_rec_unroll = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 0), 8);
    } else if (n == "spec_debug") {
        h->opt_spec_debug = (uint32_t)value;
    } else if (n == "distinct_region_cap") {
        h->opt_region_cap = (uint64_t)std::max<int64_t>(value, 0);
    } else if (n == "dedupe_block") {
        if ((value | 1) != 257 && (value | 1) != 513 && (value | 1) != 1025) return fail(h, N1K_INVALID, "dedupe_block must be 256, 512 or 1024 (+1: probe word by word)");
        h->opt_dedupe_block = (uint32_t)value;
    } else if (n == "json_threads") {
        h->opt_json_threads = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 0), 64);
    } else if (n == "partition_min_rows") {
        h->opt_partition_min_rows = (uint64_t)std::max<int64_t>(value, 1);
    } else if (n == "partition_probe_rows") {
        h->opt_partition_probe_rows = (uint64_t)std::max<int64_t>(value, 1);
    } else if (n == "partition_min_groups") {
        h->opt_partition_min_groups = (uint64_t)std::max<int64_t>(value, 1);
    } else if (n == "partition_levels") {
        h->opt_partition_levels = (int32_t)std::min<int64_t>(std::max<int64_t>(value, -1), 2);
    } else if (n == "topk_min_groups") {
        h->opt_topk_min_groups = (uint64_t)std::max<int64_t>(value, 1);
    } else if (n == "distinct_levels") {
        h->opt_distinct_levels = (int32_t)std::min<int64_t>(std::max<int64_t>(value, -1), 2);
    } else if (n == "wide_values") {
        if (h->layout_fixed) return fail(h, N1K_INVALID, "wide_values must be set before the first batch");
        h->opt_wide_values = (uint64_t)std::max<int64_t>(value, 0);
    }
    else if (n == "slabs") h->opt_slabs = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 0), 2);  // 0 off, 1 auto, 2 always
    else if (n == "block") {
        if (value != 0 && value != 256 && value != 512 && value != 1024) return fail(h, N1K_INVALID, "block must be 0 (auto), 256, 512 or 1024");
        h->opt_block = (uint32_t)value;
    } else if (n == "rows_per_lane") {
        if (value != 2 && value != 4) return fail(h, N1K_INVALID, "rows_per_lane must be 2 or 4");
        h->opt_rows_per_lane = (uint32_t)value;
    } else if (n == "lds_bytes") h->opt_lds_bytes = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 1024), 160 * 1024);
    else if (n == "device") {
        if (h->device_ready) return fail(h, N1K_INVALID, "device must be chosen before the first push");
        h->device = (int)value;
    } else if (n == "stream") {
        if (h->device_ready) return fail(h, N1K_INVALID, "stream must be chosen before the first push");
        h->stream = (hipStream_t)(uintptr_t)value;
        h->own_stream = false;
    } else if (n == "rep_row") {
        if (h->layout_fixed) return fail(h, N1K_INVALID, "rep_row must be chosen before the first push");
        h->opt_rep_row = value ? 1 : 0;
        PlanError err;
        h->agg_names.clear();
        h->has_distinct = false;
        h->n_distinct = 0;
        if (!compile_plan(h, err)) return fail(h, N1K_INVALID, "%s", err.msg.c_str());
    } else
        return fail(h, N1K_INVALID, "unknown option %s", name);
    return N1K_OK;
    });
}

n1k_status n1k_push_device_batch(n1k_handle* h, const n1k_batch* batch) {
    return guarded(h, [&]() -> n1k_status {
    if (!h) return N1K_INVALID;
    return push_device(h, batch);
    });
}

n1k_status n1k_run_device_batch(n1k_handle* h, const n1k_batch* batch, n1k_result* out) {
    if (!h || !batch || !out) return N1K_INVALID;
    n1k_status st = n1k_reset(h);
    if (st == N1K_OK) st = n1k_push_device_batch(h, batch);
    if (st == N1K_OK) st = n1k_finish(h, out);
    return st;
}

n1k_status n1k_extract_json(n1k_handle* h, uint64_t ndocs, const uint64_t* offsets, const char* bytes, n1k_batch* out) {
    return guarded(h, [&]() -> n1k_status {
    if (!h || !out || (ndocs && (!offsets || !bytes))) return N1K_INVALID;
    if (h->json_paths_state == 0) {
        h->json_paths.resize(h->plan.paths.size());
        h->json_paths_state = 1;
        for (size_t c = 0; c < h->plan.paths.size(); c++)
            if (!parse_leaf_path(h->plan.paths[c], h->json_paths[c])) h->json_paths_state = -1;
    }
    if (h->json_paths_state < 0)
        return fail(h, N1K_UNSUPPORTED, "a leaf path of the plan is not a chain of field names: extract the columns yourself");
    const size_t np = h->json_paths.size();
    uint32_t nthreads = h->opt_json_threads ? h->opt_json_threads : std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
    nthreads = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(nthreads, ndocs / 2048 + 1));
    std::vector<JsonColumns> part(nthreads);
    std::vector<std::string> errs(nthreads);
    std::vector<long long> bad(nthreads, -1);
    std::vector<std::thread> workers;
    auto range = [&](uint32_t t) { return std::make_pair(ndocs * t / nthreads, ndocs * (t + 1) / nthreads); };
    // (a worker that throws — out of memory — reports it through its slot; a thread that cannot be started is replaced
    //  by the calling thread: no exception may leave a joinable std::thread behind)
    std::vector<char> threw(nthreads, 0);
    auto work = [&](uint32_t t) {
        try {
            auto r = range(t);
            bad[t] = extract_json_range(h->json_paths, offsets, bytes, r.first, r.second, part[t], errs[t]);
        } catch (...) {
            threw[t] = 1;
        }
    };
    workers.reserve(nthreads);
    std::vector<uint32_t> inline_ranges{0};
    for (uint32_t t = 1; t < nthreads; t++) {
        try {
            workers.emplace_back(work, t);
        } catch (...) {
            inline_ranges.push_back(t);
        }
    }
    for (uint32_t t : inline_ranges) work(t);
    for (auto& w : workers) w.join();
    for (uint32_t t = 0; t < nthreads; t++)
        if (threw[t]) return fail(h, N1K_OOM, "out of host memory while scanning the documents");
    for (uint32_t t = 0; t < nthreads; t++)
        if (bad[t] >= 0) return fail(h, N1K_INVALID, "document %lld is not valid JSON: %s", bad[t], errs[t].c_str());
    // one dictionary: the threads' local strings get the handle's codes
    h->js_tags.assign(np, std::vector<uint8_t>());
    h->js_payload.assign(np, std::vector<uint64_t>());
    for (size_t c = 0; c < np; c++) {
        h->js_tags[c].resize(ndocs);
        h->js_payload[c].resize(ndocs);
    }
    for (uint32_t t = 0; t < nthreads; t++) {
        auto r = range(t);
        std::vector<uint64_t> code(part[t].strings.size());
        for (size_t i = 0; i < code.size(); i++) code[i] = intern(h, part[t].strings[i]);
        for (size_t c = 0; c < np; c++) {
            const size_t n = (size_t)(r.second - r.first);
            memcpy(h->js_tags[c].data() + r.first, part[t].tags[c].data(), n);
            uint64_t* dst = h->js_payload[c].data() + r.first;
            const uint64_t* src = part[t].payload[c].data();
            const uint8_t* tg = part[t].tags[c].data();
            for (size_t i = 0; i < n; i++) dst[i] = tg[i] >= N1K_T_STRING ? code[src[i]] : src[i];
        }
    }
    h->js_cols.assign(np, n1k_col{});
    for (size_t c = 0; c < np; c++) {
        h->js_cols[c].kind = N1K_COL_TAGGED64;
        h->js_cols[c].tags = h->js_tags[c].data();
        h->js_cols[c].payload = h->js_payload[c].data();
    }
    out->nrows = ndocs;
    out->ncols = (uint32_t)np;
    out->cols = h->js_cols.data();
    return N1K_OK;
    });
}

n1k_status n1k_push_json(n1k_handle* h, uint64_t ndocs, const uint64_t* offsets, const char* bytes) {
    return guarded(h, [&]() -> n1k_status {
    n1k_batch b{};
    n1k_status st = n1k_extract_json(h, ndocs, offsets, bytes, &b);
    if (st != N1K_OK) return st;
    return n1k_push_batch(h, &b);
    });
}

n1k_status n1k_push_batch(n1k_handle* h, const n1k_batch* batch) {
    return guarded(h, [&]() -> n1k_status {
    if (!h) return N1K_INVALID;
    if (h->stop_flag.load()) return fail(h, N1K_STOPPED, "operator was stopped");
    n1k_status st = ensure_device(h);
    if (st != N1K_OK) return st;
    st = validate_batch(h, batch);
    if (st != N1K_OK) return st;
    std::vector<n1k_col> dcols;
    st = stage_host_batch(h, batch, dcols);
    if (st != N1K_OK) return st;
    n1k_batch db = *batch;
    db.cols = dcols.data();
    st = push_device(h, &db);
    n1k_status st2 = staged_batch_issued(h);
    return st != N1K_OK ? st : st2;
    });
}

n1k_status n1k_sync(n1k_handle* h) {
    return guarded(h, [&]() -> n1k_status {
    if (!h) return N1K_INVALID;
    if (!h->device_ready) return N1K_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    unsigned long long counters[kCounters] = {0};
    HIP_TRY(h, hipMemcpyAsync(counters, h->d_counters.p, sizeof counters, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    drain_events(h);
    if (h->plan.has_group) {
        h->stats.rows_selected = counters[0];
        h->stats.groups_out = h->pending.count ? h->pending.count : counters[1];  // groups so far (in the table, or in the kept region)
    }
    h->stats.wide_key_values = counters[13];
    return N1K_OK;
    });
}

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
    const size_t shmem = distinct_dedupe_lds(D) + 512;
    const uint32_t by_threads = 2048u / std::max(256u, h->opt_dedupe_block & ~1u);
    const uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(by_threads, (160 * 1024) / shmem));
    return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(nbins, (uint64_t)h->num_cus * per_cu));
}

// COUNT(DISTINCT) over the one-word members of one aggregate: radix partition of the word log until a bin's distinct
// words fit an LDS set, per-bin LDS sets, member counts added to the groups' set sizes (see n1k_kernels.hip).
static n1k_status distinct_words_finish(n1k_handle* h, const AggSpec& ag, uint64_t nwords, bool hist_counted = true,
                                        const uint64_t* log = nullptr) {
    const uint32_t set_slots = h->opt_distinct_set_slots;
    const uint64_t per_bin = std::max<uint64_t>(set_slots / 4, 16);  // expected distinct words per final bin: load <= 1/4
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
    HIP_TRY(h, launch_distinct_dedupe(h->prog, h->table, D, dedupe_grid(h, D, nbins), h->opt_dedupe_block, h->stream));
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
static n1k_status distinct_regions_finish(n1k_handle* h, const AggSpec& ag, uint64_t nover, bool force_exact, bool* deferred) {
    const uint32_t li = ag.log_index;
    const uint64_t cap = h->wregion_cap;
    unsigned long long* const cursors = h->d_wcursor.p + (size_t)li * kWordSubs * kCursorStride;
    const uint32_t set_slots = h->opt_distinct_set_slots;
    const uint64_t per_bin = std::max<uint64_t>(set_slots / 4, 16);
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
        const uint32_t wpr = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(8, region_tiles));
        HIP_TRY(h, launch_radix_scatter_words(R, wpr, bps, h->stream));
        DedupeArgs D{};
        D.words = R.dst;
        D.bin_count = h->d_cursor.p;
        D.count_stride = 1;
        D.bin_stride = bin_cap;
        D.nbins = (uint32_t)nbins;
        D.set_slots = set_slots;
        D.key_shift = h->nw_val_bits + 3;
        D.glob_off = ag.glob_off;
        D.counts = h->d_dcounts.p;
        D.overflow = d_overflow;
        dedupe_counters(h, D);
        HIP_TRY(h, launch_distinct_dedupe(h->prog, h->table, D, dedupe_grid(h, D, D.nbins), h->opt_dedupe_block, h->stream));
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

// value.Collate for result values (value/value.go:69-79 type order; integer.go:100-118, float.go:106-172,
// string.go:116-130, boolean.go:99-113).  Arrays / objects collate element-wise in the reference: not ordered here.
static int host_collate(const n1k_handle* h, const n1k_value& a, const n1k_value& b, bool* unsupported) {
    auto cls = [](uint8_t t) -> int {
        switch (t) {
            case N1K_T_MISSING: return 0;
            case N1K_T_NULL: return 1;
            case N1K_T_FALSE: case N1K_T_TRUE: return 2;
            case N1K_T_INT: case N1K_T_FLOAT: return 3;
            case N1K_T_STRING: return 4;
            case N1K_T_ARRAY: return 5;
            default: return 6;
        }
    };
    const int ca = cls(a.tag), cb = cls(b.tag);
    if (ca != cb) return ca < cb ? -1 : 1;
    switch (ca) {
        case 2: return (int)(a.tag == N1K_T_TRUE) - (int)(b.tag == N1K_T_TRUE);
        case 3: {
            if (a.tag == N1K_T_INT && b.tag == N1K_T_INT) return a.v.i < b.v.i ? -1 : (a.v.i > b.v.i ? 1 : 0);
            if (a.tag != b.tag) {
                // The reference compares an int with a float through float64 (value/float.go:106-121) while two ints
                // compare exactly: beyond 2^53 that is not transitive (858 < 859, yet both equal the float between
                // them) and sort.Sort's result is then arbitrary.  A sort needs a strict weak order: the int and the
                // float are compared exactly here — the same answer wherever the reference's is well defined.
                const bool a_int = a.tag == N1K_T_INT;
                const int64_t i = a_int ? a.v.i : b.v.i;
                const double d = a_int ? b.v.f : a.v.f;
                int c;  // sign of (i - d)
                if (d != d) c = 1;  // NaN sorts first
                else if (d >= 9223372036854775808.0) c = -1;
                else if (d < -9223372036854775808.0) c = 1;
                else {
                    const int64_t t = (int64_t)d;  // truncation toward zero, exact in range
                    if (i != t) c = i < t ? -1 : 1;
                    else {
                        const double frac = d - (double)t;
                        c = frac > 0 ? -1 : (frac < 0 ? 1 : 0);
                    }
                }
                return a_int ? c : -c;
            }
            const double x = a.v.f, y = b.v.f;
            if (x != x) return (y != y) ? 0 : -1;  // NaN sorts first
            if (y != y) return 1;
            return x < y ? -1 : (x > y ? 1 : 0);
        }
        case 4: {
            const std::string& x = h->dict[a.v.code];
            const std::string& y = h->dict[b.v.code];
            const int c = memcmp(x.data(), y.data(), std::min(x.size(), y.size()));
            if (c) return c < 0 ? -1 : 1;
            return x.size() < y.size() ? -1 : (x.size() > y.size() ? 1 : 0);
        }
        case 5:
        case 6: {
            // arrays element by element, objects by size and sorted names (value/array.go, value/object.go:511-556): on
            // the host, over the canonical texts the dictionary holds
            if (a.v.code == b.v.code) return 0;
            int c = 0;
            if (a.v.code >= h->dict.size() || b.v.code >= h->dict.size() || !json_text_collate(h->dict[a.v.code], h->dict[b.v.code], c))
                *unsupported = true;
            return c;
        }
        default: return 0;
    }
}

// HAVING (the Filter after FinalGroup, planner/build_select_sub.go:295; execution/filter.go:49-61 over rows whose
// aggregates are read from the "aggregates" attachment, algebra/aggregate.go:97-118): the final groups become a batch
// of the inner Filter-only operator — one column per key / aggregate its condition names — and its survivors stay.
static n1k_status having_groups(n1k_handle* h, uint64_t& ng) {
    n1k_handle* f = h->having;
    const size_t nk = h->plan.keys.size(), na = h->plan.aggs.size(), nc = h->having_cols.size();
    if (ng == 0) return N1K_OK;
    if (f->device < 0 && !f->device_ready) f->device = h->device;
    std::vector<std::vector<uint8_t>> tags(nc, std::vector<uint8_t>((size_t)ng));
    std::vector<std::vector<uint64_t>> pay(nc, std::vector<uint64_t>((size_t)ng));
    h->having_codes.resize(h->dict.size(), 0xFFFFFFFFu);
    for (size_t c = 0; c < nc; c++) {
        const int src = h->having_cols[c];
        for (uint64_t g = 0; g < ng; g++) {
            const n1k_value& v = src >= 0 ? h->r_keys[g * nk + (size_t)src] : h->r_aggs[g * na + (size_t)(-src - 1)];
            tags[c][g] = v.tag;
            uint64_t p = v.v.code;
            if (v.tag >= N1K_T_STRING) {  // the inner operator has its own dictionary
                uint32_t& m = h->having_codes[(size_t)p];
                if (m == 0xFFFFFFFFu) m = intern(f, h->dict[(size_t)p]);
                p = m;
            }
            pay[c][g] = p;
        }
    }
    std::vector<n1k_col> cols(nc ? nc : 1);
    for (size_t c = 0; c < nc; c++) {
        cols[c].kind = N1K_COL_TAGGED64;
        cols[c].tags = tags[c].data();
        cols[c].payload = pay[c].data();
    }
    n1k_batch b{};
    b.nrows = ng;
    b.ncols = (uint32_t)nc;
    b.cols = cols.data();
    n1k_result res{};
    n1k_status st = n1k_reset(f);
    if (st == N1K_OK) st = n1k_push_batch(f, &b);
    if (st == N1K_OK) st = n1k_finish(f, &res);
    if (st != N1K_OK) return fail(h, st, "HAVING: %s", n1k_last_error(f));
    const uint64_t keep = res.nselected;
    std::vector<n1k_value> keys(keep * nk), aggs(keep * na);
    std::vector<n1k_partial> parts(h->r_parts.empty() ? 0 : keep * na);
    std::vector<uint64_t> rep(keep);
    for (uint64_t i = 0; i < keep; i++) {
        const uint64_t g = res.selected[i];
        for (size_t k = 0; k < nk; k++) keys[i * nk + k] = h->r_keys[g * nk + k];
        for (size_t a = 0; a < na; a++) {
            aggs[i * na + a] = h->r_aggs[g * na + a];
            if (!parts.empty()) parts[i * na + a] = h->r_parts[g * na + a];
        }
        rep[i] = g < h->r_rep.size() ? h->r_rep[g] : ~0ull;
    }
    h->r_keys.swap(keys);
    h->r_aggs.swap(aggs);
    h->r_parts.swap(parts);
    h->r_rep.swap(rep);
    ng = keep;
    return N1K_OK;
}

// InitialProject over the final groups (execution/project_initial.go:100-144): the value of every result term per
// group.  Terms that are a key, an aggregate or a constant are copied; the others were compiled into derived columns of
// the inner operator and are evaluated on the device over the groups as one column batch.
static n1k_status project_groups(n1k_handle* h, uint64_t ng) {
    n1k_handle* f = h->project;
    const size_t nk = h->plan.keys.size(), na = h->plan.aggs.size(), nc = h->project_cols.size(), nt = h->project_ops.size();
    h->r_proj.assign((size_t)ng * nt, n1k_value{});
    if (ng == 0 || nt == 0) return N1K_OK;
    auto source = [&](size_t c, uint64_t g) -> const n1k_value& {
        const int src = h->project_cols[c];
        return src >= 0 ? h->r_keys[g * nk + (size_t)src] : h->r_aggs[g * na + (size_t)(-src - 1)];
    };
    std::vector<std::vector<uint8_t>> dt(f->derived.size());
    std::vector<std::vector<uint64_t>> dp(f->derived.size());
    if (!f->derived.empty()) {
        std::vector<std::vector<uint8_t>> tags(nc, std::vector<uint8_t>((size_t)ng));
        std::vector<std::vector<uint64_t>> pay(nc, std::vector<uint64_t>((size_t)ng));
        for (size_t c = 0; c < nc; c++)
            for (uint64_t g = 0; g < ng; g++) {
                const n1k_value& v = source(c, g);
                tags[c][g] = v.tag;
                pay[c][g] = v.v.code;  // (strings keep this operator's codes: arithmetic over a non-number is NULL anyway)
            }
        std::vector<n1k_col> cols(nc ? nc : 1);
        for (size_t c = 0; c < nc; c++) {
            cols[c].kind = N1K_COL_TAGGED64;
            cols[c].tags = tags[c].data();
            cols[c].payload = pay[c].data();
        }
        n1k_batch b{};
        b.nrows = ng;
        b.ncols = (uint32_t)nc;
        b.cols = cols.data();
        if (f->device < 0 && !f->device_ready) f->device = h->device;
        n1k_status st = ensure_device(f);
        if (st == N1K_OK) st = validate_batch(f, &b);
        std::vector<n1k_col> dcols;
        if (st == N1K_OK) st = stage_host_batch(f, &b, dcols);
        if (st == N1K_OK) {
            n1k_batch db = b;
            db.cols = dcols.data();
            st = bind_columns(f, &db);  // launches the element-wise kernel of every derived column
        }
        if (st != N1K_OK) return fail(h, st, "projection: %s", n1k_last_error(f));
        for (size_t d = 0; d < f->derived.size(); d++) {
            dt[d].resize((size_t)ng);
            dp[d].resize((size_t)ng);
            HIP_TRY(h, hipMemcpyAsync(dt[d].data(), f->dv_tags[d].p, (size_t)ng, hipMemcpyDeviceToHost, f->stream));
            HIP_TRY(h, hipMemcpyAsync(dp[d].data(), f->dv_payload[d].p, (size_t)ng * 8, hipMemcpyDeviceToHost, f->stream));
        }
        HIP_TRY(h, hipStreamSynchronize(f->stream));
    }
    for (size_t t = 0; t < nt; t++) {
        const Operand& o = h->project_ops[t];
        for (uint64_t g = 0; g < ng; g++) {
            n1k_value& v = h->r_proj[g * nt + t];
            if (o.is_const) {
                v.tag = (uint8_t)o.ctag;
                v.v.code = o.cpayload;
            } else if (o.col < nc) {
                v = source(o.col, g);
            } else {
                v.tag = dt[o.col - nc][g];
                v.v.code = dp[o.col - nc][g];
            }
        }
    }
    return N1K_OK;
}

// Order / Offset / Limit over the final groups (execution/order.go:121-169: term by term Collate, DESC flips it;
// order_limit.go keeps offset + limit rows; offset.go / limit.go then cut).  sort.Sort is not stable, so the order
// among rows that tie on every term is unspecified in the reference too; here ties keep table order.
static n1k_status order_groups(n1k_handle* h, uint64_t& ng) {
    const ParsedPlan& pl = h->plan;
    const size_t nk = pl.keys.size(), na = pl.aggs.size(), np = h->r_proj.empty() ? 0 : h->project_ops.size();
    std::vector<uint32_t> perm((size_t)ng);
    for (size_t i = 0; i < perm.size(); i++) perm[i] = (uint32_t)i;
    bool unsupported = false;
    if (pl.has_order) {
        auto less = [&](uint32_t x, uint32_t y) {
            for (const OrderTerm& t : pl.order) {
                const n1k_value& a = t.proj_index >= 0 ? h->r_proj[x * np + t.proj_index]
                                     : t.key_index >= 0 ? h->r_keys[x * nk + t.key_index] : h->r_aggs[x * na + t.agg_index];
                const n1k_value& b = t.proj_index >= 0 ? h->r_proj[y * np + t.proj_index]
                                     : t.key_index >= 0 ? h->r_keys[y * nk + t.key_index] : h->r_aggs[y * na + t.agg_index];
                const int c = host_collate(h, a, b, &unsupported);
                if (c) return t.desc ? c > 0 : c < 0;
            }
            return false;
        };
        const uint64_t keep = pl.limit >= 0 ? std::min<uint64_t>(ng, (uint64_t)pl.offset + (uint64_t)pl.limit) : ng;
        if (keep < ng) std::partial_sort(perm.begin(), perm.begin() + keep, perm.end(), [&](uint32_t x, uint32_t y) {
            return less(x, y) || (!less(y, x) && x < y);
        });
        else std::stable_sort(perm.begin(), perm.end(), less);
        if (unsupported) return fail(h, N1K_UNSUPPORTED_DATA, "ORDER BY over array / object values is outside the device subset");
    }
    const uint64_t first = std::min<uint64_t>(ng, (uint64_t)pl.offset);
    const uint64_t last = pl.limit >= 0 ? std::min<uint64_t>(ng, first + (uint64_t)pl.limit) : ng;
    std::vector<n1k_value> keys((last - first) * nk), aggs((last - first) * na);
    std::vector<n1k_partial> parts((last - first) * na);
    std::vector<uint64_t> rep(last - first);
    std::vector<n1k_value> proj((last - first) * np);
    for (uint64_t i = first; i < last; i++) {
        const uint32_t g = perm[i];
        for (size_t t = 0; t < np; t++) proj[(i - first) * np + t] = h->r_proj[g * np + t];
        for (size_t k = 0; k < nk; k++) keys[(i - first) * nk + k] = h->r_keys[g * nk + k];
        for (size_t a = 0; a < na; a++) {
            aggs[(i - first) * na + a] = h->r_aggs[g * na + a];
            if (!h->r_parts.empty()) parts[(i - first) * na + a] = h->r_parts[g * na + a];
        }
        rep[i - first] = g < h->r_rep.size() ? h->r_rep[g] : ~0ull;
    }
    h->r_keys.swap(keys);
    h->r_aggs.swap(aggs);
    h->r_parts.swap(parts);
    h->r_rep.swap(rep);
    if (np) h->r_proj.swap(proj);
    ng = last - first;
    return N1K_OK;
}

// ARRAY_AGG / ARRAY_AGG(DISTINCT) (algebra/agg_array.go:86-145, agg_array_distinct.go:86-127): the scan logged every
// operand that is not MISSING with its group's packed key; FinalGroup wrote each group's packed key into the
// representative-row slot.  Here the operands are handed to their groups, sorted by value.Collate (ComputeFinal sorts
// with value.NewSorter), de-duplicated for DISTINCT (value.Set: integral floats join the ints), and the array's
// canonical JSON text becomes a dictionary entry: the aggregate's value is an ARRAY like any other on this path.
static n1k_status array_agg_groups(n1k_handle* h, uint64_t ng, const unsigned long long* counters) {
    const size_t na = h->plan.aggs.size();
    std::unordered_map<uint64_t, uint64_t> group_of;
    group_of.reserve((size_t)ng * 2);
    for (uint64_t g = 0; g < ng; g++) group_of.emplace(h->r_rep[g], g);
    for (size_t a = 0; a < na; a++) {
        const AggSpec& ag = h->prog.aggs[a];
        if (ag.kind != AGG_ARRAY) continue;
        const uint64_t n = std::min<uint64_t>(counters[8 + ag.log_index], h->log_capacity);
        std::vector<uint64_t> keys((size_t)n), vals((size_t)n);
        std::vector<uint8_t> tags((size_t)n);
        if (n) {
            HIP_TRY(h, hipMemcpyAsync(keys.data(), h->d_log_key[ag.log_index].p, n * 8, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(h, hipMemcpyAsync(vals.data(), h->d_log_val[ag.log_index].p, n * 8, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(h, hipMemcpyAsync(tags.data(), h->d_log_cls[ag.log_index].p, n, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(h, hipStreamSynchronize(h->stream));
        }
        std::vector<std::vector<n1k_value>> members((size_t)ng);
        for (uint64_t i = 0; i < n; i++) {
            auto it = group_of.find(keys[i]);
            if (it == group_of.end()) continue;  // (a group the top-k filter left on the device)
            n1k_value v{};
            v.tag = tags[i];
            v.v.code = vals[i];
            members[(size_t)it->second].push_back(v);
        }
        bool unsupported = false;
        std::string text;
        for (uint64_t g = 0; g < ng; g++) {
            auto& m = members[(size_t)g];
            n1k_value& out = h->r_aggs[g * na + a];
            memset(&out, 0, sizeof out);
            out.tag = N1K_T_NULL;  // Default(): NULL (agg_array.go:77); an empty DISTINCT set is NULL too
            if (m.empty()) continue;
            std::stable_sort(m.begin(), m.end(), [&](const n1k_value& x, const n1k_value& y) { return host_collate(h, x, y, &unsupported) < 0; });
            if (h->plan.aggs[a].distinct)
                m.erase(std::unique(m.begin(), m.end(), [&](const n1k_value& x, const n1k_value& y) { return host_collate(h, x, y, &unsupported) == 0; }),
                        m.end());
            text.assign("[");
            for (size_t i = 0; i < m.size(); i++) {
                if (i) text.push_back(',');
                switch (m[i].tag) {
                    case N1K_T_NULL: text += "null"; break;
                    case N1K_T_FALSE: text += "false"; break;
                    case N1K_T_TRUE: text += "true"; break;
                    case N1K_T_INT: text += std::to_string((long long)m[i].v.i); break;
                    case N1K_T_FLOAT: format_float(m[i].v.f, text); break;
                    case N1K_T_STRING: json_quote(h->dict[(size_t)m[i].v.code], text); break;
                    default: text += h->dict[(size_t)m[i].v.code]; break;  // arrays / objects: their canonical text
                }
            }
            text.push_back(']');
            out.tag = N1K_T_ARRAY;
            out.v.code = intern(h, text);
        }
        if (unsupported) return fail(h, N1K_UNSUPPORTED_DATA, "array_agg over values whose collation is outside the subset");
    }
    for (uint64_t g = 0; g < ng; g++) h->r_rep[g] = ~0ull;  // (the slot carried the packed keys)
    return N1K_OK;
}

n1k_status n1k_finish(n1k_handle* h, n1k_result* out) {
    return guarded(h, [&]() -> n1k_status {
    if (!h || !out) return N1K_INVALID;
    memset(out, 0, sizeof *out);
    if (h->stop_flag.load()) return fail(h, N1K_STOPPED, "operator was stopped");
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
        const bool topk_forced = pl.has_order && pl.limit >= 0 && !pl.has_having && h->opt_topk_min_groups < 4096;  // tests
        // (a table of millions of slots is not worth scanning twice: the sized pass alone then)
        if (pl.has_group && !h->has_distinct && h->table.capacity && h->table.capacity <= (1u << 20) && !topk_forced && !h->pending.count) {
            spec_groups = std::min<uint64_t>(h->table.capacity, 4096);
            const size_t off_aggs = spec_groups * rec_keys, off_parts = off_aggs + spec_groups * rec_aggs,
                         off_rep = off_parts + spec_groups * rec_parts, total = off_rep + spec_groups * 8;
            HIP_TRY(h, h->d_out.ensure(total + 16));
            if (h->pin_cap < total + sizeof counters) {
                if (h->pin_out) (void)hipHostFree(h->pin_out);
                h->pin_out = nullptr;
                h->pin_cap = 0;
                HIP_TRY(h, hipHostMalloc((void**)&h->pin_out, total + sizeof counters, hipHostMallocDefault));
                h->pin_cap = total + sizeof counters;
            }
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
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            memcpy(counters, h->pin_out + total, sizeof counters);
        } else {
            HIP_TRY(h, hipMemcpyAsync(counters, h->d_counters.p, sizeof counters, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(h, hipStreamSynchronize(h->stream));
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
            if (lean)
                HIP_TRY(h, launch_finalize_region(h->prog, h->d_emit.p, h->pending.cap, ng, nullptr, nullptr, nullptr, nullptr, h->d_errp,
                                                  h->stream, nullptr, (OutValue*)d, pl.order[0].key_index >= 0,
                                                  (uint32_t)(pl.order[0].key_index >= 0 ? pl.order[0].key_index : pl.order[0].agg_index)));
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
                HIP_TRY(h, h->d_cand.ensure(ng));
                HIP_TRY(h, h->d_topk.ensure(topk_state_bytes()));
                n1k_status rst = ensure_rank(h);
                if (rst != N1K_OK) return rst;
                const OutValue* vals = lean || t0.key_index >= 0 ? (const OutValue*)d : (const OutValue*)(d + off_aggs);
                HIP_TRY(h, launch_topk_select(h->prog, vals, lean ? 1u : (t0.key_index >= 0 ? nk : na),
                                              lean ? 0u : (uint32_t)(t0.key_index >= 0 ? t0.key_index : t0.agg_index), ng, t0.desc, keep,
                                              h->d_images.p, h->d_topk.p, h->d_cand.p, h->stream));
                unsigned long long ncand = 0;
                HIP_TRY(h, hipMemcpyAsync(&ncand, h->d_topk.p + topk_ncand_offset(), sizeof ncand, hipMemcpyDeviceToHost, h->stream));
                HIP_TRY(h, hipStreamSynchronize(h->stream));
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
            h->out_host.resize(copy_bytes);
            uint32_t veto[2] = {0, 0};
            HIP_TRY(h, hipMemcpyAsync(h->out_host.data(), src, copy_bytes, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(h, hipMemcpyAsync(&err_flags, h->d_errp, 4, hipMemcpyDeviceToHost, h->stream));
            if (sets_deferred) HIP_TRY(h, hipMemcpyAsync(veto, h->d_counters.p + 20, 8, hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            if (sets_deferred && (veto[0] | veto[1])) {
                // a set or a bin overflowed on the optimistic path: no counts were added; once more, exactly
                sets_exact = true;
                ng = counters[1];
                goto redo_sets;
            }
            hp = h->out_host.data();
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
    if (err_flags & ERR_TABLE_FULL)
        return fail(h, N1K_OOM, "group table capacity exceeded: raise the max_groups option (now %llu)",
                    (unsigned long long)h->opt_max_groups);
    if (err_flags & ERR_EXCHANGE_WIDE)
        return fail(h, N1K_UNSUPPORTED, "a sender's group keys hold float / wide integer values: use the row exchange");
    if (err_flags & ERR_EXCHANGE_OVERFLOW)
        return fail(h, N1K_OOM, "a sender's partial-group region overflowed: raise the region capacity");
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

n1k_status n1k_order_rows(n1k_handle* h, uint64_t ngroups, const n1k_value* keys, const n1k_value* aggs, n1k_result* out) {
    return guarded(h, [&]() -> n1k_status {
    if (!h || !out || (ngroups && ((!keys && !h->plan.keys.empty()) || (!aggs && !h->plan.aggs.empty())))) return N1K_INVALID;
    if (!h->plan.has_group) return fail(h, N1K_INVALID, "no groups in a Filter-only plan");
    const size_t nk = h->plan.keys.size(), na = h->plan.aggs.size();
    memset(out, 0, sizeof *out);
    out->nkeys = (uint32_t)nk;
    out->naggs = (uint32_t)na;
    for (uint64_t i = 0; i < ngroups * (nk + na); i++) {
        const n1k_value& v = i < ngroups * nk ? keys[i] : aggs[i - ngroups * nk];
        if (v.tag >= N1K_T_STRING && v.v.code >= h->dict.size()) return fail(h, N1K_INVALID, "a value's dictionary code is unknown to this handle");
    }
    h->r_keys.assign(keys, keys + ngroups * nk);
    h->r_aggs.assign(aggs, aggs + ngroups * na);
    h->r_parts.clear();
    h->r_rep.assign((size_t)ngroups, ~0ull);
    h->r_proj.clear();
    uint64_t ng = ngroups;
    n1k_status st = h->plan.has_project ? project_groups(h, ng) : N1K_OK;  // (sort terms may name projection aliases)
    if (st != N1K_OK) return st;
    st = order_groups(h, ng);
    if (st != N1K_OK) return st;
    out->nproj = h->plan.has_project ? (uint32_t)h->project_ops.size() : 0;
    out->proj = out->nproj ? h->r_proj.data() : nullptr;
    out->ngroups = ng;
    out->keys = h->r_keys.data();
    out->aggs = h->r_aggs.data();
    out->partials = nullptr;
    out->rep_row = h->r_rep.data();
    return N1K_OK;
    });
}

n1k_status n1k_jit_check(n1k_handle* h, const uint32_t* col_kinds, uint32_t ncols, char* log, size_t loglen) {
    return guarded(h, [&]() -> n1k_status {
    if (!h || !col_kinds) return N1K_INVALID;
    if (log && loglen) log[0] = 0;
    if (ncols != h->plan.paths.size()) return fail(h, N1K_INVALID, "plan has %zu columns", h->plan.paths.size());
    if (!h->plan.has_group) return fail(h, N1K_UNSUPPORTED, "Filter-only plans use fixed kernels");
    // fix the key layout from the column kinds alone (no data, no device)
    std::vector<n1k_col> cols(ncols ? ncols : 1);
    for (uint32_t c = 0; c < ncols; c++) cols[c].kind = col_kinds[c];
    n1k_batch b{};
    b.nrows = 0;
    b.ncols = ncols;
    b.cols = cols.data();
    if (!h->layout_fixed) {
        n1k_status st = fix_layout(h, &b);
        if (st != N1K_OK) return st;
    }
    for (uint32_t c = 0; c < ncols; c++) h->prog.cols[c].kind = col_kinds[c] == N1K_COL_DICT32 ? COLK_DICT32 : COLK_TAGGED64;
    for (uint32_t c = ncols; c < h->prog.ncols; c++) h->prog.cols[c].kind = COLK_TAGGED64;
    FastArgs F;
    const uint32_t max_slots = (uint32_t)std::min<uint64_t>((156u * 1024u) / (h->prog.lds_words * 8), 1u << 15);
    // (plans with arithmetic: the shape that evaluates the nodes in registers, as run_group_batch would choose it)
    const bool fuse = !h->derived.empty() && h->opt_fuse_arith;
    if (!build_fast_args(h, max_slots, F, fuse)) return fail(h, N1K_UNSUPPORTED, "the plan shape is outside the bounded family");
    SpecSig sig = make_plan_sig(h, F);
    std::string l;
    bool ok = jit_compile_check(sig, &l);
    if (ok && build_fast_args(h, 1u << 15, F, fuse, true)) {  // the same shape's partition kernels (multi-GPU row exchange)
        SpecSig ps = make_plan_sig(h, F);
        ps.mode = 1;
        ps.hashed = 0;
        ok = jit_compile_check(ps, &l);
    }
    if (log && loglen) snprintf(log, loglen, "%s", l.c_str());
    return ok ? N1K_OK : fail(h, N1K_DEVICE_ERROR, "run-time compilation failed: %s", l.substr(0, 300).c_str());
    });
}

n1k_status n1k_get_stats(const n1k_handle* h, n1k_stats* out) {
    return guarded(h, [&]() -> n1k_status {
    if (!h || !out) return N1K_INVALID;
    // account pushes whose events have completed meanwhile (no waiting: hipEventQuery)
    n1k_handle* m = const_cast<n1k_handle*>(h);
    while (!m->events.empty() && m->events.front().second && hipEventQuery(m->events.front().second) == hipSuccess) {
        auto pr = m->events.front();
        float ms = 0.f;
        if (pr.first && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) m->stats.device_ms += ms;
        if (pr.first) m->event_pool.push_back(pr.first);
        m->event_pool.push_back(pr.second);
        m->events.erase(m->events.begin());
    }
    *out = h->stats;
    return N1K_OK;
    });
}

// Filter + hash partition of a bound batch (bind_columns(h, b, defer = true) came first): the run-time-built kernel of the
// plan's shape when there is one (large batches, or jit = 2: n1k_spec.h scan_spec_partition_body — wide loads, arithmetic
// in registers, survivors written in runs), else the interpreting partition_kernel over materialised derived columns.
static n1k_status run_partition(n1k_handle* h, const n1k_batch* b, PartArgs& A) {
    // packed regions (the row exchange): A.sub_rows = rows per sub-region; whoever writes dense runs converts the counts
    const uint64_t seg_rows = A.region_bytes ? A.sub_rows : 0;
    A.nsub = 1;
    if (b->nrows == 0) {
        if (seg_rows) HIP_TRY(h, launch_dense_to_segments(A.counts, A.nparts, A.count_stride, seg_rows, h->stream));
        return materialize_derived(h, b);
    }
    const uint64_t n = b->nrows;
    FastArgs F;
    const JitKernel* jit = nullptr;
    const bool fuse = !h->derived.empty() && !h->derived_ready;
    if (h->opt_spec && h->opt_jit && (h->opt_jit == 2 || n >= h->opt_jit_min_rows) && n < (1ull << 31) && (!fuse || h->opt_fuse_arith) &&
        sizeof(Program) + sizeof(FastArgs) + sizeof(PartArgs) + 64 <= 4096 && build_fast_args(h, 1u << 15, F, fuse, true)) {
        // what the staging needs in LDS (n1k_spec.h PartLds: 2048 rows x (9 B per TAGGED64 column, 4 B per DICT32 column, 1))
        size_t lds = 2048 + 2048;
        for (uint32_t c = 0; c < F.ncols; c++) lds += 2048u * (F.cols[c].kind == COLK_DICT32 ? 4u : 9u);
        if (lds <= 60 * 1024) {
            SpecSig sig = make_plan_sig(h, F);
            sig.mode = 1;
            sig.hashed = 0;  // (no table in this mode)
            jit = jit_get(sig);
            if (jit->failed || !jit->part_wide) {
                h->jit_log = jit->log;
                jit = nullptr;
            }
        }
    }
    hipEvent_t e0 = get_event(h), e1 = get_event(h);
    if (jit) {
        bool aligned = true;
        for (uint32_t c = 0; c < F.ncols; c++) {
            F.cols[c] = h->prog.cols[c];
            aligned &= ((uintptr_t)F.cols[c].tags % 2 == 0) && ((uintptr_t)F.cols[c].payload % 16 == 0) && ((uintptr_t)F.cols[c].codes % 8 == 0);
        }
        const bool wide = aligned && h->opt_wide && n >= 2;
        F.nrows = (uint32_t)n;
        F.row_base = h->row_base;
        F.err_flags = h->d_errp;
        // 256-thread workgroups (tiles of 1024 rows, one in flight, six per CU: many independent workgroups overlap the wait
        // for each tile's reservation) or 512-thread ones (2048 rows, two tiles in flight, two per CU)
        const uint32_t pblock = wide && h->opt_part_block == 256 ? 256u : 512u;
        const uint64_t tiles = (n + pblock * 4 - 1) / (pblock * 4);
        uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)h->num_cus * (h->opt_part_per_cu ? h->opt_part_per_cu : (pblock == 256 ? 6u : 2u)), tiles));
        // many tiles: every destination's region in kRowSubs sub-regions with their own counters, workgroups dealt round-robin
        // (tile t goes to sub-region t % kRowSubs: an even share of the rows whatever their order)
        if (seg_rows && h->opt_part_subs && (tiles >= 4096 || h->opt_part_subs == 2)) {
            A.nsub = kRowSubs;
            grid = (grid + kRowSubs - 1) / kRowSubs * kRowSubs;
        }
        if (e0) (void)hipEventRecord(e0, h->stream);
        HIP_TRY(h, jit_launch_partition(jit, h->prog, F, A, grid, wide, pblock, h->stream));
        h->stats.spec_kernel = F.nderived ? 3u : 2u;
    } else {
        n1k_status st = materialize_derived(h, b);
        if (st != N1K_OK) return st;
        const uint64_t ntiles = (n + 2047) / 2048;  // partition_kernel<4, 512>
        const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)h->num_cus * 4, ntiles));
        if (e0) (void)hipEventRecord(e0, h->stream);
        HIP_TRY(h, launch_partition(h->prog, A, grid, h->stream));
        h->stats.spec_kernel = 0;
    }
    if (seg_rows && A.nsub == 1) HIP_TRY(h, launch_dense_to_segments(A.counts, A.nparts, A.count_stride, seg_rows, h->stream));
    if (e1) (void)hipEventRecord(e1, h->stream);
    h->events.emplace_back(e0, e1);
    return N1K_OK;
}

n1k_status n1k_partition_device_batch(n1k_handle* h, const n1k_batch* batch, uint32_t nparts, uint64_t capacity_rows,
                                      const n1k_col* out_cols, uint64_t* out_counts) {
    return guarded(h, [&]() -> n1k_status {
    if (!h || !batch || !out_cols || !out_counts || nparts == 0) return N1K_INVALID;
    if (nparts > kMaxParts) return fail(h, N1K_INVALID, "at most %u destinations per partition call", kMaxParts);
    if (h->stop_flag.load()) return fail(h, N1K_STOPPED, "operator was stopped");
    if (!h->plan.has_group) return fail(h, N1K_INVALID, "partitioning needs group keys");
    if (sizeof(Program) + sizeof(PartArgs) + 64 > 4096) return fail(h, N1K_UNSUPPORTED, "kernel arguments exceed 4 KiB");
    n1k_status st = ensure_device(h);
    if (st != N1K_OK) return st;
    st = validate_batch(h, batch);
    if (st != N1K_OK) return st;
    if (!h->layout_fixed) {
        st = fix_layout(h, batch);
        if (st != N1K_OK) return st;
    }
    PartArgs A{};
    st = bind_columns(h, batch, true);
    if (st != N1K_OK) return st;
    A.ncopy = (uint32_t)h->plan.paths.size();  // derived columns are recomputed by the receiver
    for (uint32_t c = 0; c < A.ncopy; c++) {
        if (out_cols[c].kind != batch->cols[c].kind) return fail(h, N1K_INVALID, "output column %u has another kind", c);
        A.out_tags[c] = (uint8_t*)out_cols[c].tags;
        A.out_payload[c] = (uint64_t*)out_cols[c].payload;
        A.out_codes[c] = (uint32_t*)out_cols[c].codes;
    }
    st = ensure_rank(h);
    if (st != N1K_OK) return st;
    A.nrows = batch->nrows;
    A.capacity = capacity_rows;
    A.nparts = nparts;
    A.counts = (unsigned long long*)out_counts;
    A.err_flags = h->d_errp;
    HIP_TRY(h, hipMemsetAsync(out_counts, 0, nparts * sizeof(uint64_t), h->stream));
    st = run_partition(h, batch, A);
    if (st != N1K_OK) return st;
    uint32_t err_flags = 0;
    HIP_TRY(h, hipMemcpyAsync(&err_flags, h->d_errp, 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (err_flags & ERR_TABLE_FULL) return fail(h, N1K_OOM, "partition region capacity (%llu rows) exceeded", (unsigned long long)capacity_rows);
    if (err_flags & ERR_UNPACKABLE_KEY) return fail(h, N1K_UNSUPPORTED_DATA, "a group key value does not fit the packed key");
    if (err_flags & ERR_UNSUPPORTED_VALUE) return fail(h, N1K_UNSUPPORTED_DATA, "a value outside the device subset was met");
    h->stats.rows_in += batch->nrows;
    h->stats.batches += 1;
    return N1K_OK;
    });
}

uint32_t n1k_partial_words(const n1k_handle* h) { return h ? h->prog.glob_words : 0; }

uint64_t n1k_partial_region_bytes(const n1k_handle* h, uint64_t capacity_groups) {
    if (!h) return 0;
    return 8ull * (2 + capacity_groups * (1 + (uint64_t)h->prog.glob_words));
}

n1k_status n1k_export_partials_async(n1k_handle* h, uint32_t nparts, uint64_t capacity_groups, void* out) {
    return guarded(h, [&]() -> n1k_status {
    if (!h || !out || nparts == 0 || capacity_groups == 0) return N1K_INVALID;
    if (h->pending.count) {
        n1k_status pst = flush_pending(h);
        if (pst != N1K_OK) return pst;
    }
    if (!h->plan.has_group) return fail(h, N1K_INVALID, "no groups in a Filter-only plan");
    if (h->has_distinct) return fail(h, N1K_UNSUPPORTED, "DISTINCT sets do not travel with partial groups");
    n1k_status st = ensure_device(h);
    if (st != N1K_OK) return st;
    uint64_t region_words = 2 + capacity_groups * (1 + (uint64_t)h->prog.glob_words);
    HIP_TRY(h, hipMemsetAsync(out, 0, (size_t)nparts * region_words * 8, h->stream));  // headers (and padding) to zero
    if (h->table.capacity)
        HIP_TRY(h, launch_export_partials(h->prog, h->table, nparts, capacity_groups, (uint64_t*)out, region_words,
                                          h->d_errp, h->stream));
    return N1K_OK;
    });
}

n1k_status n1k_export_partials_device(n1k_handle* h, uint32_t nparts, uint64_t capacity_groups, void* out) {
    return guarded(h, [&]() -> n1k_status {
    n1k_status st = n1k_export_partials_async(h, nparts, capacity_groups, out);
    if (st != N1K_OK) return st;
    uint32_t err_flags = 0;
    unsigned long long sel = 0, wide = 0;
    HIP_TRY(h, hipMemcpyAsync(&err_flags, h->d_errp, 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(&sel, h->d_counters.p, sizeof sel, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(&wide, h->d_counters.p + 13, sizeof wide, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->stats.rows_selected = sel;
    h->stats.wide_key_values = wide;
    // codes of the wide-value tables mean nothing on another device: such groups travel as rows instead
    if (wide) {
        HIP_TRY(h, hipMemsetAsync(h->d_errp, 0, 4, h->stream));  // a region overflow of the abandoned export is moot
        return fail(h, N1K_UNSUPPORTED, "group keys hold %llu float / wide integer values: use the row exchange", wide);
    }
    if (err_flags & ERR_TABLE_FULL) {
        HIP_TRY(h, hipMemsetAsync(h->d_errp, 0, 4, h->stream));
        return fail(h, N1K_OOM, "more than %llu groups for one destination: raise the region capacity",
                    (unsigned long long)capacity_groups);
    }
    return N1K_OK;
    });
}

n1k_status n1k_merge_partials_device(n1k_handle* h, uint32_t nregions, uint64_t capacity_groups, const void* in) {
    return guarded(h, [&]() -> n1k_status {
    if (!h || !in || nregions == 0 || capacity_groups == 0) return N1K_INVALID;
    if (h->has_distinct) return fail(h, N1K_UNSUPPORTED, "DISTINCT sets do not travel with partial groups");
    if (!h->layout_fixed) return fail(h, N1K_INVALID, "merge needs the key layout: push a batch (even an empty one) first");
    if (h->pending.count) {
        n1k_status pst = flush_pending(h);
        if (pst != N1K_OK) return pst;
    }
    n1k_status st = ensure_device(h);
    if (st != N1K_OK) return st;
    // the incoming groups bound the growth of the table
    uint64_t saved = h->row_base;
    h->row_base += (uint64_t)nregions * capacity_groups;
    st = ensure_table(h, 0);
    h->row_base = saved;
    if (st != N1K_OK) return st;
    uint64_t region_words = 2 + capacity_groups * (1 + (uint64_t)h->prog.glob_words);
    HIP_TRY(h, launch_merge_partials(h->prog, h->table, nregions, capacity_groups, (const uint64_t*)in, region_words,
                                     h->d_errp, h->d_counters.p + 1, h->stream));
    h->merged_groups_bound += (uint64_t)nregions * capacity_groups;
    return N1K_OK;
    });
}

n1k_status n1k_export_groups(n1k_handle* h, const void** blob, size_t* len) {
    return guarded(h, [&]() -> n1k_status {
    if (!h || !blob || !len) return N1K_INVALID;
    n1k_status st = ensure_device(h);
    if (st != N1K_OK) return st;
    st = flush_pending(h);
    if (st != N1K_OK) return st;
    unsigned long long ng = 0;
    HIP_TRY(h, hipMemcpyAsync(&ng, h->d_counters.p + 1, sizeof ng, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    uint64_t cap = std::max<uint64_t>(ng, 1);
    uint64_t bytes = n1k_partial_region_bytes(h, cap);
    DevBuf<uint64_t> tmp;
    HIP_TRY(h, tmp.ensure(bytes / 8));
    st = n1k_export_partials_device(h, 1, cap, tmp.p);
    if (st == N1K_OK) {
        h->export_blob.resize(bytes + 16);
        uint64_t hdr[2] = {0x4e314b5041525431ull /* "N1KPART1" */, cap};
        memcpy(h->export_blob.data(), hdr, 16);
        hipError_t e = hipMemcpy(h->export_blob.data() + 16, tmp.p, bytes, hipMemcpyDeviceToHost);
        if (e != hipSuccess) st = fail(h, N1K_DEVICE_ERROR, "copy of exported groups failed: %s", hipGetErrorString(e));
    }
    tmp.release();
    if (st != N1K_OK) return st;
    *blob = h->export_blob.data();
    *len = h->export_blob.size();
    return N1K_OK;
    });
}

n1k_status n1k_merge_groups(n1k_handle* h, const void* blob, size_t len) {
    return guarded(h, [&]() -> n1k_status {
    if (!h || !blob || len < 32) return N1K_INVALID;
    uint64_t hdr[2];
    memcpy(hdr, blob, 16);
    if (hdr[0] != 0x4e314b5041525431ull) return fail(h, N1K_INVALID, "not an exported group blob");
    uint64_t cap = hdr[1];
    if (n1k_partial_region_bytes(h, cap) + 16 != len) return fail(h, N1K_INVALID, "blob does not match this plan");
    n1k_status st = ensure_device(h);
    if (st != N1K_OK) return st;
    DevBuf<uint64_t> tmp;
    HIP_TRY(h, tmp.ensure((len - 16) / 8));
    HIP_TRY(h, hipMemcpy(tmp.p, (const char*)blob + 16, len - 16, hipMemcpyHostToDevice));
    st = n1k_merge_partials_device(h, 1, cap, tmp.p);
    if (st == N1K_OK) HIP_TRY(h, hipStreamSynchronize(h->stream));
    tmp.release();
    return st;
    });
}

// ---------------------------------------------------------------- multi-GPU: RCCL behind the C ABI
//
// One communicator per rank (one rank per GPU; on one node every GPU pair has its own xGMI link, so the grouped
// send / recv of an all-to-all keeps all of a GPU's links busy at once).  Everything below is ordered on the sending
// handle's stream; the receiving handle's stream waits on an event; nothing waits on the host before n1k_finish.

}  // extern "C" (the communicator struct is C++)

// Loopback transport (n1k_comm_create_loopback): the ranks are threads of ONE process sharing one device — every
// collective is a rendezvous (barrier), device-to-device copies out of the peers' buffers, and a second rendezvous before
// anybody reuses its send buffer.  It exists so that the world_size > 1 code paths of the exchange (region offsets, header
// lists, verdicts, segmented receives, agreed capacities) can be run and checked on a single GPU; RCCL refuses two ranks on
// one device.
struct LoopHub {
    int world = 1;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    std::vector<const void*> ptr;
    std::vector<unsigned long long> val;
    int refs = 0;
    void barrier() {
        std::unique_lock<std::mutex> lk(mu);
        const uint64_t g = generation;
        if (++arrived == world) {
            arrived = 0;
            generation++;
            cv.notify_all();
        } else
            cv.wait(lk, [&] { return generation != g; });
    }
};

struct n1k_comm {
    ncclComm_t comm = nullptr;
    LoopHub* hub = nullptr;  // non-null: loopback transport
    int rank = 0, world = 1, device = 0;
    DevBuf<char> send, recv, gsend, grecv;
    DevBuf<unsigned long long> scalar;
    hipEvent_t ev = nullptr;
    std::string last_error;
    uint64_t gather_cap = 1024;    // records per slot of n1k_gather_groups (the same on every rank, see there)
    std::vector<char> ghost;       // gathered records on the host
    std::vector<n1k_value> gkeys, gaggs;
};

namespace {

n1k_status cfail(n1k_comm* c, n1k_status st, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->last_error = buf;
    g_create_error = buf;
    return st;
}

#define NCCL_TRY(c, expr)                                                                                   \
    do {                                                                                                    \
        ncclResult_t _r = (expr);                                                                           \
        if (_r != ncclSuccess) return cfail(c, N1K_DEVICE_ERROR, "%s failed: %s", #expr, ncclGetErrorString(_r)); \
    } while (0)
#define CHIP_TRY(c, expr)                                                                                   \
    do {                                                                                                    \
        hipError_t _e = (expr);                                                                             \
        if (_e != hipSuccess)                                                                               \
            return cfail(c, _e == hipErrorOutOfMemory ? N1K_OOM : N1K_DEVICE_ERROR, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

// all-to-all of equal regions: region p of `send` goes to rank p, region s of `recv` comes from rank s.  This rank's own
// region is not copied: the caller reads it where it lies (`self` returns its address).
// loopback: what every peer published, copied (or read) by everybody between two rendezvous
template <class Copy>
n1k_status loop_collective(n1k_comm* c, const void* send, hipStream_t st, Copy copy) {
    // (a rank whose own part fails still keeps both rendezvous: its peers must not be left waiting for it)
    hipError_t e = hipStreamSynchronize(st);
    c->hub->ptr[c->rank] = e == hipSuccess ? send : nullptr;
    c->hub->barrier();  // every rank's send buffer is complete and published
    if (e == hipSuccess) {
        for (int p = 0; p < c->world; p++)
            if (!c->hub->ptr[p]) e = hipErrorUnknown;  // (a peer failed before publishing)
    }
    if (e == hipSuccess) e = copy();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    c->hub->barrier();  // everybody has read everybody: the send buffers may be overwritten
    return e == hipSuccess ? N1K_OK : cfail(c, N1K_DEVICE_ERROR, "loopback: collective failed: %s", hipGetErrorString(e));
}

n1k_status all_gather_bytes(n1k_comm* c, const char* send, char* recv, size_t bytes, hipStream_t st) {
    if (c->hub)
        return loop_collective(c, send, st, [&]() -> hipError_t {
            for (int p = 0; p < c->world; p++) {
                hipError_t e = hipMemcpyAsync(recv + (size_t)p * bytes, c->hub->ptr[p], bytes, hipMemcpyDeviceToDevice, st);
                if (e != hipSuccess) return e;
            }
            return hipSuccess;
        });
    NCCL_TRY(c, ncclAllGather(send, recv, bytes, ncclChar, c->comm, st));
    return N1K_OK;
}

n1k_status all_to_all_regions(n1k_comm* c, const char* send, char* recv, size_t region, hipStream_t st, const char** self) {
    *self = send + (size_t)c->rank * region;
    if (c->hub)
        return loop_collective(c, send, st, [&]() -> hipError_t {
            for (int p = 0; p < c->world; p++) {
                if (p == c->rank) continue;
                hipError_t e = hipMemcpyAsync(recv + (size_t)p * region, (const char*)c->hub->ptr[p] + (size_t)c->rank * region, region,
                                              hipMemcpyDeviceToDevice, st);
                if (e != hipSuccess) return e;
            }
            return hipSuccess;
        });
    NCCL_TRY(c, ncclGroupStart());
    for (int p = 0; p < c->world; p++) {
        if (p == c->rank) continue;
        NCCL_TRY(c, ncclSend(send + (size_t)p * region, region, ncclChar, p, c->comm, st));
        NCCL_TRY(c, ncclRecv(recv + (size_t)p * region, region, ncclChar, p, c->comm, st));
    }
    NCCL_TRY(c, ncclGroupEnd());
    return N1K_OK;
}

int sender_column(const n1k_handle* snd, const std::string& path) {
    for (size_t j = 0; j < snd->plan.paths.size(); j++)
        if (snd->plan.paths[j] == path) return (int)j;
    return -1;
}

// the receiving handle learns the key layout (column kinds) and the dictionary from the sending one: both were built
// from the same plan, in one process
n1k_status prepare_receiver(n1k_handle* snd, n1k_handle* rcv) {
    if (!snd->layout_fixed) return fail(snd, N1K_INVALID, "the sender has seen no batch yet");
    for (size_t i = rcv->dict.size(); i < snd->dict.size(); i++)
        if (intern(rcv, snd->dict[i]) != (uint32_t)i) return fail(rcv, N1K_INVALID, "sender and receiver dictionaries differ");
    n1k_status st = ensure_device(rcv);
    if (st != N1K_OK) return st;
    if (!rcv->layout_fixed) {
        // (the receiver has no Filter: its columns are the sender's in another order — matched by their path text)
        std::vector<n1k_col> cols(std::max<size_t>(1, rcv->plan.paths.size()));
        for (size_t i = 0; i < rcv->plan.paths.size(); i++) {
            const int j = sender_column(snd, rcv->plan.paths[i]);
            if (j < 0) return fail(rcv, N1K_INVALID, "the receiver's column %s is not a column of the sender", rcv->plan.paths[i].c_str());
            cols[i].kind = snd->col_kinds[j];
        }
        n1k_batch b{};
        b.ncols = (uint32_t)rcv->plan.paths.size();
        b.cols = cols.data();
        st = push_device(rcv, &b);  // an empty batch: fixes the layout, runs nothing
    }
    return st;
}

n1k_status order_streams(n1k_comm* c, n1k_handle* snd, n1k_handle* rcv) {
    if (snd->stream == rcv->stream) return N1K_OK;
    CHIP_TRY(c, hipEventRecord(c->ev, snd->stream));
    CHIP_TRY(c, hipStreamWaitEvent(rcv->stream, c->ev, 0));
    return N1K_OK;
}

// layout of one packed row region for `cap` rows (kRowSubs sub-regions of cap / kRowSubs rows) of the plan's input columns:
// the header (sub-region x's row count at word x * kCursorStride, the verdict in word 1), then per column its arrays, each
// starting on a 16-byte boundary
size_t row_region_layout(const n1k_handle* h, uint64_t cap, std::vector<size_t>& off_a, std::vector<size_t>& off_b) {
    size_t at = (size_t)kRowSubs * kCursorStride * 8;  // header: the sub-regions' counts, 128 bytes apart; verdict in word 1
    const size_t nc = h->plan.paths.size();
    off_a.assign(nc, 0);
    off_b.assign(nc, 0);
    auto pad = [](size_t x) { return (x + 15) / 16 * 16; };
    for (size_t i = 0; i < nc; i++) {
        if (h->col_kinds[i] == N1K_COL_DICT32) {
            off_a[i] = at;
            at = pad(at + cap * 4);
        } else {
            off_a[i] = at;  // payload
            at = pad(at + cap * 8);
            off_b[i] = at;  // tags
            at = pad(at + cap);
        }
    }
    return (at + 127) / 128 * 128;
}

}  // namespace

extern "C" {

n1k_status n1k_comm_unique_id(void* id) {
    return guarded(nullptr, [&]() -> n1k_status {
        if (!id) return N1K_INVALID;
        static_assert(sizeof(ncclUniqueId) == N1K_COMM_ID_BYTES, "N1K_COMM_ID_BYTES must be sizeof(ncclUniqueId)");
        ncclUniqueId u;
        NCCL_TRY(nullptr, ncclGetUniqueId(&u));
        memcpy(id, &u, sizeof u);
        return N1K_OK;
    });
}

n1k_status n1k_comm_create(const void* id, int rank, int world, int device, n1k_comm** out) {
    return guarded(nullptr, [&]() -> n1k_status {
        if (out) *out = nullptr;
        if (!id || !out || world < 1 || rank < 0 || rank >= world || world > (int)kMaxParts) return cfail(nullptr, N1K_INVALID, "bad communicator arguments");
        auto* c = new n1k_comm();
        c->rank = rank;
        c->world = world;
        c->device = device;
        auto bail = [&](n1k_status st) {
            delete c;
            return st;
        };
        if (hipSetDevice(device) != hipSuccess) return bail(cfail(nullptr, N1K_DEVICE_ERROR, "no HIP device %d", device));
        ncclUniqueId u;
        memcpy(&u, id, sizeof u);
        ncclResult_t r = ncclCommInitRank(&c->comm, world, u, rank);
        if (r != ncclSuccess) return bail(cfail(nullptr, N1K_DEVICE_ERROR, "ncclCommInitRank failed: %s", ncclGetErrorString(r)));
        if (hipEventCreateWithFlags(&c->ev, hipEventDisableTiming) != hipSuccess) return bail(cfail(nullptr, N1K_DEVICE_ERROR, "hipEventCreate failed"));
        *out = c;
        return N1K_OK;
    });
}

n1k_status n1k_comm_create_loopback(int world, int device, n1k_comm** out) {
    return guarded(nullptr, [&]() -> n1k_status {
        if (!out || world < 1 || world > (int)kMaxParts) return cfail(nullptr, N1K_INVALID, "bad communicator arguments");
        if (hipSetDevice(device) != hipSuccess) return cfail(nullptr, N1K_DEVICE_ERROR, "no HIP device %d", device);
        auto* hub = new LoopHub();
        hub->world = world;
        hub->ptr.assign(world, nullptr);
        hub->val.assign(world, 0);
        hub->refs = world;
        for (int r = 0; r < world; r++) {
            auto* c = new n1k_comm();
            c->rank = r;
            c->world = world;
            c->device = device;
            c->hub = hub;
            (void)hipEventCreateWithFlags(&c->ev, hipEventDisableTiming);
            out[r] = c;
        }
        return N1K_OK;
    });
}

void n1k_comm_destroy(n1k_comm* c) {
    if (!c) return;
    try {
        (void)hipSetDevice(c->device);
        if (c->hub) {
            bool last;
            {
                std::lock_guard<std::mutex> lk(c->hub->mu);
                last = --c->hub->refs == 0;
            }
            if (last) delete c->hub;
        }
        if (c->comm) (void)ncclCommDestroy(c->comm);
        if (c->ev) (void)hipEventDestroy(c->ev);
        c->send.release();
        c->recv.release();
        c->gsend.release();
        c->grecv.release();
        c->scalar.release();
        delete c;
    } catch (...) {
    }
}

const char* n1k_comm_last_error(const n1k_comm* c) { return c ? c->last_error.c_str() : g_create_error.c_str(); }
int n1k_comm_rank(const n1k_comm* c) { return c ? c->rank : -1; }
int n1k_comm_world(const n1k_comm* c) { return c ? c->world : 0; }

n1k_status n1k_comm_max_u64(n1k_comm* c, n1k_handle* h, uint64_t value, uint64_t* out) {
    return guarded(h, [&]() -> n1k_status {
        if (!c || !h || !out) return N1K_INVALID;
        n1k_status st = ensure_device(h);
        if (st != N1K_OK) return st;
        if (c->hub) {  // loopback: values through the hub
            c->hub->val[c->rank] = value;
            c->hub->barrier();
            unsigned long long mx = 0;
            for (int p = 0; p < c->world; p++) mx = std::max(mx, c->hub->val[p]);
            c->hub->barrier();
            *out = mx;
            return N1K_OK;
        }
        HIP_TRY(h, c->scalar.ensure(4));
        unsigned long long v = value, m = 0;
        HIP_TRY(h, hipMemcpyAsync(c->scalar.p, &v, 8, hipMemcpyHostToDevice, h->stream));
        NCCL_TRY(c, ncclAllReduce(c->scalar.p, c->scalar.p + 1, 1, ncclUint64, ncclMax, c->comm, h->stream));
        HIP_TRY(h, hipMemcpyAsync(&m, c->scalar.p + 1, 8, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        *out = m;
        return N1K_OK;
    });
}

n1k_status n1k_exchange_partials(n1k_comm* c, n1k_handle* sender, n1k_handle* receiver, uint64_t capacity_groups, int gathered) {
    return guarded(sender, [&]() -> n1k_status {
        if (!c || !sender || !receiver || capacity_groups == 0) return N1K_INVALID;
        n1k_status st = prepare_receiver(sender, receiver);
        if (st != N1K_OK) return st == N1K_INVALID && sender->last_error.empty() ? fail(sender, st, "%s", receiver->last_error.c_str()) : st;
        const size_t region = (size_t)n1k_partial_region_bytes(sender, capacity_groups);
        const uint32_t nsend = gathered ? 1u : (uint32_t)c->world;
        HIP_TRY(sender, c->send.ensure(region * nsend));
        HIP_TRY(sender, c->recv.ensure(region * (size_t)c->world));
        st = n1k_export_partials_async(sender, nsend, capacity_groups, c->send.p);
        if (st != N1K_OK) return st;
        if (gathered) {
            // every rank ends with every rank's partial groups: no second collective for the result
            st = all_gather_bytes(c, c->send.p, c->recv.p, region, sender->stream);
            if (st != N1K_OK) return fail(sender, st, "%s", c->last_error.c_str());
        } else {
            const char* self = nullptr;
            st = all_to_all_regions(c, c->send.p, c->recv.p, region, sender->stream, &self);
            if (st != N1K_OK) return fail(sender, st, "%s", c->last_error.c_str());
            // (this rank's own region joins the received ones by a device copy of G groups, not through the fabric)
            HIP_TRY(sender, hipMemcpyAsync(c->recv.p + (size_t)c->rank * region, self, region, hipMemcpyDeviceToDevice, sender->stream));
        }
        st = order_streams(c, sender, receiver);
        if (st != N1K_OK) return fail(sender, st, "%s", c->last_error.c_str());
        st = n1k_merge_partials_device(receiver, (uint32_t)c->world, capacity_groups, c->recv.p);
        if (st != N1K_OK) return fail(sender, st, "receiver: %s", receiver->last_error.c_str());
        return N1K_OK;
    });
}

n1k_status n1k_exchange_rows(n1k_comm* c, n1k_handle* sender, const n1k_batch* batch, n1k_handle* receiver, uint64_t capacity_rows) {
    return guarded(sender, [&]() -> n1k_status {
        if (!c || !sender || !receiver || !batch || capacity_rows == 0) return N1K_INVALID;
        if (!sender->plan.has_group) return fail(sender, N1K_INVALID, "the row exchange partitions on group keys");
        if (sender->stop_flag.load()) return fail(sender, N1K_STOPPED, "operator was stopped");
        n1k_status st = ensure_device(sender);
        if (st != N1K_OK) return st;
        st = validate_batch(sender, batch);
        if (st != N1K_OK) return st;
        if (!sender->layout_fixed) {
            st = fix_layout(sender, batch);
            if (st != N1K_OK) return st;
        }
        st = prepare_receiver(sender, receiver);
        if (st != N1K_OK) return st;
        const uint64_t cap = (capacity_rows + 16 * kRowSubs - 1) / (16 * kRowSubs) * (16 * kRowSubs);  // kRowSubs sub-regions of whole 16-row groups
        if (cap >= (1ull << 31)) return fail(sender, N1K_INVALID, "row regions hold fewer than 2^31 rows");
        std::vector<size_t> off_a, off_b;
        const size_t region = row_region_layout(sender, cap, off_a, off_b);
        const uint32_t P = (uint32_t)c->world;
        HIP_TRY(sender, c->send.ensure(region * P));
        HIP_TRY(sender, c->recv.ensure(region * P));
        // 1. Filter + hash partition on the group key values into the packed regions (headers zeroed first)
        for (uint32_t d = 0; d < P; d++) HIP_TRY(sender, hipMemsetAsync(c->send.p + (size_t)d * region, 0, (size_t)kRowSubs * kCursorStride * 8, sender->stream));
        st = bind_columns(sender, batch, true);
        if (st != N1K_OK) return st;
        st = ensure_rank(sender);
        if (st != N1K_OK) return st;
        PartArgs A{};
        A.nrows = batch->nrows;
        A.capacity = cap;
        A.nparts = P;
        A.ncopy = (uint32_t)sender->plan.paths.size();
        A.counts = (unsigned long long*)c->send.p;
        A.count_stride = (uint32_t)(region / 8);
        A.region_bytes = region;
        A.sub_rows = cap / kRowSubs;
        A.err_flags = sender->d_errp;
        for (uint32_t i = 0; i < A.ncopy; i++) {
            if (sender->col_kinds[i] == N1K_COL_DICT32) A.out_codes[i] = (uint32_t*)(c->send.p + off_a[i]);
            else {
                A.out_payload[i] = (uint64_t*)(c->send.p + off_a[i]);
                A.out_tags[i] = (uint8_t*)(c->send.p + off_b[i]);
            }
        }
        st = run_partition(sender, batch, A);
        if (st != N1K_OK) return st;
        sender->stats.rows_in += batch->nrows;
        sender->stats.batches += 1;
        // 2. ONE all-to-all: counts, verdicts and rows of every column travel in the same region
        const char* self = nullptr;
        st = all_to_all_regions(c, c->send.p, c->recv.p, region, sender->stream, &self);
        if (st != N1K_OK) return fail(sender, st, "%s", c->last_error.c_str());
        st = order_streams(c, sender, receiver);
        if (st != N1K_OK) return fail(sender, st, "%s", c->last_error.c_str());
        // 3. the owner's InitialGroup over what it received: one batch per source, each with its row count on the device.
        //    (Headers are checked first: a sender that overflowed voids the step on every rank.)
        std::vector<const char*> src(P);
        for (uint32_t sidx = 0; sidx < P; sidx++) src[sidx] = (int)sidx == c->rank ? self : c->recv.p + (size_t)sidx * region;
        {
            HeaderList H{};
            for (uint32_t sidx = 0; sidx < P; sidx++) H.h[sidx] = (unsigned long long*)src[sidx];
            HIP_TRY(receiver, launch_exchange_verdict(H, P, receiver->d_errp, receiver->stream));
        }
        for (uint32_t sidx = 0; sidx < P; sidx++) {
            const uint32_t rnc = (uint32_t)receiver->plan.paths.size();
            std::vector<n1k_col> cols(std::max<size_t>(1, rnc));
            for (uint32_t i = 0; i < rnc; i++) {
                const int j = sender_column(sender, receiver->plan.paths[i]);  // (prepare_receiver checked that it exists)
                cols[i].kind = sender->col_kinds[j];
                if (cols[i].kind == N1K_COL_DICT32) cols[i].codes = (const uint32_t*)(src[sidx] + off_a[j]);
                else {
                    cols[i].payload = (const uint64_t*)(src[sidx] + off_a[j]);
                    cols[i].tags = (const uint8_t*)(src[sidx] + off_b[j]);
                }
            }
            n1k_batch rb{};
            rb.nrows = cap;
            rb.ncols = rnc;
            rb.cols = cols.data();
            receiver->push_seg_counts = (const unsigned long long*)src[sidx];  // (the region's header)
            receiver->push_nseg = kRowSubs;
            receiver->push_seg_rows = cap / kRowSubs;
            st = push_device(receiver, &rb);
            receiver->push_seg_counts = nullptr;
            receiver->push_nseg = 0;
            if (st != N1K_OK) return fail(sender, st, "receiver: %s", receiver->last_error.c_str());
        }
        return N1K_OK;
    });
}

n1k_status n1k_rows_step(n1k_comm* c, n1k_handle* sender, const n1k_batch* batch, n1k_handle* receiver, n1k_handle* merger,
                         uint64_t capacity_rows, n1k_result* out, int* worst_status) {
    if (!c || !sender || !batch || !receiver || !merger || !out || !worst_status) return N1K_INVALID;
    *worst_status = N1K_OK;
    n1k_status st = n1k_reset(receiver);
    if (st == N1K_OK) st = n1k_reset(sender);  // (the sender holds no groups in this mode; its counters and timers start over)
    if (st == N1K_OK) st = n1k_exchange_rows(c, sender, batch, receiver, capacity_rows);
    if (st != N1K_OK) return st;
    n1k_result local;
    st = n1k_finish(receiver, &local);
    // a region that overflowed fails the step on EVERY rank alike (the verdict travelled in the headers): no gather, the
    // caller enlarges the regions and repeats; any other failure is this owner's alone and travels in the gather
    if (st == N1K_OOM && receiver->last_error.find("region") != std::string::npos) return st;
    n1k_status gs = n1k_gather_groups_status(c, merger, st == N1K_OK ? &local : nullptr, (int)st, out, worst_status);
    return gs != N1K_OK ? gs : st;
}

n1k_status n1k_gather_groups(n1k_comm* c, n1k_handle* h, const n1k_result* local, n1k_result* out) {
    int worst = 0;
    n1k_status st = n1k_gather_groups_status(c, h, local, N1K_OK, out, &worst);
    if (st == N1K_OK && worst != N1K_OK) return fail(h, (n1k_status)worst, "a peer rank's step failed with status %d", worst);
    return st;
}

n1k_status n1k_gather_groups_status(n1k_comm* c, n1k_handle* h, const n1k_result* local, int local_status, n1k_result* out, int* worst_status) {
    return guarded(h, [&]() -> n1k_status {
        if (!c || !h || !out || !worst_status || (!local && local_status == N1K_OK)) return N1K_INVALID;
        *worst_status = local_status;
        static const n1k_result kNone{};
        if (!local || local_status != N1K_OK) local = &kNone;  // a rank whose step failed contributes no groups, only its status
        if (h->has_array_agg) return fail(h, N1K_UNSUPPORTED, "array_agg values are interned per rank: gather the rows on the host");
        n1k_status st = ensure_device(h);
        if (st != N1K_OK) return st;
        const size_t nk = h->plan.keys.size(), na = h->plan.aggs.size();
        const size_t rec = (nk + na) * sizeof(n1k_value);
        // ONE all-gather of fixed-size slots [count][records]: the slot size is part of the collective's shape, so it is the
        // same on every rank by construction — it starts at 1024 records and only ever changes on what ALL ranks read in
        // the gathered headers (a count beyond the slot: everybody doubles to fit the largest and gathers again)
        unsigned long long mine = local->ngroups;
        std::vector<char> stage;
        size_t slot = 0;
        for (;;) {
            const uint64_t capg = c->gather_cap;
            slot = 16 + (size_t)capg * rec;
            stage.assign(slot, 0);
            memcpy(stage.data(), &mine, 8);
            const unsigned long long my_status = (unsigned long long)(unsigned)local_status;
            memcpy(stage.data() + 8, &my_status, 8);
            for (uint64_t g = 0; g < std::min<uint64_t>(local->ngroups, capg); g++) {
                char* p = stage.data() + 16 + (size_t)g * rec;
                if (nk) memcpy(p, local->keys + g * nk, nk * sizeof(n1k_value));
                if (na) memcpy(p + nk * sizeof(n1k_value), local->aggs + g * na, na * sizeof(n1k_value));
            }
            HIP_TRY(h, c->gsend.ensure(slot));
            HIP_TRY(h, c->grecv.ensure(slot * (size_t)c->world));
            HIP_TRY(h, hipMemcpyAsync(c->gsend.p, stage.data(), slot, hipMemcpyHostToDevice, h->stream));
            st = all_gather_bytes(c, c->gsend.p, c->grecv.p, slot, h->stream);
            if (st != N1K_OK) return fail(h, st, "%s", c->last_error.c_str());
            c->ghost.resize(slot * (size_t)c->world);
            HIP_TRY(h, hipMemcpyAsync(c->ghost.data(), c->grecv.p, c->ghost.size(), hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            unsigned long long most = 0, bad = 0;
            for (int r = 0; r < c->world; r++) {
                unsigned long long n = 0, s = 0;
                memcpy(&n, c->ghost.data() + (size_t)r * slot, 8);
                memcpy(&s, c->ghost.data() + (size_t)r * slot + 8, 8);
                most = std::max(most, n);
                if (s && !bad) bad = s;  // (the lowest rank's failure: the same on every rank)
            }
            if (bad) {  // some rank's step failed: every rank learns it here, in the collective it would otherwise hang in
                *worst_status = (int)bad;
                memset(out, 0, sizeof *out);
                return N1K_OK;
            }
            if (most <= capg) break;
            while (c->gather_cap < most) c->gather_cap *= 2;
        }
        // 3. the union, in rank order; the plan's grouped tail (ORDER BY / OFFSET / LIMIT, projection) over it
        c->gkeys.clear();
        c->gaggs.clear();
        for (int r = 0; r < c->world; r++) {
            const char* base = c->ghost.data() + (size_t)r * slot;
            unsigned long long n = 0;
            memcpy(&n, base, 8);
            for (unsigned long long g = 0; g < n; g++) {
                const n1k_value* v = (const n1k_value*)(base + 16 + (size_t)g * rec);
                c->gkeys.insert(c->gkeys.end(), v, v + nk);
                c->gaggs.insert(c->gaggs.end(), v + nk, v + nk + na);
            }
        }
        const uint64_t total = nk ? c->gkeys.size() / nk : (na ? c->gaggs.size() / na : 0);
        return n1k_order_rows(h, total, c->gkeys.data(), c->gaggs.data(), out);
    });
}

n1k_status n1k_synth_columns(int device, void* stream, const n1k_synth_spec* spec, uint32_t* cat_codes, uint8_t* price_tags,
                             uint64_t* price_payload, uint8_t* user_tags, uint64_t* user_payload, uint8_t* region_tags,
                             uint64_t* region_payload) {
    return guarded(nullptr, [&]() -> n1k_status {
    if (!spec) return N1K_INVALID;
    if (hipSetDevice(device) != hipSuccess) return N1K_DEVICE_ERROR;
    SynthArgs a{};
    a.seed = spec->seed;
    a.first_row = spec->first_row;
    a.nrows = spec->nrows;
    a.total_rows = spec->total_rows;
    a.k_cat = spec->k_cat;
    a.cat_cdf = spec->zipf ? spec->cat_cdf : nullptr;
    a.cat_codes = cat_codes;
    a.price_tags = price_tags;
    a.price_payload = price_payload;
    a.user_tags = user_tags;
    a.user_payload = user_payload;
    a.region_tags = region_tags;
    a.region_payload = region_payload;
    if (launch_synth(a, (hipStream_t)stream) != hipSuccess) return N1K_DEVICE_ERROR;
    return N1K_OK;
    });
}

}  // extern "C"
