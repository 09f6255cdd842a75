// n1k_engine.cpp — host engine behind the C ABI (include/n1k.h).
//
// Host-side mirror of the reference operators for this path: one Handle plays the role of one
// Parallel copy of Sequence[Filter, InitialGroup] plus the serial IntermediateGroup / FinalGroup
// (execution/parallel.go:52-83, filter.go, group_initial.go, group_intermediate.go, group_final.go),
// with the consumer life cycle of execution/base.go:492-545:
//     create = beforeItems, push_batch = processItem*, finish = afterItems, reset = reopen,
//     stop = SendStop, destroy = Done.
// There is no CPU fallback: without a HIP device every compute call fails with N1K_DEVICE_ERROR.
#include "n1k_engine.h"

using namespace n1k;
using namespace n1k_eng;

namespace n1k_eng {

thread_local std::string g_create_error;

n1k_status fail(n1k_handle* h, n1k_status st, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    h->last_error = buf;
    return st;
}

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
             : f == "floor" ? AR_FLOOR : f == "sign" ? AR_SIGN : f == "sqrt" ? AR_SQRT : f == "greatest" ? AR_GREATEST : AR_LEAST;
        n1k_handle::Derived d{};
        d.op = op;
        if (op >= AR_GREATEST) h->need_rank = true;  // (strings collate by their bytewise rank)
        for (auto& c : e->ch) {
            Operand x;
            if (!to_operand(h, c.get(), x, err)) return false;
            if (d.nops == 4) {
                // GREATEST / LEAST of more than four arguments: the winner so far is the first argument of the next node
                // (exact: arguments at or below NULL are skipped, ties keep the earlier one)
                if (h->plan.paths.size() + h->derived.size() >= (size_t)kMaxCols) {
                    err.unsupported = true;
                    err.msg = "too many columns (inputs + arithmetic nodes > 16)";
                    return false;
                }
                Operand acc{};
                acc.col = (uint32_t)(h->plan.paths.size() + h->derived.size());
                h->derived.push_back(d);
                d = n1k_handle::Derived{};
                d.op = op;
                d.ops[d.nops++] = acc;
            }
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

n1k_status ensure_pinned_counters(n1k_handle* h) {
    if (!h->pin_counters) HIP_TRY(h, hipHostMalloc((void**)&h->pin_counters, (kCounters + kPinScratch) * sizeof(unsigned long long), hipHostMallocDefault));
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

}  // namespace n1k_eng

// ================================================================== C ABI


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
    if (h->host_us[5] > 0)
        fprintf(stderr, "n1k host trace: %.0f one-call executions; per call: reset %.1f us, push %.1f us, finish %.1f us (of which waiting %.1f us, after the wait %.1f us)\n",
                h->host_us[5], h->host_us[0] / h->host_us[5], h->host_us[1] / h->host_us[5], h->host_us[2] / h->host_us[5],
                h->host_us[3] / h->host_us[5], h->host_us[4] / h->host_us[5]);
    if (h->having) n1k_destroy(h->having);
    h->having = nullptr;
    if (h->project) n1k_destroy(h->project);
    h->project = nullptr;
    if (h->device_ready) {
        (void)hipSetDevice(h->device);
        if (h->stream) (void)hipStreamSynchronize(h->stream);
        drain_events(h);
        for (auto e : h->event_pool) (void)hipEventDestroy(e);
        if (h->ev_q0) (void)hipEventDestroy(h->ev_q0);
        if (h->ev_q1) (void)hipEventDestroy(h->ev_q1);
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
        if (h->pin_rows) (void)hipHostFree(h->pin_rows);
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
        h->jd_bytes.release();
        h->jd_offsets.release();
        h->jd_new_first.release();
        h->jd_patch_docs.release();
        h->jd_patch_pay.release();
        h->jd_tab.release();
        h->jd_status.release();
        h->jd_patch_tags.release();
        h->jd_new_list.release();
        h->jd_code_of.release();
        h->jd_codes.release();
        for (auto& b : h->jd_tags) b.release();
        for (auto& b : h->jd_payload) b.release();
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
    h->failure_global = false;
    h->tail_in_merge = false;
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
        if (!h->ev_q0) {
            (void)hipEventCreate(&h->ev_q0);
            (void)hipEventCreate(&h->ev_q1);
        }
        h->q1_recorded = false;
        h->q0_recorded = h->ev_q0 && hipEventRecord(h->ev_q0, h->stream) == hipSuccess;  // the query starts here (stats.query_ms)
        if (h->device_clean) return N1K_OK;  // the last query's final kernel left the device as the launches below would
        h->device_clean = true;
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
    else if (n == "topk_sample") h->opt_topk_sample = value ? 1 : 0;
    else if (n == "filter_stream") h->opt_filter_stream = value ? 1 : 0;
    else if (n == "fused_tail") h->opt_fused_tail = value ? 1 : 0;
    else if (n == "distinct_fill_pct") h->opt_distinct_fill_pct = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 1), 75);
    else if (n == "dedupe_unroll") h->opt_dedupe_unroll = (uint32_t)value;
    else if (n == "tail_in_merge") h->opt_tail_in_merge = value ? 1 : 0;
    else if (n == "agg_spec") h->opt_agg_spec = value ? 1 : 0;
    else if (n == "merge_chunks") h->opt_merge_chunks = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 0), 16);
    else if (n == "inject_failure") h->opt_inject_failure = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 0), 3);
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
    } else if (n == "json_device") {
        h->opt_json_device = value ? 1 : 0;
    } else if (n == "json_device_left_pct") {
        h->opt_json_device_left_pct = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 0), 100);
    } else if (n == "json_device_min_docs") {
        h->opt_json_device_min_docs = (uint64_t)std::max<int64_t>(value, 0);
    } else if (n == "json_threads") {
        h->opt_json_threads = (uint32_t)std::min<int64_t>(std::max<int64_t>(value, 0), 64);
    } else if (n == "partition_min_rows") {
        h->opt_partition_min_rows = (uint64_t)std::max<int64_t>(value, 1);
    } else if (n == "partition_probe_rows") {
        h->opt_partition_probe_rows = (uint64_t)std::max<int64_t>(value, 1);
    } else if (n == "partition_min_groups") {
        h->opt_partition_min_groups = (uint64_t)std::max<int64_t>(value, 1);
    } else if (n == "partition_sticky") {
        h->opt_partition_sticky = value ? 1 : 0;
        h->sticky.valid = false;
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
    static const bool trace = getenv("N1K_HOST_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = trace ? now() : 0;
    n1k_status st = n1k_reset(h);
    const double t1 = trace ? now() : 0;
    h->one_call = true;  // (the batch is the whole query: the scan's merge may run the tail, n1k_scan.cpp)
    h->tail_in_merge = false;
    if (st == N1K_OK) st = n1k_push_device_batch(h, batch);
    h->one_call = false;
    const double t2 = trace ? now() : 0;
    if (st == N1K_OK) {
        h->clear_on_finish = true;  // the result leaves the device; the state behind it is the next execution's reset
        st = n1k_finish(h, out);
        h->clear_on_finish = false;
    }
    if (trace) {
        h->host_us[0] += t1 - t0;
        h->host_us[1] += t2 - t1;
        h->host_us[2] += now() - t2;  // (finish in all; its wait is accounted inside)
        h->host_us[5] += 1;
    }
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
        if (bad[t] >= 0) {
            if (errs[t].find("nested deeper than 256") != std::string::npos)  // (well formed, but beyond what the extractor re-serialises)
                return fail(h, N1K_UNSUPPORTED_DATA, "document %lld: %s", bad[t], errs[t].c_str());
            return fail(h, N1K_INVALID, "document %lld is not valid JSON: %s", bad[t], errs[t].c_str());
        }
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
    if (!h || (ndocs && (!offsets || !bytes))) return N1K_INVALID;
    if (h->stop_flag.load()) return fail(h, N1K_STOPPED, "operator was stopped");
    // large batches: the device extractor (the bytes cross PCIe as they are; n1k_jsonpush.cpp) — it takes the batch or
    // leaves all of it to the host path below
    if (h->json_paths_state == 0) {
        h->json_paths.resize(h->plan.paths.size());
        h->json_paths_state = 1;
        for (size_t c = 0; c < h->plan.paths.size(); c++)
            if (!parse_leaf_path(h->plan.paths[c], h->json_paths[c])) h->json_paths_state = -1;
    }
    if (h->json_paths_state > 0 && ndocs) {
        bool done = false;
        n1k_status dst = push_json_device(h, ndocs, offsets, bytes, &done);
        if (dst != N1K_OK || done) return dst;
    }
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
    if (h->plan.has_group && !h->device_clean) {  // (clean: the query was finished and its counters zeroed behind the result)
        h->stats.rows_selected = counters[0];
        h->stats.groups_out = h->pending.count ? h->pending.count : counters[1];  // groups so far (in the table, or in the kept region)
    }
    if (!h->device_clean) h->stats.wide_key_values = counters[13];
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
    if (m->q0_recorded && m->q1_recorded && hipEventQuery(m->ev_q1) == hipSuccess) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, m->ev_q0, m->ev_q1) == hipSuccess) m->stats.query_ms = ms;
    }
    *out = h->stats;
    return N1K_OK;
    });
}

n1k_status n1k_synth_documents(uint64_t nrows, uint64_t first_id, const uint32_t* cat_codes, const uint8_t* price_tags,
                               const uint64_t* price_payload, const uint64_t* user_payload, const uint64_t* region_payload, uint32_t pad,
                               char* bytes, size_t cap, uint64_t* offsets, size_t* used) {
    return guarded(nullptr, [&]() -> n1k_status {
    if (!cat_codes || !price_tags || !price_payload || !user_payload || !region_payload || !offsets || !used || (cap && !bytes)) return N1K_INVALID;
    const uint32_t nthreads = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::min(32u, std::max(1u, std::thread::hardware_concurrency())), nrows / 65536 + 1));
    std::vector<std::string> part(nthreads);
    std::vector<std::vector<uint32_t>> lens(nthreads);
    auto work = [&](uint32_t t) {
        const uint64_t lo = nrows * t / nthreads, hi = nrows * (t + 1) / nthreads;
        std::string& o = part[t];
        o.reserve((size_t)(hi - lo) * (110 + pad));
        lens[t].reserve((size_t)(hi - lo));
        const std::string padding(pad, 'x');
        char num[32];
        for (uint64_t i = lo; i < hi; i++) {
            const size_t at = o.size();
            o += "{\"id\":\"d";
            o += std::to_string(first_id + i);
            o += "\"";
            if (cat_codes[i] == 0xFFFFFFFEu) o += ",\"cat\":null";
            else if (cat_codes[i] != 0xFFFFFFFFu) {
                o += ",\"cat\":\"cat_";
                o += std::to_string(cat_codes[i]);
                o += "\"";
            }
            switch (price_tags[i]) {
                case N1K_T_MISSING: break;
                case N1K_T_NULL: o += ",\"price\":null"; break;
                case N1K_T_INT: snprintf(num, sizeof num, "%lld", (long long)price_payload[i]); o += ",\"price\":"; o += num; break;
                case N1K_T_FLOAT: {
                    double d;
                    memcpy(&d, &price_payload[i], 8);
                    o += ",\"price\":";
                    format_float(d, o);
                    break;
                }
                default: o += ",\"price\":\"n/a\""; break;
            }
            snprintf(num, sizeof num, "%lld", (long long)user_payload[i]);
            o += ",\"user_id\":";
            o += num;
            snprintf(num, sizeof num, "%lld", (long long)region_payload[i]);
            o += ",\"region_id\":";
            o += num;
            o += ",\"pad\":\"";
            o += padding;
            o += "\"}";
            lens[t].push_back((uint32_t)(o.size() - at));
        }
    };
    std::vector<std::thread> th;
    for (uint32_t t = 1; t < nthreads; t++) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    size_t total = 0;
    for (auto& p : part) total += p.size();
    *used = total;
    if (total > cap) return N1K_OOM;
    size_t at = 0;
    uint64_t row = 0;
    for (uint32_t t = 0; t < nthreads; t++) {
        memcpy(bytes + at, part[t].data(), part[t].size());
        size_t o = at;
        for (uint32_t l : lens[t]) {
            offsets[row++] = o;
            o += l;
        }
        at += part[t].size();
    }
    offsets[nrows] = total;
    return N1K_OK;
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