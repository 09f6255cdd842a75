// n1k_tail.cpp — what follows FinalGroup over the final groups: HAVING, InitialProject / FinalProject, ORDER BY / OFFSET /
// LIMIT, ARRAY_AGG assembly (§8 f1, f2, f4).
#include "n1k_engine.h"

using namespace n1k;
using namespace n1k_eng;

namespace n1k_eng {

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
n1k_status build_projection(n1k_handle* h) {
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
    for (const auto& d : f->derived)
        if (d.op >= AR_GREATEST) {  // (the inner operator sees this handle's dictionary codes, not its own: no string ranks there)
            g_create_error = "GREATEST / LEAST as a projection term over the groups does not run on the device";
            return N1K_UNSUPPORTED;
        }
    f->prog.ncols = (uint32_t)(f->plan.paths.size() + f->derived.size());
    return N1K_OK;
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
n1k_status having_groups(n1k_handle* h, uint64_t& ng) {
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
n1k_status project_groups(n1k_handle* h, uint64_t ng) {
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
n1k_status order_groups(n1k_handle* h, uint64_t& ng) {
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
n1k_status array_agg_groups(n1k_handle* h, uint64_t ng, const unsigned long long* counters) {
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

}  // namespace n1k_eng

extern "C" {

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

}  // extern "C"
