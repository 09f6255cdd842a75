// n1k_plan.cpp — plan JSON + expression.Stringer text parser for the device subset.
#include "n1k_plan.h"

#include <cmath>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <map>

namespace n1k {
namespace {

// ------------------------------------------------------------------ minimal JSON (plan documents only)

struct JVal {
    enum Type { Null, Bool, Num, Str, Arr, Obj } type = Null;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<JVal> arr;
    std::vector<std::pair<std::string, JVal>> obj;
    const JVal* get(const char* name) const {
        for (auto& kv : obj)
            if (kv.first == name) return &kv.second;
        return nullptr;
    }
};

struct JParser {
    const char* s;
    size_t n, i = 0;
    std::string err;
    void ws() {
        while (i < n && (s[i] == ' ' || s[i] == '\n' || s[i] == '\t' || s[i] == '\r')) i++;
    }
    bool fail(const std::string& m) {
        if (err.empty()) err = m + " at offset " + std::to_string(i);
        return false;
    }
    static void put_utf8(std::string& o, unsigned cp) {
        if (cp < 0x80) o += (char)cp;
        else if (cp < 0x800) {
            o += (char)(0xC0 | (cp >> 6));
            o += (char)(0x80 | (cp & 0x3F));
        } else if (cp < 0x10000) {
            o += (char)(0xE0 | (cp >> 12));
            o += (char)(0x80 | ((cp >> 6) & 0x3F));
            o += (char)(0x80 | (cp & 0x3F));
        } else {
            o += (char)(0xF0 | (cp >> 18));
            o += (char)(0x80 | ((cp >> 12) & 0x3F));
            o += (char)(0x80 | ((cp >> 6) & 0x3F));
            o += (char)(0x80 | (cp & 0x3F));
        }
    }
    bool str(std::string& out) {
        if (i >= n || s[i] != '"') return fail("expected string");
        i++;
        while (i < n && s[i] != '"') {
            char c = s[i++];
            if (c != '\\') {
                out += c;
                continue;
            }
            if (i >= n) return fail("bad escape");
            char e = s[i++];
            switch (e) {
                case 'n': out += '\n'; break;
                case 't': out += '\t'; break;
                case 'r': out += '\r'; break;
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                case 'u': {
                    if (i + 4 > n) return fail("bad \\u");
                    unsigned cp = (unsigned)strtoul(std::string(s + i, 4).c_str(), nullptr, 16);
                    i += 4;
                    if (cp >= 0xD800 && cp < 0xDC00 && i + 6 <= n && s[i] == '\\' && s[i + 1] == 'u') {
                        unsigned lo = (unsigned)strtoul(std::string(s + i + 2, 4).c_str(), nullptr, 16);
                        i += 6;
                        cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                    }
                    put_utf8(out, cp);
                    break;
                }
                default: out += e;
            }
        }
        if (i >= n) return fail("unterminated string");
        i++;
        return true;
    }
    bool value(JVal& v, int depth = 0) {
        if (depth > 64) return fail("too deep");
        ws();
        if (i >= n) return fail("unexpected end");
        char c = s[i];
        if (c == '{') {
            v.type = JVal::Obj;
            i++;
            ws();
            if (i < n && s[i] == '}') { i++; return true; }
            for (;;) {
                ws();
                std::string k;
                if (!str(k)) return false;
                ws();
                if (i >= n || s[i] != ':') return fail("expected :");
                i++;
                JVal c2;
                if (!value(c2, depth + 1)) return false;
                v.obj.emplace_back(std::move(k), std::move(c2));
                ws();
                if (i < n && s[i] == ',') { i++; continue; }
                if (i < n && s[i] == '}') { i++; return true; }
                return fail("expected , or }");
            }
        }
        if (c == '[') {
            v.type = JVal::Arr;
            i++;
            ws();
            if (i < n && s[i] == ']') { i++; return true; }
            for (;;) {
                JVal c2;
                if (!value(c2, depth + 1)) return false;
                v.arr.push_back(std::move(c2));
                ws();
                if (i < n && s[i] == ',') { i++; continue; }
                if (i < n && s[i] == ']') { i++; return true; }
                return fail("expected , or ]");
            }
        }
        if (c == '"') {
            v.type = JVal::Str;
            return str(v.str);
        }
        if (!strncmp(s + i, "true", 4)) { v.type = JVal::Bool; v.b = true; i += 4; return true; }
        if (!strncmp(s + i, "false", 5)) { v.type = JVal::Bool; v.b = false; i += 5; return true; }
        if (!strncmp(s + i, "null", 4)) { v.type = JVal::Null; i += 4; return true; }
        char* end = nullptr;
        v.num = strtod(s + i, &end);
        if (end == s + i) return fail("unexpected character");
        v.type = JVal::Num;
        i = (size_t)(end - s);
        return true;
    }
};

// ------------------------------------------------------------------ expression tokenizer / parser

enum class TK { End, LParen, RParen, LBrack, RBrack, Comma, Dot, Plus, Minus, Star, Slash, Percent, Eq, Lt, Le, Ident, Word, Str, Num, Other };

struct Tok {
    TK kind = TK::End;
    std::string text;   // Ident: name without backticks; Word: lower-cased; Str: decoded bytes; Num: literal
    size_t begin = 0, end = 0;
};

struct Lexer {
    const std::string& s;
    std::vector<Tok> toks;
    std::string err;
    explicit Lexer(const std::string& src) : s(src) {}
    static bool wordc(char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9') || c == '_'; }
    bool run() {
        size_t i = 0, n = s.size();
        while (i < n) {
            char c = s[i];
            if (c == ' ' || c == '\t' || c == '\n') { i++; continue; }
            Tok t;
            t.begin = i;
            auto single = [&](TK k) { t.kind = k; i++; };
            // a '-' directly followed by a digit is a negative literal only where a value may start
            bool value_pos = toks.empty() || !(toks.back().kind == TK::RParen || toks.back().kind == TK::RBrack ||
                                               toks.back().kind == TK::Ident || toks.back().kind == TK::Str ||
                                               toks.back().kind == TK::Num ||
                                               (toks.back().kind == TK::Word && (toks.back().text == "true" || toks.back().text == "false" ||
                                                                                  toks.back().text == "null" || toks.back().text == "missing")));
            if (c == '(') single(TK::LParen);
            else if (c == ')') single(TK::RParen);
            else if (c == '[') single(TK::LBrack);
            else if (c == ']') single(TK::RBrack);
            else if (c == ',') single(TK::Comma);
            else if (c == '.') single(TK::Dot);
            else if (c == '+') single(TK::Plus);
            else if (c == '*') single(TK::Star);
            else if (c == '/') single(TK::Slash);
            else if (c == '%') single(TK::Percent);
            else if (c == '=') { single(TK::Eq); if (i < n && s[i] == '=') i++; }
            else if (c == '<') {
                i++;
                if (i < n && s[i] == '=') { t.kind = TK::Le; i++; }
                else t.kind = TK::Lt;
            } else if (c == '`') {
                size_t j = s.find('`', i + 1);
                if (j == std::string::npos) { err = "unterminated identifier"; return false; }
                t.kind = TK::Ident;
                t.text = s.substr(i + 1, j - i - 1);
                i = j + 1;
                if (i < n && s[i] == 'i' && (i + 1 >= n || !wordc(s[i + 1]))) { err = "case-insensitive identifier"; return false; }
            } else if (c == '"') {
                JParser jp{s.c_str(), s.size(), i, ""};
                if (!jp.str(t.text)) { err = "bad string literal"; return false; }
                t.kind = TK::Str;
                i = jp.i;
            } else if ((c >= '0' && c <= '9') || (c == '-' && value_pos && i + 1 < n && s[i + 1] >= '0' && s[i + 1] <= '9')) {
                size_t j = i + 1;
                while (j < n) {
                    char d = s[j];
                    if ((d >= '0' && d <= '9') || d == '.' || d == 'e' || d == 'E') j++;
                    else if ((d == '+' || d == '-') && (s[j - 1] == 'e' || s[j - 1] == 'E')) j++;
                    else break;
                }
                t.kind = TK::Num;
                t.text = s.substr(i, j - i);
                i = j;
            } else if (c == '-') single(TK::Minus);
            else if (wordc(c)) {
                size_t j = i;
                while (j < n && wordc(s[j])) j++;
                t.kind = TK::Word;
                t.text = s.substr(i, j - i);
                for (auto& ch : t.text) if (ch >= 'A' && ch <= 'Z') ch = (char)(ch + 32);
                i = j;
            } else {
                t.kind = TK::Other;
                t.text = std::string(1, c);
                i++;
            }
            t.end = i;
            toks.push_back(std::move(t));
        }
        Tok e;
        e.kind = TK::End;
        e.begin = e.end = n;
        toks.push_back(e);
        return true;
    }
};

struct EParser {
    const std::string& src;
    Lexer lx;
    size_t p = 0;
    PlanError& err;
    EParser(const std::string& s, PlanError& e) : src(s), lx(s), err(e) {}

    const Tok& cur() const { return lx.toks[p]; }
    bool is_word(const char* w) const { return cur().kind == TK::Word && cur().text == w; }
    std::unique_ptr<Expr> unsupported(const std::string& what) {
        if (err.msg.empty()) {
            err.unsupported = true;
            err.msg = what + " is outside the device subset (in: " + src + ")";
        }
        return nullptr;
    }
    std::unique_ptr<Expr> bad(const std::string& what) {
        if (err.msg.empty()) {
            err.unsupported = false;
            err.msg = "cannot parse expression: " + what + " (in: " + src + ")";
        }
        return nullptr;
    }
    static std::unique_ptr<Expr> mk(EK k) {
        auto e = std::make_unique<Expr>();
        e->kind = k;
        return e;
    }

    // constants are value.MarshalJSON text (expression/stringer.go:386-394); integer literals that fit int64
    // stay int64, everything else goes through float64 + NewValue folding (value/value.go:375-382)
    std::unique_ptr<Expr> number(const std::string& lit) {
        auto e = mk(EK::Const);
        bool isint = lit.find_first_of(".eE") == std::string::npos;
        if (isint) {
            errno = 0;
            char* end = nullptr;
            long long v = strtoll(lit.c_str(), &end, 10);
            if (errno == 0 && end && *end == 0) {
                e->ctag = T_INT;
                e->cpayload = (uint64_t)v;
                return e;
            }
        }
        double d = strtod(lit.c_str(), nullptr);
        bool inrange = d >= -9223372036854775808.0 && d < 9223372036854775808.0;
        if (inrange && d == (double)(int64_t)d) {
            e->ctag = T_INT;
            e->cpayload = (uint64_t)(int64_t)d;
        } else {
            e->ctag = T_FLOAT;
            memcpy(&e->cpayload, &d, 8);
        }
        return e;
    }

    std::unique_ptr<Expr> primary() {
        const Tok& t = cur();
        switch (t.kind) {
            case TK::LParen: return paren();
            case TK::Ident: {
                auto e = mk(EK::Path);
                e->text = src.substr(t.begin, t.end - t.begin);
                p++;
                return e;
            }
            case TK::Str: {
                auto e = mk(EK::Const);
                e->ctag = T_STRING;
                e->cstr = t.text;
                p++;
                return e;
            }
            case TK::Num: {
                auto e = number(t.text);
                p++;
                return e;
            }
            case TK::Word: {
                std::string w = t.text;
                if (w == "true" || w == "false" || w == "null" || w == "missing") {
                    auto e = mk(EK::Const);
                    e->ctag = w == "true" ? T_TRUE : (w == "false" ? T_FALSE : (w == "null" ? T_NULL : T_MISSING));
                    p++;
                    return e;
                }
                if ((w == "idiv" || w == "imod") && lx.toks[p + 1].kind == TK::LParen) {
                    p += 2;
                    auto e = mk(w == "idiv" ? EK::IDiv : EK::IMod);
                    auto a = primary();
                    if (!a) return nullptr;
                    if (cur().kind != TK::Comma) return bad("expected , in " + w);
                    p++;
                    auto b = primary();
                    if (!b) return nullptr;
                    if (cur().kind != TK::RParen) return bad("expected ) in " + w);
                    p++;
                    e->ch.push_back(std::move(a));
                    e->ch.push_back(std::move(b));
                    return e;
                }
                if (w == "cover" && lx.toks[p + 1].kind == TK::LParen) {
                    // cover (expr): the value of a covered expression comes out of the index entry (expression/cover.go); for
                    // this path it is a leaf like any other — the caller evaluates it per row and hands over the column
                    const size_t b = t.begin;
                    p += 2;
                    auto inner = primary();
                    if (!inner) return nullptr;
                    if (cur().kind != TK::RParen) return bad("expected ) after cover");
                    const size_t e_end = cur().end;
                    p++;
                    auto e = mk(EK::Path);
                    e->text = src.substr(b, e_end - b);
                    return e;
                }
                if (w == "meta" && lx.toks[p + 1].kind == TK::LParen && lx.toks[p + 2].kind == TK::Ident &&
                    lx.toks[p + 3].kind == TK::RParen) {  // meta(`alias`): the document's meta data, a leaf root (expression/func_meta.go)
                    auto e = mk(EK::Path);
                    e->text = src.substr(t.begin, lx.toks[p + 3].end - t.begin);
                    p += 4;
                    return e;
                }
                if ((w == "round" || w == "trunc" || w == "abs" || w == "ceil" || w == "floor" || w == "sign" || w == "sqrt") &&
                    lx.toks[p + 1].kind == TK::LParen) {  // expression/func_num.go; stringer: name(arg, arg)
                    p += 2;
                    auto e = mk(EK::Func);
                    e->fname = w;
                    for (;;) {
                        auto a = primary();
                        if (!a) return nullptr;
                        e->ch.push_back(std::move(a));
                        if (cur().kind == TK::Comma) { p++; continue; }
                        break;
                    }
                    if (cur().kind != TK::RParen) return bad("expected ) in " + w);
                    p++;
                    const size_t maxargs = (w == "round" || w == "trunc") ? 2 : 1;
                    if (e->ch.empty() || e->ch.size() > maxargs) return bad(w + " takes 1" + (maxargs == 2 ? " or 2" : "") + " arguments");
                    return e;
                }
                if ((w == "greatest" || w == "least") && lx.toks[p + 1].kind == TK::LParen) {  // expression/func_comp.go
                    p += 2;
                    auto e = mk(EK::Func);
                    e->fname = w;
                    for (;;) {
                        auto a = primary();
                        if (!a) return nullptr;
                        e->ch.push_back(std::move(a));
                        if (cur().kind == TK::Comma) { p++; continue; }
                        break;
                    }
                    if (cur().kind != TK::RParen) return bad("expected ) in " + w);
                    p++;
                    if (e->ch.size() < 2) return bad(w + " takes at least 2 arguments");
                    return e;
                }
                return unsupported("function or keyword '" + w + "'");
            }
            case TK::LBrack: return unsupported("array constructor");
            case TK::Other: return unsupported("token '" + t.text + "'");
            default: return bad("unexpected token at offset " + std::to_string(t.begin));
        }
    }

    // everything expression.Stringer wraps in parentheses
    std::unique_ptr<Expr> paren() {
        size_t open = cur().begin;
        p++;  // (
        if (cur().kind == TK::Minus) {  // (-x)
            p++;
            auto o = primary();
            if (!o) return nullptr;
            if (cur().kind != TK::RParen) return bad("expected ) after negation");
            p++;
            auto e = mk(EK::Neg);
            e->ch.push_back(std::move(o));
            return e;
        }
        if (is_word("not")) {  // (not x)
            p++;
            auto o = primary();
            if (!o) return nullptr;
            if (cur().kind != TK::RParen) return bad("expected ) after not");
            p++;
            auto e = mk(EK::Not);
            e->ch.push_back(std::move(o));
            return e;
        }
        auto first = primary();
        if (!first) return nullptr;
        const Tok& t = cur();
        auto close = [&](std::unique_ptr<Expr> e) -> std::unique_ptr<Expr> {
            if (cur().kind != TK::RParen) return bad("expected ) at offset " + std::to_string(cur().begin));
            p++;
            return e;
        };
        switch (t.kind) {
            case TK::RParen: p++; return first;
            case TK::Dot: {  // (x.`name`)  nav_field
                p++;
                if (cur().kind != TK::Ident || first->kind != EK::Path) return unsupported("computed field access");
                p++;
                if (cur().kind != TK::RParen) return bad("expected ) after field");
                size_t close_end = cur().end;
                p++;
                auto e = mk(EK::Path);
                e->text = src.substr(open, close_end - open);
                return e;
            }
            case TK::LBrack: {  // (x[const])  nav_element with a constant index: still a host-extracted leaf
                p++;
                if (cur().kind != TK::Num || first->kind != EK::Path || lx.toks[p + 1].kind != TK::RBrack)
                    return unsupported("computed element access");
                p += 2;
                if (cur().kind != TK::RParen) return bad("expected ) after element");
                size_t close_end = cur().end;
                p++;
                auto e = mk(EK::Path);
                e->text = src.substr(open, close_end - open);
                return e;
            }
            case TK::Plus:
            case TK::Star: {
                TK op = t.kind;
                auto e = mk(op == TK::Plus ? EK::Add : EK::Mult);
                e->ch.push_back(std::move(first));
                while (cur().kind == op) {
                    p++;
                    auto o = primary();
                    if (!o) return nullptr;
                    e->ch.push_back(std::move(o));
                }
                return close(std::move(e));
            }
            case TK::Minus:
            case TK::Slash:
            case TK::Percent:
            case TK::Eq:
            case TK::Lt:
            case TK::Le: {
                EK k = t.kind == TK::Minus ? EK::Sub : t.kind == TK::Slash ? EK::Div : t.kind == TK::Percent ? EK::Mod
                       : t.kind == TK::Eq ? EK::Eq : t.kind == TK::Lt ? EK::LT : EK::LE;
                p++;
                auto o = primary();
                if (!o) return nullptr;
                auto e = mk(k);
                e->ch.push_back(std::move(first));
                e->ch.push_back(std::move(o));
                return close(std::move(e));
            }
            case TK::Word: {
                if (t.text == "and" || t.text == "or") {
                    std::string w = t.text;
                    auto e = mk(w == "and" ? EK::And : EK::Or);
                    e->ch.push_back(std::move(first));
                    while (is_word(w.c_str())) {
                        p++;
                        auto o = primary();
                        if (!o) return nullptr;
                        e->ch.push_back(std::move(o));
                    }
                    return close(std::move(e));
                }
                if (t.text == "is") {
                    p++;
                    bool neg = false;
                    if (is_word("not")) { neg = true; p++; }
                    EK k;
                    if (is_word("null")) k = neg ? EK::IsNotNull : EK::IsNull;
                    else if (is_word("missing")) k = neg ? EK::IsNotMissing : EK::IsMissing;
                    else if (is_word("valued")) k = neg ? EK::IsNotValued : EK::IsValued;
                    else return bad("bad IS predicate");
                    p++;
                    auto e = mk(k);
                    e->ch.push_back(std::move(first));
                    return close(std::move(e));
                }
                if (t.text == "between") {
                    p++;
                    auto lo = primary();
                    if (!lo) return nullptr;
                    if (!is_word("and")) return bad("expected AND in BETWEEN");
                    p++;
                    auto hi = primary();
                    if (!hi) return nullptr;
                    auto e = mk(EK::Between);
                    e->ch.push_back(std::move(first));
                    e->ch.push_back(std::move(lo));
                    e->ch.push_back(std::move(hi));
                    return close(std::move(e));
                }
                return unsupported("operator '" + t.text + "'");
            }
            default: return unsupported("operator at offset " + std::to_string(t.begin));
        }
    }

    std::unique_ptr<Expr> full() {
        if (!lx.run()) return bad(lx.err);
        auto e = primary();
        if (!e) return nullptr;
        if (cur().kind != TK::End) return bad("trailing text at offset " + std::to_string(cur().begin));
        return e;
    }
};

void collect_paths(const Expr* e, std::vector<std::string>& paths) {
    if (!e) return;
    if (e->kind == EK::Path) {
        for (auto& p : paths)
            if (p == e->text) return;
        paths.push_back(e->text);
        return;
    }
    for (auto& c : e->ch) collect_paths(c.get(), paths);
}

bool plan_node(const JVal& node, ParsedPlan& out, PlanError& err, int depth) {
    if (node.type != JVal::Obj || depth > 8) {
        err.msg = "plan node is not an object";
        return false;
    }
    const JVal* op = node.get("#operator");
    if (!op || op->type != JVal::Str) {
        err.msg = "plan node without #operator";
        return false;
    }
    const std::string& name = op->str;
    if (name == "Parallel") {  // plan/parallel.go:54-67
        const JVal* child = node.get("~child");
        if (!child) { err.msg = "Parallel without ~child"; return false; }
        if (const JVal* mp = node.get("maxParallelism")) out.max_parallelism = (int)mp->num;
        return plan_node(*child, out, err, depth + 1);
    }
    if (name == "Sequence") {  // plan/sequence.go:48-57
        const JVal* ch = node.get("~children");
        if (!ch || ch->type != JVal::Arr) { err.msg = "Sequence without ~children"; return false; }
        for (auto& c : ch->arr)
            if (!plan_node(c, out, err, depth + 1)) return false;
        return true;
    }
    if (name == "Filter" && out.has_group) {  // HAVING: the Filter that follows FinalGroup
        if (out.has_having || out.has_order || out.limit >= 0 || out.offset > 0) {
            err.unsupported = true;
            err.msg = "only one HAVING Filter, before Order / Offset / Limit, runs on the device";
            return false;
        }
        const JVal* c = node.get("condition");
        if (!c || c->type != JVal::Str) { err.msg = "Filter without condition"; return false; }
        out.having_text = c->str;
        out.has_having = true;
        return true;
    }
    if (name == "Filter") {  // plan/filter.go:46-53
        if (out.has_filter) {
            err.unsupported = true;
            err.msg = "only [Filter?, InitialGroup] sequences run on the device";
            return false;
        }
        const JVal* c = node.get("condition");
        if (!c || c->type != JVal::Str) { err.msg = "Filter without condition"; return false; }
        out.condition = parse_expression(c->str, err);
        if (!out.condition) return false;
        out.has_filter = true;
        return true;
    }
    if (name == "InitialGroup") {  // plan/group.go:54-70
        if (out.has_group) { err.unsupported = true; err.msg = "more than one InitialGroup"; return false; }
        const JVal* ks = node.get("group_keys");
        const JVal* as = node.get("aggregates");
        if (ks && ks->type == JVal::Arr)
            for (auto& k : ks->arr) {
                if (k.type != JVal::Str) { err.msg = "group key is not a string"; return false; }
                auto e = parse_expression(k.str, err);
                if (!e) return false;
                out.keys.push_back(std::move(e));
                out.key_texts.push_back(k.str);
            }
        if (as && as->type == JVal::Arr)
            for (auto& a : as->arr) {
                if (a.type != JVal::Str) { err.msg = "aggregate is not a string"; return false; }
                AggDef d;
                if (!parse_aggregate(a.str, d, err)) return false;
                out.aggs.push_back(std::move(d));
            }
        out.has_group = true;
        return true;
    }
    if (name == "IntermediateGroup" || name == "FinalGroup") {  // plan/group.go:106-273: subsumed by the device operator
        if (!out.has_group) { err.msg = name + " before InitialGroup"; return false; }
        const JVal* ks = node.get("group_keys");
        const JVal* as = node.get("aggregates");
        size_t nk = ks && ks->type == JVal::Arr ? ks->arr.size() : 0, na = as && as->type == JVal::Arr ? as->arr.size() : 0;
        bool same = nk == out.key_texts.size() && na == out.aggs.size();
        for (size_t i = 0; same && i < nk; i++) same = ks->arr[i].type == JVal::Str && ks->arr[i].str == out.key_texts[i];
        for (size_t i = 0; same && i < na; i++) same = as->arr[i].type == JVal::Str && as->arr[i].str == out.aggs[i].text;
        if (!same) { err.msg = name + " does not match the InitialGroup"; return false; }
        return true;
    }
    auto const_count = [&](const JVal* v, const char* what, int64_t& dst) -> bool {
        // LIMIT / OFFSET expressions: non-negative integer constants only (anything else needs the evaluator)
        if (!v || v->type != JVal::Str) { err.msg = std::string(what) + " without expression"; return false; }
        std::string t = v->str;
        while (!t.empty() && (t.front() == '(' || t.front() == ' ')) t.erase(t.begin());
        while (!t.empty() && (t.back() == ')' || t.back() == ' ')) t.pop_back();
        if (t.empty() || t.size() > 18 || t.find_first_not_of("0123456789") != std::string::npos) {
            err.unsupported = true;
            err.msg = std::string(what) + " expression is not an integer constant: " + v->str;
            return false;
        }
        dst = (int64_t)strtoll(t.c_str(), nullptr, 10);
        return true;
    };
    if (name == "Order") {  // plan/order.go:51-79 (the node carries its own offset / limit for the top-k sort)
        if (!out.has_group || out.has_order) { err.unsupported = true; err.msg = "Order runs on the device only over the groups"; return false; }
        const JVal* ts = node.get("sort_terms");
        if (!ts || ts->type != JVal::Arr || ts->arr.empty()) { err.msg = "Order without sort_terms"; return false; }
        for (auto& t : ts->arr) {
            const JVal* e = t.type == JVal::Obj ? t.get("expr") : nullptr;
            if (!e || e->type != JVal::Str) { err.msg = "sort term without expr"; return false; }
            OrderTerm ot;
            ot.text = e->str;
            const JVal* d = t.get("desc");
            ot.desc = d && d->type == JVal::Bool && d->b;
            for (size_t i = 0; i < out.key_texts.size() && ot.key_index < 0; i++)
                if (out.key_texts[i] == ot.text) ot.key_index = (int)i;
            for (size_t i = 0; i < out.aggs.size() && ot.key_index < 0 && ot.agg_index < 0; i++)
                if (out.aggs[i].text == ot.text) ot.agg_index = (int)i;
            // after an InitialProject the sort term may name a projection alias (the projected item carries the alias as
            // a field, execution/project_initial.go:118-121) or repeat a term's expression
            for (size_t i = 0; i < out.project.size() && ot.key_index < 0 && ot.agg_index < 0 && ot.proj_index < 0; i++)
                if ((!out.project[i].as.empty() && ot.text == "`" + out.project[i].as + "`") || out.project[i].text == ot.text)
                    ot.proj_index = (int)i;
            if (ot.key_index < 0 && ot.agg_index < 0 && ot.proj_index < 0) {
                err.unsupported = true;
                err.msg = "sort term is neither a group key, an aggregate nor a projection term of the plan: " + ot.text;
                return false;
            }
            out.order.push_back(ot);
        }
        out.has_order = true;
        if (node.get("offset") && !const_count(node.get("offset"), "offset", out.offset)) return false;
        if (node.get("limit") && !const_count(node.get("limit"), "limit", out.limit)) return false;
        return true;
    }
    if (name == "InitialProject") {  // plan/project.go:73-110: result_terms [{expr, as?, star?}], raw?, distinct?
        if (!out.has_group || out.has_project || out.has_order || out.limit >= 0 || out.offset > 0) {
            err.unsupported = true;
            err.msg = "InitialProject runs on the device only over the final groups, before Order / Offset / Limit";
            return false;
        }
        for (const char* flag : {"raw", "distinct"})
            if (const JVal* f = node.get(flag))
                if (f->type == JVal::Bool && f->b) {
                    err.unsupported = true;
                    err.msg = std::string("InitialProject with ") + flag + " does not run on the device";
                    return false;
                }
        const JVal* ts = node.get("result_terms");
        if (!ts || ts->type != JVal::Arr) { err.msg = "InitialProject without result_terms"; return false; }
        for (auto& t : ts->arr) {
            if (t.type != JVal::Obj) { err.msg = "result term is not an object"; return false; }
            const JVal* star = t.get("star");
            if (star && star->type == JVal::Bool && star->b) {
                err.unsupported = true;
                err.msg = "a star projection does not run on the device";
                return false;
            }
            const JVal* e = t.get("expr");
            if (!e || e->type != JVal::Str) { err.msg = "result term without expr"; return false; }
            ProjectTerm pt;
            pt.text = e->str;
            const JVal* as = t.get("as");
            if (as && as->type == JVal::Str) pt.as = as->str;
            out.project.push_back(std::move(pt));
        }
        out.has_project = true;
        return true;
    }
    if (name == "FinalProject") {  // plan/project.go:171-181: no data; the projection attachment becomes the row
        if (!out.has_project) { err.unsupported = true; err.msg = "FinalProject without an InitialProject over the groups"; return false; }
        return true;
    }
    if (name == "Limit" || name == "Offset") {  // plan/limit.go:46-53, plan/offset.go
        if (!out.has_group) { err.unsupported = true; err.msg = name + " runs on the device only over the groups"; return false; }
        return const_count(node.get("expr"), name == "Limit" ? "limit" : "offset", name == "Limit" ? out.limit : out.offset);
    }
    err.unsupported = true;
    err.msg = "operator " + name + " does not run on the device";
    return false;
}

}  // namespace

// value.Collate over parsed JSON (value/array.go arrayCollate: element by element, then the shorter first;
// value/object.go:511-556 objectCollate: fewer fields first, then name by name over the sorted union of the names — a
// name the other lacks makes this one larger — then the values; scalars by type order, value/value.go:69-79)
static int jval_collate(const JVal& a, const JVal& b) {
    auto cls = [](const JVal& v) { return v.type == JVal::Null ? 1 : v.type == JVal::Bool ? 2 : v.type == JVal::Num ? 3 : v.type == JVal::Str ? 4 : v.type == JVal::Arr ? 5 : 6; };
    const int ca = cls(a), cb = cls(b);
    if (ca != cb) return ca < cb ? -1 : 1;
    switch (a.type) {
        case JVal::Null: return 0;
        case JVal::Bool: return (int)a.b - (int)b.b;
        case JVal::Num: return a.num < b.num ? -1 : (a.num > b.num ? 1 : 0);
        case JVal::Str: {
            const int c = a.str.compare(b.str);  // bytewise, like Go strings (value/string.go:116-130)
            return c < 0 ? -1 : (c > 0 ? 1 : 0);
        }
        case JVal::Arr:
            for (size_t i = 0; i < a.arr.size(); i++) {
                if (i >= b.arr.size()) return 1;
                const int c = jval_collate(a.arr[i], b.arr[i]);
                if (c) return c;
            }
            return a.arr.size() < b.arr.size() ? -1 : 0;
        default: {
            if (a.obj.size() != b.obj.size()) return a.obj.size() < b.obj.size() ? -1 : 1;
            std::vector<std::string> names;
            for (auto& kv : a.obj) names.push_back(kv.first);
            for (auto& kv : b.obj) names.push_back(kv.first);
            std::sort(names.begin(), names.end());
            names.erase(std::unique(names.begin(), names.end()), names.end());
            for (auto& n : names) {
                const JVal* x = a.get(n.c_str());
                const JVal* y = b.get(n.c_str());
                if (!x) return 1;
                if (!y) return -1;
                const int c = jval_collate(*x, *y);
                if (c) return c;
            }
            return 0;
        }
    }
}

bool json_text_collate(const std::string& a, const std::string& b, int& out) {
    JParser pa{a.c_str(), a.size(), 0, ""}, pb{b.c_str(), b.size(), 0, ""};
    JVal va, vb;
    if (!pa.value(va) || !pb.value(vb)) return false;
    out = jval_collate(va, vb);
    return true;
}

std::unique_ptr<Expr> parse_expression(const std::string& s, PlanError& err) {
    EParser ep(s, err);
    return ep.full();
}

// name([distinct ]operand | *)  — expression/stringer.go:581-604, registry algebra/agg_registry.go:24-62
bool parse_aggregate(const std::string& s, AggDef& out, PlanError& err) {
    size_t lp = s.find('(');
    size_t rp = s.rfind(')');
    if (lp == std::string::npos || rp == std::string::npos || rp < lp) {
        err.msg = "cannot parse aggregate: " + s;
        return false;
    }
    std::string name = s.substr(0, lp);
    for (auto& c : name) if (c >= 'A' && c <= 'Z') c = (char)(c + 32);
    while (!name.empty() && name.back() == ' ') name.pop_back();
    if (name == "sum") out.kind = AGG_SUM;
    else if (name == "count") out.kind = AGG_COUNT;
    else if (name == "countn") out.kind = AGG_COUNTN;
    else if (name == "avg") out.kind = AGG_AVG;
    else if (name == "min") out.kind = AGG_MIN;
    else if (name == "max") out.kind = AGG_MAX;
    else if (name == "array_agg") out.kind = AGG_ARRAY;  // algebra/agg_array.go, agg_array_distinct.go
    else {
        err.unsupported = true;
        err.msg = "aggregate " + name + " is outside the device subset";
        return false;
    }
    std::string inner = s.substr(lp + 1, rp - lp - 1);
    size_t b = inner.find_first_not_of(' ');
    inner = b == std::string::npos ? "" : inner.substr(b);
    out.distinct = false;
    if (inner.size() > 9 && strncasecmp(inner.c_str(), "distinct ", 9) == 0) {
        out.distinct = true;
        inner = inner.substr(9);
    }
    out.text = s;
    if (inner == "*") {
        if (out.kind != AGG_COUNT || out.distinct) { err.msg = "only count(*) takes *"; return false; }
        out.operand.reset();
        return true;
    }
    out.operand = parse_expression(inner, err);
    if (!out.operand) return false;
    if (out.distinct && (out.kind == AGG_MIN || out.kind == AGG_MAX)) {
        err.msg = "min/max(distinct) do not exist (algebra/agg_registry.go:41-47)";
        return false;
    }
    return true;
}

bool parse_plan_json(const char* json, size_t len, ParsedPlan& out, PlanError& err) {
    JParser jp{json, len, 0, ""};
    JVal root;
    if (!jp.value(root)) {
        err.msg = "plan JSON: " + jp.err;
        return false;
    }
    if (!plan_node(root, out, err, 0)) return false;
    if (!out.has_filter && !out.has_group) {
        err.unsupported = true;
        err.msg = "plan holds neither Filter nor InitialGroup";
        return false;
    }
    collect_paths(out.condition.get(), out.paths);
    for (auto& k : out.keys) collect_paths(k.get(), out.paths);
    for (auto& a : out.aggs) collect_paths(a.operand.get(), out.paths);
    return true;
}

}  // namespace n1k
