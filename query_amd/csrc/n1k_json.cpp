// n1k_json.cpp — see n1k_json.h.  A plain scanner: no DOM for the document, only the wanted fields are decoded;
// arrays and objects that ARE wanted values are re-serialised canonically (sorted names, compact: what
// objectValue.MarshalJSON emits, value/object.go:30-78) so that equal values get equal dictionary codes.
#include "n1k_json.h"
#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unordered_map>
#include "n1k_types.h"

namespace n1k {

bool parse_leaf_path(const std::string& text, JsonPath& out) {
    out.names.clear();
    std::vector<JsonStep> parts;
    size_t i = 0;
    const size_t n = text.size();
    while (i < n) {
        const char c = text[i];
        if (c == '(' || c == ')' || c == '.' || c == ' ') {
            i++;
            continue;
        }
        JsonStep st;
        if (c == '[') {  // [-?digits]
            size_t j = i + 1;
            if (j < n && text[j] == '-') j++;
            const size_t d0 = j;
            while (j < n && text[j] >= '0' && text[j] <= '9') j++;
            if (j == d0 || j - d0 > 18 || j >= n || text[j] != ']' || parts.empty()) return false;
            st.is_index = true;
            st.index = strtoll(text.c_str() + i + 1, nullptr, 10);
            i = j + 1;
            parts.push_back(st);
            continue;
        }
        if (c != '`') return false;
        i++;
        while (i < n && text[i] != '`') st.name.push_back(text[i++]);
        if (i >= n) return false;
        i++;
        parts.push_back(st);
    }
    if (parts.size() < 2 || parts[0].is_index) return false;  // alias alone names the whole document
    out.names.assign(parts.begin() + 1, parts.end());
    return true;
}

namespace {

constexpr size_t kMaxJsonDepth = 10000;  // Go's encoding/json scanner gives up at the same depth

struct Scanner {
    const char* p;
    const char* end;
    std::string* err;
    std::string open;  // skip(): brackets still open
    bool fail(const char* m) {
        if (err->empty()) *err = m;
        return false;
    }
    void ws() {
        while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++;
    }
    static void put_utf8(std::string& o, unsigned cp) {
        if (cp < 0x80) o.push_back((char)cp);
        else if (cp < 0x800) { o.push_back((char)(0xC0 | (cp >> 6))); o.push_back((char)(0x80 | (cp & 0x3F))); }
        else if (cp < 0x10000) { o.push_back((char)(0xE0 | (cp >> 12))); o.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); o.push_back((char)(0x80 | (cp & 0x3F))); }
        else { o.push_back((char)(0xF0 | (cp >> 18))); o.push_back((char)(0x80 | ((cp >> 12) & 0x3F))); o.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); o.push_back((char)(0x80 | (cp & 0x3F))); }
    }
    bool hex4(unsigned& v) {
        if (end - p < 4) return fail("short \\u escape");
        v = 0;
        for (int i = 0; i < 4; i++) {
            const char c = *p++;
            v <<= 4;
            if (c >= '0' && c <= '9') v |= (unsigned)(c - '0');
            else if (c >= 'a' && c <= 'f') v |= (unsigned)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= (unsigned)(c - 'A' + 10);
            else return fail("bad \\u escape");
        }
        return true;
    }
    // p at the opening quote; decodes into out (or only skips when out == nullptr)
    bool string(std::string* out) {
        if (p >= end || *p != '"') return fail("string expected");
        p++;
        for (;;) {
            const char* q = (const char*)memchr(p, '"', (size_t)(end - p));
            if (!q) return fail("unterminated string");
            const char* b = (const char*)memchr(p, '\\', (size_t)(q - p));
            if (!b) {
                if (out) out->append(p, q);
                p = q + 1;
                return true;
            }
            if (out) out->append(p, b);
            p = b + 1;
            if (p >= end) return fail("unterminated escape");
            const char c = *p++;
            unsigned cp = 0;
            switch (c) {
                case '"': cp = '"'; break;
                case '\\': cp = '\\'; break;
                case '/': cp = '/'; break;
                case 'b': cp = '\b'; break;
                case 'f': cp = '\f'; break;
                case 'n': cp = '\n'; break;
                case 'r': cp = '\r'; break;
                case 't': cp = '\t'; break;
                case 'u': {
                    if (!hex4(cp)) return false;
                    if (cp >= 0xD800 && cp < 0xDC00 && end - p >= 6 && p[0] == '\\' && p[1] == 'u') {
                        const char* save = p;
                        p += 2;
                        unsigned lo = 0;
                        if (!hex4(lo)) return false;
                        if (lo >= 0xDC00 && lo < 0xE000) cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                        else { p = save; cp = 0xFFFD; }
                    } else if (cp >= 0xD800 && cp < 0xE000)
                        cp = 0xFFFD;
                    break;
                }
                default: return fail("bad escape");
            }
            if (out) put_utf8(*out, cp);
        }
    }
    bool literal(const char* s, size_t n) {
        if ((size_t)(end - p) < n || memcmp(p, s, n)) return fail("bad literal");
        p += n;
        return true;
    }
    // number text [b, e)
    bool number(const char*& b, const char*& e) {
        b = p;
        if (p < end && *p == '-') p++;
        if (p >= end || *p < '0' || *p > '9') return fail("bad number");
        while (p < end && *p >= '0' && *p <= '9') p++;
        if (p < end && *p == '.') {
            p++;
            if (p >= end || *p < '0' || *p > '9') return fail("bad number");
            while (p < end && *p >= '0' && *p <= '9') p++;
        }
        if (p < end && (*p == 'e' || *p == 'E')) {
            p++;
            if (p < end && (*p == '+' || *p == '-')) p++;
            if (p >= end || *p < '0' || *p > '9') return fail("bad number");
            while (p < end && *p >= '0' && *p <= '9') p++;
        }
        e = p;
        return true;
    }
    // Any value, skipped without recursion: a stack of open brackets instead, bounded like Go's encoding/json scanner
    // (10000 levels; a document nested deeper than that is malformed for the reference too).
    bool member_name() {
        ws();
        if (!string(nullptr)) return false;
        ws();
        if (p >= end || *p != ':') return fail("':' expected");
        p++;
        return true;
    }
    bool skip() {
        open.clear();
        for (;;) {
            // a value starts here
            ws();
            if (p >= end) return fail("value expected");
            bool opened = false;
            switch (*p) {
                case '"':
                    if (!string(nullptr)) return false;
                    break;
                case '{':
                case '[': {
                    const char c = *p++;
                    ws();
                    if (p < end && *p == (c == '{' ? '}' : ']')) { p++; break; }
                    if (open.size() >= kMaxJsonDepth) return fail("nesting too deep");
                    open.push_back(c);
                    if (c == '{' && !member_name()) return false;
                    opened = true;
                    break;
                }
                case 't': if (!literal("true", 4)) return false; break;
                case 'f': if (!literal("false", 5)) return false; break;
                case 'n': if (!literal("null", 4)) return false; break;
                default: {
                    const char *b, *e;
                    if (!number(b, e)) return false;
                }
            }
            if (opened) continue;
            // a value ended: close what it completes, or move to the next member / element
            for (;;) {
                if (open.empty()) return true;
                ws();
                const char top = open.back();
                if (p < end && *p == ',') {
                    p++;
                    if (top == '{' && !member_name()) return false;
                    break;
                }
                if (p < end && *p == (top == '{' ? '}' : ']')) {
                    p++;
                    open.pop_back();
                    continue;
                }
                return fail(top == '{' ? "',' or '}' expected" : "',' or ']' expected");
            }
        }
    }
};

// value.NewValue's typing of a JSON number (value/value.go:375-382; go_json hands int64 for integer literals that
// fit, float64 otherwise; a float64 with no fraction that fits int64 folds to the int, integer.go:354-356)
void type_number(const char* b, const char* e, uint8_t& tag, uint64_t& payload) {
    bool integral = true;
    for (const char* q = b; q < e; q++)
        if (*q == '.' || *q == 'e' || *q == 'E') integral = false;
    char buf[64];
    std::string big;
    const size_t n = (size_t)(e - b);
    const char* z = buf;
    if (n < sizeof buf) {
        memcpy(buf, b, n);
        buf[n] = 0;
    } else {
        big.assign(b, e);
        z = big.c_str();
    }
    if (integral && n <= 20) {
        errno = 0;
        char* endp = nullptr;
        const long long v = strtoll(z, &endp, 10);
        if (errno == 0 && endp && *endp == 0) {
            tag = T_INT;
            payload = (uint64_t)v;
            return;
        }
    }
    const double d = strtod(z, nullptr);
    if (d >= -9223372036854775808.0 && d < 9223372036854775808.0 && d == (double)(int64_t)d) {
        tag = T_INT;
        payload = (uint64_t)(int64_t)d;
        return;
    }
    tag = T_FLOAT;
    memcpy(&payload, &d, 8);
}

}  // namespace

// strconv.FormatFloat(f, 'f', -1, 64) (value/float.go:31-48): shortest digits that round-trip, positional notation
void format_float(double f, std::string& o) {
    if (f != f) { o += "\"NaN\""; return; }
    if (std::isinf(f)) { o += f > 0 ? "\"+Infinity\"" : "\"-Infinity\""; return; }
    if (f == 0) { o += "0"; return; }
    char e[40];
    for (int prec = 0; prec < 17; prec++) {
        snprintf(e, sizeof e, "%.*e", prec, f);
        if (strtod(e, nullptr) == f) break;
    }
    std::string digits;
    const char* q = e;
    bool neg = false;
    if (*q == '-') { neg = true; q++; }
    for (; *q && *q != 'e'; q++)
        if (*q != '.') digits.push_back(*q);
    const int ex = atoi(q + 1);
    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
    if (neg) o.push_back('-');
    const int point = ex + 1;
    if (point <= 0) {
        o += "0.";
        o.append((size_t)(-point), '0');
        o += digits;
    } else if (point >= (int)digits.size()) {
        o += digits;
        o.append((size_t)point - digits.size(), '0');
    } else {
        o.append(digits, 0, (size_t)point);
        o.push_back('.');
        o.append(digits, (size_t)point, std::string::npos);
    }
}

void json_quote(const std::string& s, std::string& o) {
    o.push_back('"');
    for (unsigned char c : s) {
        switch (c) {
            case '"': o += "\\\""; break;
            case '\\': o += "\\\\"; break;
            case '\n': o += "\\n"; break;
            case '\r': o += "\\r"; break;
            case '\t': o += "\\t"; break;
            case '\b': o += "\\b"; break;
            case '\f': o += "\\f"; break;
            default:
                if (c < 0x20) {
                    char b[8];
                    snprintf(b, sizeof b, "\\u%04x", c);
                    o += b;
                } else
                    o.push_back((char)c);
        }
    }
    o.push_back('"');
}

namespace {
void quote(const std::string& s, std::string& o) { json_quote(s, o); }

// canonical text of the value at sc.p, appended to o
// (Recursive, one frame of three strings per level and every object level copies its children's text: bounded at
//  kMaxCanonDepth levels — a wanted VALUE nested deeper than that is N1K_UNSUPPORTED_DATA, not a stack overflow on a small
//  thread stack; documents are skipped without recursion up to Go's own limit, above.)
constexpr int kMaxCanonDepth = 256;
bool canon(Scanner& sc, std::string& o, int depth) {
    if (depth > kMaxCanonDepth) return sc.fail("a wanted array / object value nested deeper than 256 levels");
    sc.ws();
    if (sc.p >= sc.end) return sc.fail("value expected");
    switch (*sc.p) {
        case '"': {
            std::string s;
            if (!sc.string(&s)) return false;
            quote(s, o);
            return true;
        }
        case '{': {
            sc.p++;
            std::vector<std::pair<std::string, std::string>> fields;
            sc.ws();
            if (sc.p < sc.end && *sc.p == '}') { sc.p++; o += "{}"; return true; }
            for (;;) {
                sc.ws();
                std::string k, v;
                if (!sc.string(&k)) return false;
                sc.ws();
                if (sc.p >= sc.end || *sc.p != ':') return sc.fail("':' expected");
                sc.p++;
                if (!canon(sc, v, depth + 1)) return false;
                bool dup = false;
                for (auto& f : fields)
                    if (f.first == k) { f.second = v; dup = true; }  // a Go map keeps the last one
                if (!dup) fields.emplace_back(std::move(k), std::move(v));
                sc.ws();
                if (sc.p < sc.end && *sc.p == ',') { sc.p++; continue; }
                if (sc.p < sc.end && *sc.p == '}') { sc.p++; break; }
                return sc.fail("',' or '}' expected");
            }
            std::sort(fields.begin(), fields.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
            o.push_back('{');
            for (size_t i = 0; i < fields.size(); i++) {
                if (i) o.push_back(',');
                quote(fields[i].first, o);
                o.push_back(':');
                o += fields[i].second;
            }
            o.push_back('}');
            return true;
        }
        case '[': {
            sc.p++;
            o.push_back('[');
            sc.ws();
            if (sc.p < sc.end && *sc.p == ']') { sc.p++; o.push_back(']'); return true; }
            for (bool first = true;; first = false) {
                if (!first) o.push_back(',');
                if (!canon(sc, o, depth + 1)) return false;
                sc.ws();
                if (sc.p < sc.end && *sc.p == ',') { sc.p++; continue; }
                if (sc.p < sc.end && *sc.p == ']') { sc.p++; o.push_back(']'); return true; }
                return sc.fail("',' or ']' expected");
            }
        }
        case 't': o += "true"; return sc.literal("true", 4);
        case 'f': o += "false"; return sc.literal("false", 5);
        case 'n': o += "null"; return sc.literal("null", 4);
        default: {
            const char *b, *e;
            if (!sc.number(b, e)) return false;
            uint8_t tag;
            uint64_t pay;
            type_number(b, e, tag, pay);
            if (tag == T_INT) o += std::to_string((long long)pay);
            else {
                double d;
                memcpy(&d, &pay, 8);
                format_float(d, o);
            }
            return true;
        }
    }
}

struct Interner {
    std::unordered_map<std::string, uint64_t> index;
    std::vector<std::string>* strings;
    uint64_t code(const std::string& s) {
        auto it = index.find(s);
        if (it != index.end()) return it->second;
        const uint64_t c = strings->size();
        strings->push_back(s);
        index.emplace(s, c);
        return c;
    }
};

// sc.p at a value: decode it as a tagged scalar
bool decode(Scanner& sc, Interner& in, std::string& tmp, uint8_t& tag, uint64_t& payload) {
    sc.ws();
    if (sc.p >= sc.end) return sc.fail("value expected");
    payload = 0;
    switch (*sc.p) {
        case '"':
            tmp.clear();
            if (!sc.string(&tmp)) return false;
            tag = T_STRING;
            payload = in.code(tmp);
            return true;
        case '{':
        case '[':
            tag = *sc.p == '{' ? T_OBJECT : T_ARRAY;
            tmp.clear();
            if (!canon(sc, tmp, 0)) return false;
            payload = in.code(tmp);
            return true;
        case 't': tag = T_TRUE; return sc.literal("true", 4);
        case 'f': tag = T_FALSE; return sc.literal("false", 5);
        case 'n': tag = T_NULL; return sc.literal("null", 4);
        default: {
            const char *b, *e;
            if (!sc.number(b, e)) return false;
            type_number(b, e, tag, payload);
            return true;
        }
    }
}

bool object_level(Scanner& sc, const std::vector<JsonPath>& paths, std::vector<uint32_t>& want, uint32_t depth, Interner& in,
                  std::string& tmp, std::string& name, uint8_t* tags, uint64_t* pay);

// sc.p at the start of a value that the paths in `hit` reach after `depth` steps: the paths that end here take the
// value; the others go on — through its fields when it is an object, through its elements when it is an array
// (a field of a non-object and an element of a non-array are MISSING: value/parsed.go:159-163, :236-243).
bool value_level(Scanner& sc, const std::vector<JsonPath>& paths, const std::vector<uint32_t>& hit, uint32_t depth, Interner& in,
                 std::string& tmp, uint8_t* tags, uint64_t* pay) {
    std::vector<uint32_t> by_name, by_index;
    bool leaf = false;
    for (uint32_t h : hit) {
        if (paths[h].names.size() == depth) leaf = true;
        else if (paths[h].names[depth].is_index) by_index.push_back(h);
        else by_name.push_back(h);
    }
    sc.ws();
    const char* start = sc.p;
    const char* after = nullptr;
    if (leaf) {
        uint8_t t;
        uint64_t v;
        if (!decode(sc, in, tmp, t, v)) return false;
        for (uint32_t h : hit)
            if (paths[h].names.size() == depth) { tags[h] = t; pay[h] = v; }
        after = sc.p;
        if (by_name.empty() && by_index.empty()) return true;
        sc.p = start;
    }
    if (sc.p < sc.end && *sc.p == '{' && !by_name.empty()) {
        sc.p++;
        std::string nm;
        if (!object_level(sc, paths, by_name, depth, in, tmp, nm, tags, pay)) return false;
    } else if (sc.p < sc.end && *sc.p == '[' && !by_index.empty()) {
        // element starts, then the wanted elements (a negative index needs the length)
        sc.p++;
        std::vector<const char*> starts;
        sc.ws();
        if (sc.p < sc.end && *sc.p == ']') sc.p++;
        else
            for (;;) {
                sc.ws();
                starts.push_back(sc.p);
                if (!sc.skip()) return false;
                sc.ws();
                if (sc.p < sc.end && *sc.p == ',') { sc.p++; continue; }
                if (sc.p < sc.end && *sc.p == ']') { sc.p++; break; }
                return sc.fail("',' or ']' expected");
            }
        const char* end_of_array = sc.p;
        std::vector<uint32_t> same;
        while (!by_index.empty()) {
            const long long want_ix = paths[by_index[0]].names[depth].index;
            same.clear();
            for (size_t i = 0; i < by_index.size();) {
                if (paths[by_index[i]].names[depth].index == want_ix) {
                    same.push_back(by_index[i]);
                    by_index[i] = by_index.back();
                    by_index.pop_back();
                } else
                    i++;
            }
            long long ix = want_ix < 0 ? want_ix + (long long)starts.size() : want_ix;
            if (ix < 0 || ix >= (long long)starts.size()) continue;  // MISSING (value/array.go:204-214)
            sc.p = starts[(size_t)ix];
            if (!value_level(sc, paths, same, depth + 1, in, tmp, tags, pay)) return false;
        }
        sc.p = end_of_array;
    } else if (!leaf && !sc.skip())
        return false;
    if (after) sc.p = after;
    return true;
}

// One object level: sc.p just after '{'.  `want` lists the paths whose step `depth` is a field name looked up here;
// the FIRST field of that name counts (go_json.FirstFind, value/parsed.go:189-193).
bool object_level(Scanner& sc, const std::vector<JsonPath>& paths, std::vector<uint32_t>& want, uint32_t depth, Interner& in,
                  std::string& tmp, std::string& name, uint8_t* tags, uint64_t* pay) {
    sc.ws();
    if (sc.p < sc.end && *sc.p == '}') { sc.p++; return true; }
    for (;;) {
        sc.ws();
        name.clear();
        if (!sc.string(&name)) return false;
        sc.ws();
        if (sc.p >= sc.end || *sc.p != ':') return sc.fail("':' expected");
        sc.p++;
        // which wanted paths continue through this field?
        std::vector<uint32_t> hit;
        for (size_t i = 0; i < want.size();) {
            if (paths[want[i]].names[depth].name == name) {
                hit.push_back(want[i]);
                want[i] = want.back();
                want.pop_back();
            } else
                i++;
        }
        if (hit.empty()) {
            if (!sc.skip()) return false;
        } else if (!value_level(sc, paths, hit, depth + 1, in, tmp, tags, pay))
            return false;
        sc.ws();
        if (sc.p < sc.end && *sc.p == ',') { sc.p++; continue; }
        if (sc.p < sc.end && *sc.p == '}') { sc.p++; return true; }
        return sc.fail("',' or '}' expected");
    }
}

}  // namespace

long long extract_json_range(const std::vector<JsonPath>& paths, const uint64_t* offsets, const char* bytes, uint64_t first,
                             uint64_t last, JsonColumns& out, std::string& err) {
    const size_t np = paths.size();
    out.tags.assign(np, std::vector<uint8_t>((size_t)(last - first)));
    out.payload.assign(np, std::vector<uint64_t>((size_t)(last - first)));
    out.strings.clear();
    Interner in;
    in.strings = &out.strings;
    std::string tmp, name;
    std::vector<uint8_t> t(np);
    std::vector<uint64_t> v(np);
    std::vector<uint32_t> want;
    for (uint64_t d = first; d < last; d++) {
        Scanner sc{bytes + offsets[d], bytes + offsets[d + 1], &err};
        for (size_t i = 0; i < np; i++) { t[i] = T_MISSING; v[i] = 0; }
        want.clear();
        for (uint32_t i = 0; i < np; i++) want.push_back(i);
        // (a scalar document has no fields and no elements: every path is MISSING)
        if (!value_level(sc, paths, want, 0, in, tmp, t.data(), v.data())) return (long long)d;
        sc.ws();
        if (sc.p != sc.end) {
            err = "trailing bytes after the document";
            return (long long)d;
        }
        for (size_t i = 0; i < np; i++) {
            out.tags[i][(size_t)(d - first)] = t[i];
            out.payload[i][(size_t)(d - first)] = v[i];
        }
    }
    return -1;
}

}  // namespace n1k
