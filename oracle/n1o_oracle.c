/*
 * n1o_oracle.c — CPU oracle: plain-C restatement of the reference's
 * Filter -> InitialGroup -> IntermediateGroup -> FinalGroup path.
 *
 * TEST INFRASTRUCTURE ONLY (see n1o.h): never linked into, imported by or
 * called from the product.  Pinned against the reference's own golden case
 * files by tests/test_oracle_golden.py.
 *
 * Structure mirrors the reference (all paths relative to the reference tree):
 *   val, collate/compare/equals ........ value/{integer,float,string,boolean,null,missing}.go
 *   num_add/sub/mult/neg/idiv/imod ..... value/integer.go:266-352, value/float.go:331-385
 *   new_value_f64 ...................... value/value.go:377-382, value/integer.go:354-356
 *   eval() ............................. expression/{comp_*,arith_*,logic_*,nav_field,identifier,constant}.go
 *   parse_*() .......................... inverse of expression/stringer.go
 *   group_key() ........................ execution/group_util.go:18-35 + value/object.go:30-78
 *   agg_* .............................. algebra/agg_{sum,count,countn,avg,min,max,*_distinct,util}.go
 *   set_* .............................. value/set.go:22-215
 *   run_worker/intermediate/final ...... execution/{parallel,filter,group_initial,group_intermediate,group_final}.go
 */
#define _GNU_SOURCE
#include "n1o.h"
#include <dirent.h>
#include <math.h>
#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------ values */

/* value/value.go:69-79 */
enum { TY_MISSING = 0, TY_NULL, TY_BOOLEAN, TY_NUMBER, TY_STRING, TY_ARRAY, TY_OBJECT, TY_JSON, TY_BINARY };

typedef struct val {
    uint8_t type;
    uint8_t isf;   /* NUMBER: 0 = intValue, 1 = floatValue */
    uint8_t b;     /* BOOLEAN */
    uint32_t code; /* STRING/ARRAY/OBJECT from a column: dictionary code, else UINT32_MAX */
    int64_t i;
    double f;
    const char *s; /* STRING/ARRAY/OBJECT bytes */
    uint32_t slen;
} val;

static const val V_MISSING = {TY_MISSING, 0, 0, UINT32_MAX, 0, 0.0, NULL, 0};
static const val V_NULL = {TY_NULL, 0, 0, UINT32_MAX, 0, 0.0, NULL, 0};

static inline val v_int(int64_t i) {
    val v = V_NULL;
    v.type = TY_NUMBER;
    v.isf = 0;
    v.i = i;
    return v;
}
static inline val v_float(double f) {
    val v = V_NULL;
    v.type = TY_NUMBER;
    v.isf = 1;
    v.f = f;
    return v;
}
static inline val v_bool(int b) {
    val v = V_NULL;
    v.type = TY_BOOLEAN;
    v.b = b ? 1 : 0;
    return v;
}

/* Go's float64 -> int64 conversion on amd64 (cvttsd2si): out of range / NaN -> MinInt64 */
static inline int64_t go_f2i(double f) {
    if (!(f >= -9223372036854775808.0 && f < 9223372036854775808.0)) return INT64_MIN;
    return (int64_t)f;
}
/* value/integer.go:354-356 */
static inline int is_int(double x) { return x == (double)go_f2i(x); }
/* value/value.go:377-382: NewValue(float64) */
static inline val new_value_f64(double d) { return is_int(d) ? v_int(go_f2i(d)) : v_float(d); }
/* intValue.Actual() / floatValue.Actual(): value/integer.go:57-59 */
static inline double num_actual(val v) { return v.isf ? v.f : (double)v.i; }

/* value/float.go:123-172 */
static int collate_float(double t, double o) {
    if (isnan(t)) return isnan(o) ? 0 : -1;
    if (isnan(o)) return 1;
    if (isinf(t) && t < 0) return (isinf(o) && o < 0) ? 0 : -1;
    if (isinf(o) && o < 0) return 1;
    if (isinf(t) && t > 0) return (isinf(o) && o > 0) ? 0 : 1;
    if (isinf(o) && o > 0) return -1;
    double r = t - o;
    return r < 0.0 ? -1 : (r > 0.0 ? 1 : 0);
}

static int bytes_cmp(const char *a, uint32_t la, const char *b, uint32_t lb) {
    uint32_t m = la < lb ? la : lb;
    int c = m ? memcmp(a, b, m) : 0;
    if (c) return c < 0 ? -1 : 1;
    return la < lb ? -1 : (la > lb ? 1 : 0);
}

/* X.Collate(other) for every type: value/integer.go:100-118, float.go:106-121,
 * string.go:116-130, boolean.go:99-113, null.go:90-92, missing.go:104-106.
 * *unsupported is set when two ARRAY/OBJECT values would have to be ordered. */
static int collate(val a, val b, int *unsupported) {
    if (a.type != b.type) return (int)a.type - (int)b.type;
    switch (a.type) {
    case TY_NUMBER:
        if (!a.isf && !b.isf) return a.i < b.i ? -1 : (a.i > b.i ? 1 : 0);
        if (a.isf) return collate_float(a.f, num_actual(b));
        return -collate_float(b.f, (double)a.i); /* intValue.Collate(floatValue) = -other.Collate(this) */
    case TY_STRING:
        return bytes_cmp(a.s, a.slen, b.s, b.slen);
    case TY_BOOLEAN:
        return a.b == b.b ? 0 : (!a.b ? -1 : 1);
    case TY_ARRAY:
    case TY_OBJECT:
        if (a.slen == b.slen && memcmp(a.s, b.s, a.slen) == 0) return 0;
        if (unsupported) *unsupported = 1;
        return 0;
    default:
        return 0;
    }
}

/* X.Compare(other): MISSING if either MISSING, else NULL if either NULL, else intValue(Collate) */
static val compare(val a, val b, int *unsupported) {
    if (a.type == TY_MISSING || b.type == TY_MISSING) return V_MISSING;
    if (a.type == TY_NULL || b.type == TY_NULL) return V_NULL;
    return v_int(collate(a, b, unsupported));
}

/* X.Equals(other): value/integer.go:68-87, float.go:74-93, string.go:82-96, boolean.go:74-88,
 * null.go:72-80, missing.go:88-90 */
static val equals(val a, val b, int *unsupported) {
    if (a.type == TY_MISSING || b.type == TY_MISSING) return V_MISSING;
    if (a.type == TY_NULL || b.type == TY_NULL) return V_NULL;
    if (a.type != b.type) return v_bool(0);
    switch (a.type) {
    case TY_NUMBER:
        if (!a.isf && !b.isf) return v_bool(a.i == b.i);
        return v_bool(num_actual(a) == num_actual(b));
    case TY_STRING:
        return v_bool(a.slen == b.slen && memcmp(a.s, b.s, a.slen) == 0);
    case TY_BOOLEAN:
        return v_bool(a.b == b.b);
    default:
        if (a.slen == b.slen && memcmp(a.s, b.s, a.slen) == 0) return v_bool(1);
        if (unsupported) *unsupported = 1; /* element-wise equality of arrays/objects not restated */
        return v_bool(0);
    }
}

/* X.Truth(): value/integer.go:136-138, float.go:190-192, string.go:148, boolean.go:125, null.go:106, missing.go:113 */
static int truth(val v, int *unsupported) {
    switch (v.type) {
    case TY_BOOLEAN: return v.b;
    case TY_NUMBER: return v.isf ? (!isnan(v.f) && v.f != 0.0) : (v.i != 0);
    case TY_STRING: return v.slen > 0;
    case TY_ARRAY:
    case TY_OBJECT:
        /* len > 0: "[]" / "{}" are the only empty canonical texts */
        return v.slen > 2;
    default: return 0;
    }
    (void)unsupported;
}

/* ---- NumberValue methods ---- */

/* value/integer.go:266-277, value/float.go:331-333 */
static val num_add(val a, val b) {
    if (!a.isf && !b.isf) {
        int64_t rv = (int64_t)((uint64_t)a.i + (uint64_t)b.i);
        if ((a.i >= 0 && b.i >= 0 && rv >= 0) || (a.i < 0 && b.i < 0 && rv < 0)) return v_int(rv);
        /* NOTE: mixed signs fall through to float in the reference as written (the
         * condition only accepts same-sign operands) — restated literally. */
        return v_float((double)a.i + (double)b.i);
    }
    return v_float(num_actual(a) + num_actual(b));
}
/* value/integer.go:331-335 */
static val num_neg(val a) {
    if (a.isf) return v_float(-a.f);
    if (a.i == INT64_MIN) return v_float(-(double)a.i);
    return v_int(-a.i);
}
/* value/integer.go:337-346, value/float.go:377-379 */
static val num_sub(val a, val b) {
    if (!a.isf && !b.isf && b.i > INT64_MIN) return num_add(a, v_int(-b.i));
    return v_float(num_actual(a) - num_actual(b));
}
/* value/integer.go:318-329, value/float.go:369-371 */
static val num_mult(val a, val b) {
    if (!a.isf && !b.isf) {
        int64_t rv = (int64_t)((uint64_t)a.i * (uint64_t)b.i);
        if (a.i == 0) return v_int(rv);
        /* rv/this == n ; guard the one trapping division */
        if (!(a.i == -1 && rv == INT64_MIN) && rv / a.i == b.i) return v_int(rv);
        if (a.i == -1 && rv == INT64_MIN && b.i == INT64_MIN) return v_int(rv); /* Go: MinInt64/-1 == MinInt64 */
        return v_float((double)a.i * (double)b.i);
    }
    return v_float(num_actual(a) * num_actual(b));
}
static inline int64_t go_idiv(int64_t a, int64_t b) { return (a == INT64_MIN && b == -1) ? INT64_MIN : a / b; }
static inline int64_t go_imod(int64_t a, int64_t b) { return (a == INT64_MIN && b == -1) ? 0 : a % b; }
/* value/integer.go:279-297, value/float.go:335-351 */
static val num_idiv(val a, val b) {
    if (!b.isf) {
        if (b.i == 0) return V_NULL;
        return v_int(go_idiv(a.isf ? go_f2i(a.f) : a.i, b.i));
    }
    if (b.f == 0.0) return V_NULL;
    int64_t d = go_f2i(b.f);
    if (d == 0) return V_NULL; /* Go would panic on integer divide by zero; treated as NULL here */
    return v_int(go_idiv(a.isf ? go_f2i(a.f) : a.i, d));
}
/* value/integer.go:299-316, value/float.go:353-367 */
static val num_imod(val a, val b) {
    if (!b.isf) {
        if (b.i == 0) return V_NULL;
        return v_int(go_imod(a.isf ? go_f2i(a.f) : a.i, b.i));
    }
    if (b.f == 0.0) return V_NULL;
    int64_t d = go_f2i(b.f);
    if (d == 0) return V_NULL;
    return v_int(go_imod(a.isf ? go_f2i(a.f) : a.i, d));
}

/* --------------------------------------------------------------- AST/parse */

enum {
    E_CONST, E_PATH, E_ADD, E_SUB, E_MULT, E_DIV, E_MOD, E_NEG, E_IDIV, E_IMOD, E_EQ, E_LT, E_LE, E_BETWEEN,
    E_AND, E_OR, E_NOT, E_ISNULL, E_ISNOTNULL, E_ISMISSING, E_ISNOTMISSING, E_ISVALUED, E_ISNOTVALUED,
    E_ROUND, E_TRUNC, E_ABS, E_CEIL, E_FLOOR, E_SIGN, E_SQRT, /* expression/func_num.go */
    E_GREATEST, E_LEAST /* expression/func_comp.go:54-67, 124-138 */
};

typedef struct node {
    int kind;
    int nch;
    struct node **ch;
    val cval;   /* E_CONST */
    char *text; /* E_PATH: stringer text; E_CONST string bytes owner */
    int col;    /* E_PATH: bound column */
} node;

typedef struct parser {
    const char *s;
    size_t pos, len;
    char err[256];
} parser;

static void perr(parser *p, const char *fmt, ...) {
    if (p->err[0]) return;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(p->err, sizeof p->err, fmt, ap);
    va_end(ap);
}
static node *mknode(int kind) {
    node *n = calloc(1, sizeof *n);
    n->kind = kind;
    n->col = -1;
    return n;
}
static void addch(node *n, node *c) {
    n->ch = realloc(n->ch, sizeof(node *) * (size_t)(n->nch + 1));
    n->ch[n->nch++] = c;
}
static void free_node(node *n) {
    if (!n) return;
    for (int i = 0; i < n->nch; i++) free_node(n->ch[i]);
    free(n->ch);
    free(n->text);
    free(n);
}
static void skipws(parser *p) {
    while (p->pos < p->len && (p->s[p->pos] == ' ' || p->s[p->pos] == '\t' || p->s[p->pos] == '\n')) p->pos++;
}
static int peekc(parser *p) { return p->pos < p->len ? (unsigned char)p->s[p->pos] : -1; }
static int is_wordc(int c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9') || c == '_'; }
/* does a keyword start here (followed by a non-word char)? */
static int at_word(parser *p, const char *w) {
    size_t n = strlen(w);
    if (p->pos + n > p->len) return 0;
    if (strncasecmp(p->s + p->pos, w, n) != 0) return 0;
    int c = p->pos + n < p->len ? (unsigned char)p->s[p->pos + n] : -1;
    return !is_wordc(c);
}
static int eat_word(parser *p, const char *w) {
    skipws(p);
    if (at_word(p, w)) {
        p->pos += strlen(w);
        return 1;
    }
    return 0;
}
static int eat_char(parser *p, char c) {
    skipws(p);
    if (peekc(p) == c) {
        p->pos++;
        return 1;
    }
    return 0;
}

static node *parse_expr(parser *p);

/* `name` -> returns malloc'd "`name`" text */
static char *parse_backtick(parser *p) {
    size_t st = p->pos;
    p->pos++;
    while (p->pos < p->len && p->s[p->pos] != '`') p->pos++;
    if (p->pos >= p->len) {
        perr(p, "unterminated identifier");
        return NULL;
    }
    p->pos++;
    if (peekc(p) == 'i' && !is_wordc(p->pos + 1 < p->len ? (unsigned char)p->s[p->pos + 1] : -1)) {
        perr(p, "case-insensitive identifiers are not supported");
        return NULL;
    }
    return strndup(p->s + st, p->pos - st);
}

/* JSON string literal -> raw bytes (constants are value.MarshalJSON text, expression/stringer.go:386-394) */
static node *parse_string(parser *p) {
    p->pos++;
    char *buf = malloc(p->len - p->pos + 1);
    size_t n = 0;
    while (p->pos < p->len && p->s[p->pos] != '"') {
        char c = p->s[p->pos++];
        if (c == '\\' && p->pos < p->len) {
            char e = p->s[p->pos++];
            switch (e) {
            case 'n': buf[n++] = '\n'; break;
            case 't': buf[n++] = '\t'; break;
            case 'r': buf[n++] = '\r'; break;
            case 'b': buf[n++] = '\b'; break;
            case 'f': buf[n++] = '\f'; break;
            case 'u': {
                unsigned cp = 0;
                for (int k = 0; k < 4 && p->pos < p->len; k++) {
                    char h = p->s[p->pos++];
                    cp = cp * 16 + (unsigned)(h <= '9' ? h - '0' : (h | 32) - 'a' + 10);
                }
                if (cp < 0x80) buf[n++] = (char)cp;
                else if (cp < 0x800) {
                    buf[n++] = (char)(0xC0 | (cp >> 6));
                    buf[n++] = (char)(0x80 | (cp & 0x3F));
                } else {
                    buf[n++] = (char)(0xE0 | (cp >> 12));
                    buf[n++] = (char)(0x80 | ((cp >> 6) & 0x3F));
                    buf[n++] = (char)(0x80 | (cp & 0x3F));
                }
                break;
            }
            default: buf[n++] = e;
            }
        } else
            buf[n++] = c;
    }
    if (p->pos >= p->len) {
        free(buf);
        perr(p, "unterminated string");
        return NULL;
    }
    p->pos++;
    buf[n] = 0;
    node *nd = mknode(E_CONST);
    nd->text = buf;
    nd->cval = V_NULL;
    nd->cval.type = TY_STRING;
    nd->cval.s = buf;
    nd->cval.slen = (uint32_t)n;
    return nd;
}

/* JSON number literal: int64 when it is an integer literal that fits (go_json), else float64 then
 * NewValue folding (value/value.go:375-382) */
static node *parse_number(parser *p) {
    size_t st = p->pos;
    int isint = 1;
    if (peekc(p) == '-') p->pos++;
    while (p->pos < p->len) {
        char c = p->s[p->pos];
        if (c >= '0' && c <= '9') p->pos++;
        else if (c == '.' || c == 'e' || c == 'E' || c == '+' || c == '-') {
            if ((c == '+' || c == '-') && !(p->s[p->pos - 1] == 'e' || p->s[p->pos - 1] == 'E')) break;
            isint = 0;
            p->pos++;
        } else
            break;
    }
    char tmp[64];
    size_t n = p->pos - st;
    if (n == 0 || n >= sizeof tmp) {
        perr(p, "bad number");
        return NULL;
    }
    memcpy(tmp, p->s + st, n);
    tmp[n] = 0;
    node *nd = mknode(E_CONST);
    if (isint) {
        char *end;
        long long ll = strtoll(tmp, &end, 10);
        /* overflow -> falls back to float64 */
        char chk[64];
        snprintf(chk, sizeof chk, "%lld", ll);
        if (strcmp(chk, tmp) == 0 || (tmp[0] == '-' && tmp[1] == '0' && tmp[2] == 0)) {
            nd->cval = v_int(ll);
            return nd;
        }
    }
    nd->cval = new_value_f64(strtod(tmp, NULL));
    return nd;
}

static node *mk_path(char *text) {
    node *n = mknode(E_PATH);
    n->text = text;
    return n;
}

/* name(args) : only idiv / imod are scalar functions in the device subset */
static node *parse_function(parser *p) {
    size_t st = p->pos;
    while (p->pos < p->len && is_wordc((unsigned char)p->s[p->pos])) p->pos++;
    char name[32];
    size_t n = p->pos - st;
    if (n == 0 || n >= sizeof name) {
        perr(p, "unexpected token at %zu", st);
        return NULL;
    }
    memcpy(name, p->s + st, n);
    name[n] = 0;
    for (size_t k = 0; k < n; k++)
        if (name[k] >= 'A' && name[k] <= 'Z') name[k] += 32;
    if (!strcmp(name, "true")) { node *c = mknode(E_CONST); c->cval = v_bool(1); return c; }
    if (!strcmp(name, "false")) { node *c = mknode(E_CONST); c->cval = v_bool(0); return c; }
    if (!strcmp(name, "null")) { node *c = mknode(E_CONST); c->cval = V_NULL; return c; }
    if (!strcmp(name, "missing")) { node *c = mknode(E_CONST); c->cval = V_MISSING; return c; }
    int kind = -1;
    if (!strcmp(name, "idiv")) kind = E_IDIV;
    else if (!strcmp(name, "imod")) kind = E_IMOD;
    else if (!strcmp(name, "round") || !strcmp(name, "trunc") || !strcmp(name, "abs") || !strcmp(name, "ceil") ||
             !strcmp(name, "floor") || !strcmp(name, "sign") || !strcmp(name, "sqrt")) {
        /* numeric functions of one argument; ROUND / TRUNC take an optional digit count */
        kind = !strcmp(name, "round") ? E_ROUND : !strcmp(name, "trunc") ? E_TRUNC : !strcmp(name, "abs") ? E_ABS
               : !strcmp(name, "ceil") ? E_CEIL : !strcmp(name, "floor") ? E_FLOOR : !strcmp(name, "sign") ? E_SIGN : E_SQRT;
        if (!eat_char(p, '(')) { perr(p, "expected ( after %s", name); return NULL; }
        node *nd = mknode(kind);
        for (;;) {
            node *a = parse_expr(p);
            if (!a) { free_node(nd); return NULL; }
            addch(nd, a);
            if (!eat_char(p, ',')) break;
        }
        if (!eat_char(p, ')')) { perr(p, "expected )"); free_node(nd); return NULL; }
        if (nd->nch > ((kind == E_ROUND || kind == E_TRUNC) ? 2 : 1)) { perr(p, "too many arguments to %s", name); free_node(nd); return NULL; }
        return nd;
    } else if (!strcmp(name, "greatest") || !strcmp(name, "least")) {
        /* GREATEST / LEAST: two or more arguments (func_comp.go:73-79) */
        kind = !strcmp(name, "greatest") ? E_GREATEST : E_LEAST;
        if (!eat_char(p, '(')) { perr(p, "expected ( after %s", name); return NULL; }
        node *nd = mknode(kind);
        for (;;) {
            node *a = parse_expr(p);
            if (!a) { free_node(nd); return NULL; }
            addch(nd, a);
            if (!eat_char(p, ',')) break;
        }
        if (!eat_char(p, ')')) { perr(p, "expected )"); free_node(nd); return NULL; }
        if (nd->nch < 2) { perr(p, "%s takes at least 2 arguments", name); free_node(nd); return NULL; }
        return nd;
    } else {
        perr(p, "function %s is outside the oracle subset", name);
        return NULL;
    }
    if (!eat_char(p, '(')) { perr(p, "expected ( after %s", name); return NULL; }
    node *nd = mknode(kind);
    node *a = parse_expr(p);
    if (!a) { free_node(nd); return NULL; }
    addch(nd, a);
    if (!eat_char(p, ',')) { perr(p, "expected ,"); free_node(nd); return NULL; }
    node *b = parse_expr(p);
    if (!b) { free_node(nd); return NULL; }
    addch(nd, b);
    if (!eat_char(p, ')')) { perr(p, "expected )"); free_node(nd); return NULL; }
    return nd;
}

static node *parse_paren(parser *p) {
    size_t open_pos = p->pos;
    p->pos++; /* '(' */
    skipws(p);
    /* (-x)  expression/stringer.go:95-101 */
    if (peekc(p) == '-' && !(p->pos + 1 < p->len && p->s[p->pos + 1] >= '0' && p->s[p->pos + 1] <= '9')) {
        /* "(-5 + x)" starts with a negative literal, "(-x)" / "(-(..))" is a negation */
        p->pos++;
        node *o = parse_expr(p);
        if (!o) return NULL;
        node *n = mknode(E_NEG);
        addch(n, o);
        if (!eat_char(p, ')')) { perr(p, "expected ) after neg"); free_node(n); return NULL; }
        return n;
    }
    /* (not x)  stringer.go:485-491 */
    if (at_word(p, "not")) {
        p->pos += 3;
        node *o = parse_expr(p);
        if (!o) return NULL;
        node *n = mknode(E_NOT);
        addch(n, o);
        if (!eat_char(p, ')')) { perr(p, "expected ) after not"); free_node(n); return NULL; }
        return n;
    }
    node *first = parse_expr(p);
    if (!first) return NULL;
    skipws(p);
    int c = peekc(p);
    node *res = NULL;
    if (c == ')') {
        p->pos++;
        return first;
    }
    if (c == '.') { /* (first.`name`)  stringer.go:521-544 */
        p->pos++;
        if (peekc(p) != '`' || first->kind != E_PATH) {
            perr(p, "computed field access is outside the oracle subset");
            free_node(first);
            return NULL;
        }
        char *nm = parse_backtick(p);
        if (!nm) { free_node(first); return NULL; }
        free(nm);
        if (!eat_char(p, ')')) { perr(p, "expected ) after field"); free_node(first); return NULL; }
        /* path text is the exact stringer text of this Field node */
        char *text = strndup(p->s + open_pos, p->pos - open_pos);
        free_node(first);
        return mk_path(text);
    }
    if (c == '[') { /* (first[const])  stringer.go:511-519 — constant index only: still a leaf the host extracts */
        p->pos++;
        node *ix = parse_expr(p);
        if (!ix || ix->kind != E_CONST || ix->cval.type != TY_NUMBER || ix->cval.isf || first->kind != E_PATH) {
            perr(p, "computed element access is outside the oracle subset");
            free_node(ix);
            free_node(first);
            return NULL;
        }
        free_node(ix);
        if (!eat_char(p, ']') || !eat_char(p, ')')) { perr(p, "expected ]) after element"); free_node(first); return NULL; }
        char *text = strndup(p->s + open_pos, p->pos - open_pos);
        free_node(first);
        return mk_path(text);
    }
    /* n-ary chains */
    const char *nary_w = NULL;
    int nary_kind = -1;
    if (c == '+') { nary_kind = E_ADD; }
    else if (c == '*') { nary_kind = E_MULT; }
    else if (at_word(p, "and")) { nary_kind = E_AND; nary_w = "and"; }
    else if (at_word(p, "or")) { nary_kind = E_OR; nary_w = "or"; }
    if (nary_kind >= 0) {
        res = mknode(nary_kind);
        addch(res, first);
        for (;;) {
            skipws(p);
            if (nary_w) {
                if (!at_word(p, nary_w)) break;
                p->pos += strlen(nary_w);
            } else {
                if (peekc(p) != (nary_kind == E_ADD ? '+' : '*')) break;
                p->pos++;
            }
            node *o = parse_expr(p);
            if (!o) { free_node(res); return NULL; }
            addch(res, o);
        }
        if (!eat_char(p, ')')) { perr(p, "expected ) after n-ary"); free_node(res); return NULL; }
        return res;
    }
    int bkind = -1;
    if (c == '-') { bkind = E_SUB; p->pos++; }
    else if (c == '/') { bkind = E_DIV; p->pos++; }
    else if (c == '%') { bkind = E_MOD; p->pos++; }
    else if (c == '=') { bkind = E_EQ; p->pos++; }
    else if (c == '<') {
        p->pos++;
        if (peekc(p) == '=') { p->pos++; bkind = E_LE; } else bkind = E_LT;
    }
    if (bkind >= 0) {
        node *o = parse_expr(p);
        if (!o) { free_node(first); return NULL; }
        res = mknode(bkind);
        addch(res, first);
        addch(res, o);
        if (!eat_char(p, ')')) { perr(p, "expected ) after binary"); free_node(res); return NULL; }
        return res;
    }
    if (at_word(p, "is")) { /* stringer.go:320-366 */
        p->pos += 2;
        int neg = eat_word(p, "not");
        int kind = -1;
        if (eat_word(p, "null")) kind = neg ? E_ISNOTNULL : E_ISNULL;
        else if (eat_word(p, "missing")) kind = neg ? E_ISNOTMISSING : E_ISMISSING;
        else if (eat_word(p, "valued")) kind = neg ? E_ISNOTVALUED : E_ISVALUED;
        if (kind < 0) { perr(p, "bad IS predicate"); free_node(first); return NULL; }
        res = mknode(kind);
        addch(res, first);
        if (!eat_char(p, ')')) { perr(p, "expected ) after is"); free_node(res); return NULL; }
        return res;
    }
    if (at_word(p, "between")) { /* stringer.go:268-278 */
        p->pos += 7;
        node *lo = parse_expr(p);
        if (!lo) { free_node(first); return NULL; }
        if (!eat_word(p, "and")) { perr(p, "expected and in between"); free_node(first); free_node(lo); return NULL; }
        node *hi = parse_expr(p);
        if (!hi) { free_node(first); free_node(lo); return NULL; }
        res = mknode(E_BETWEEN);
        addch(res, first);
        addch(res, lo);
        addch(res, hi);
        if (!eat_char(p, ')')) { perr(p, "expected ) after between"); free_node(res); return NULL; }
        return res;
    }
    perr(p, "operator at offset %zu is outside the oracle subset", p->pos);
    free_node(first);
    return NULL;
}

static node *parse_expr(parser *p) {
    skipws(p);
    int c = peekc(p);
    if (c < 0) { perr(p, "unexpected end"); return NULL; }
    if (c == '(') return parse_paren(p);
    if (c == '`') {
        char *t = parse_backtick(p);
        return t ? mk_path(t) : NULL;
    }
    if (c == '"') return parse_string(p);
    if ((c >= '0' && c <= '9') || c == '-') return parse_number(p);
    if (c == '[' || c == '{') { perr(p, "array/object constants are outside the oracle subset"); return NULL; }
    return parse_function(p);
}

static node *parse_full(const char *s, char *err, size_t errlen) {
    parser p;
    memset(&p, 0, sizeof p);
    p.s = s;
    p.len = strlen(s);
    node *n = parse_expr(&p);
    if (n) {
        skipws(&p);
        if (p.pos != p.len) {
            perr(&p, "trailing text at %zu", p.pos);
            free_node(n);
            n = NULL;
        }
    }
    if (!n && err) snprintf(err, errlen, "parse '%s': %s", s, p.err);
    return n;
}

/* aggregates: name([distinct ]operand|*)  stringer.go:581-604 */
enum { A_SUM, A_COUNT, A_COUNTN, A_AVG, A_MIN, A_MAX, A_ARRAY };
typedef struct aggdef {
    int kind;
    int distinct;
    node *operand; /* NULL for count(*) */
} aggdef;

static int parse_aggregate(const char *s, aggdef *a, char *err, size_t errlen) {
    parser p;
    memset(&p, 0, sizeof p);
    p.s = s;
    p.len = strlen(s);
    skipws(&p);
    size_t st = p.pos;
    while (p.pos < p.len && is_wordc((unsigned char)p.s[p.pos])) p.pos++;
    char name[16];
    size_t n = p.pos - st;
    if (n == 0 || n >= sizeof name) goto bad;
    memcpy(name, p.s + st, n);
    name[n] = 0;
    for (size_t k = 0; k < n; k++)
        if (name[k] >= 'A' && name[k] <= 'Z') name[k] += 32;
    /* algebra/agg_registry.go:24-62 */
    if (!strcmp(name, "sum")) a->kind = A_SUM;
    else if (!strcmp(name, "count")) a->kind = A_COUNT;
    else if (!strcmp(name, "countn")) a->kind = A_COUNTN;
    else if (!strcmp(name, "avg")) a->kind = A_AVG;
    else if (!strcmp(name, "min")) a->kind = A_MIN;
    else if (!strcmp(name, "max")) a->kind = A_MAX;
    else if (!strcmp(name, "array_agg")) a->kind = A_ARRAY;
    else goto bad;
    if (!eat_char(&p, '(')) goto bad;
    a->distinct = eat_word(&p, "distinct");
    a->operand = NULL;
    skipws(&p);
    if (peekc(&p) == '*') {
        p.pos++;
        if (a->kind != A_COUNT || a->distinct) goto bad;
    } else {
        a->operand = parse_expr(&p);
        if (!a->operand) {
            snprintf(err, errlen, "aggregate '%s': %s", s, p.err);
            return -1;
        }
    }
    if (!eat_char(&p, ')')) goto bad;
    skipws(&p);
    if (p.pos != p.len) goto bad;
    if (a->distinct && (a->kind == A_MIN || a->kind == A_MAX)) goto bad;
    return 0;
bad:
    snprintf(err, errlen, "aggregate '%s' is outside the oracle subset", s);
    return -1;
}

/* bind E_PATH nodes to table columns by exact stringer text */
static int bind(node *n, const n1o_table *t, char *err, size_t errlen) {
    if (!n) return 0;
    if (n->kind == E_PATH) {
        for (uint32_t c = 0; c < t->ncols; c++)
            if (!strcmp(t->names[c], n->text)) {
                n->col = (int)c;
                return 0;
            }
        snprintf(err, errlen, "no column for leaf path %s", n->text);
        return -1;
    }
    for (int i = 0; i < n->nch; i++)
        if (bind(n->ch[i], t, err, errlen)) return -1;
    return 0;
}

/* ------------------------------------------------------------- evaluation */

typedef struct ectx {
    const n1o_table *t;
    int unsupported; /* run-time value outside the restated subset */
} ectx;

/* Leaf access: what Field.Apply / Identifier.Evaluate return for this row
 * (expression/nav_field.go:134-160, identifier.go:48-51) */
static val load_col(const n1o_table *t, int col, uint64_t row) {
    const n1k_col *c = &t->cols[col];
    val v = V_NULL;
    uint8_t tag;
    uint64_t pay;
    if (c->kind == N1K_COL_DICT32) {
        uint32_t code = c->codes[row];
        if (code == N1K_CODE_MISSING) return V_MISSING;
        if (code == N1K_CODE_NULL) return V_NULL;
        tag = N1K_T_STRING;
        pay = code;
    } else {
        tag = c->tags[row];
        pay = c->payload[row];
    }
    switch (tag) {
    case N1K_T_MISSING: return V_MISSING;
    case N1K_T_NULL: return V_NULL;
    case N1K_T_FALSE: return v_bool(0);
    case N1K_T_TRUE: return v_bool(1);
    case N1K_T_INT: return v_int((int64_t)pay);
    case N1K_T_FLOAT: {
        double d;
        memcpy(&d, &pay, 8);
        return v_float(d);
    }
    case N1K_T_STRING:
    case N1K_T_ARRAY:
    case N1K_T_OBJECT:
        v.type = tag == N1K_T_STRING ? TY_STRING : (tag == N1K_T_ARRAY ? TY_ARRAY : TY_OBJECT);
        v.code = (uint32_t)pay;
        if (v.code >= t->dict_n) {
            v.s = "";
            v.slen = 0;
        } else {
            v.s = t->dict_bytes + t->dict_offsets[v.code];
            v.slen = (uint32_t)(t->dict_offsets[v.code + 1] - t->dict_offsets[v.code]);
        }
        return v;
    default: return V_NULL;
    }
}

/* roundFloat (expression/func_num.go:1715-1736) */
static double round_float(double x, int prec) {
    if (isnan(x) || isinf(x)) return x;
    double sign = 1.0;
    if (x < 0) { sign = -1.0; x = -x; }
    double pw = pow(10, (double)prec);
    double intermed = x * pw + 0.5;
    double rounder = floor(intermed);
    if (rounder == intermed && fmod(rounder, 2) != 0) rounder--;
    return sign * rounder / pw;
}

static val eval(const node *n, uint64_t row, ectx *cx) {
    switch (n->kind) {
    case E_ROUND: case E_TRUNC: case E_ABS: case E_CEIL: case E_FLOOR: case E_SIGN: case E_SQRT: {
        /* Round.Apply func_num.go:1304-1333, Trunc.Apply :1647-1680, Abs :63-71, Ceil :367-375, Floor :774-782,
         * Sign :1396-1412, Sqrt :1524-1532: MISSING in, MISSING out; a non-number is NULL; the number goes through
         * float64 (intValue.Actual(), value/integer.go:57-59) and the result through value.NewValue */
        val a = eval(n->ch[0], row, cx);
        if (a.type == TY_MISSING) return V_MISSING;
        if (a.type != TY_NUMBER) return V_NULL;
        double v = num_actual(a);
        int prec = 0;
        if (n->nch == 2) {
            val pv = eval(n->ch[1], row, cx);
            if (pv.type == TY_MISSING) return V_MISSING;
            if (pv.type != TY_NUMBER) return V_NULL;
            double pf = num_actual(pv);
            if (pf != trunc(pf)) return V_NULL;
            prec = (int)pf;
        }
        switch (n->kind) {
        case E_ROUND: return new_value_f64(round_float(v, prec));
        case E_TRUNC: { double pw = pow(10, (double)prec); return new_value_f64(trunc(v * pw) / pw); } /* truncateFloat :1703-1708 */
        case E_ABS: return new_value_f64(fabs(v));
        case E_CEIL: return new_value_f64(ceil(v));
        case E_FLOOR: return new_value_f64(floor(v));
        case E_SIGN: return new_value_f64(v < 0.0 ? -1.0 : (v > 0.0 ? 1.0 : 0.0));
        default: return new_value_f64(sqrt(v));
        }
    }
    case E_GREATEST: case E_LEAST: {
        /* Greatest.Apply func_comp.go:54-67 / Least.Apply :124-138: the largest (smallest) argument above NULL in
         * value.Collate order, the first one on ties; NULL when there is none */
        val rv = V_NULL;
        for (int i = 0; i < n->nch; i++) {
            val a = eval(n->ch[i], row, cx);
            if (a.type <= TY_NULL) continue;
            if (rv.type == TY_NULL) rv = a;
            else {
                int c = collate(a, rv, &cx->unsupported);
                if (n->kind == E_GREATEST ? c > 0 : c < 0) rv = a;
            }
        }
        return rv;
    }
    case E_CONST: return n->cval;
    case E_PATH: return load_col(cx->t, n->col, row);
    case E_ADD: { /* expression/arith_add.go:51-70 */
        int null = 0;
        val sum = v_int(0);
        for (int i = 0; i < n->nch; i++) {
            val a = eval(n->ch[i], row, cx);
            if (!null && a.type == TY_NUMBER) sum = num_add(sum, a);
            else if (a.type == TY_MISSING) return V_MISSING;
            else null = 1;
        }
        return null ? V_NULL : sum;
    }
    case E_MULT: { /* expression/arith_mult.go:51-70 */
        int null = 0;
        val prod = v_int(1);
        for (int i = 0; i < n->nch; i++) {
            val a = eval(n->ch[i], row, cx);
            if (!null && a.type == TY_NUMBER) prod = num_mult(prod, a);
            else if (a.type == TY_MISSING) return V_MISSING;
            else null = 1;
        }
        return null ? V_NULL : prod;
    }
    case E_SUB: { /* expression/arith_sub.go:53-61 */
        val a = eval(n->ch[0], row, cx), b = eval(n->ch[1], row, cx);
        if (a.type == TY_NUMBER && b.type == TY_NUMBER) return num_sub(a, b);
        if (a.type == TY_MISSING || b.type == TY_MISSING) return V_MISSING;
        return V_NULL;
    }
    case E_DIV: { /* expression/arith_div.go:46-64 */
        val a = eval(n->ch[0], row, cx), b = eval(n->ch[1], row, cx);
        if (a.type == TY_MISSING || b.type == TY_MISSING) return V_MISSING;
        if (b.type == TY_NUMBER) {
            double s = num_actual(b);
            if (s == 0.0) return V_NULL;
            if (a.type == TY_NUMBER) return new_value_f64(num_actual(a) / s);
        }
        return V_NULL;
    }
    case E_MOD: { /* expression/arith_mod.go:48-66 */
        val a = eval(n->ch[0], row, cx), b = eval(n->ch[1], row, cx);
        if (a.type == TY_MISSING || b.type == TY_MISSING) return V_MISSING;
        if (b.type == TY_NUMBER) {
            double s = num_actual(b);
            if (s == 0.0) return V_NULL;
            if (a.type == TY_NUMBER) return new_value_f64(fmod(num_actual(a), s));
        }
        return V_NULL;
    }
    case E_NEG: { /* expression/arith_neg.go:51-59 */
        val a = eval(n->ch[0], row, cx);
        if (a.type == TY_NUMBER) return num_neg(a);
        if (a.type == TY_MISSING) return V_MISSING;
        return V_NULL;
    }
    case E_IDIV:   /* expression/arith_idiv.go:46-56 */
    case E_IMOD: { /* expression/arith_imod.go */
        val a = eval(n->ch[0], row, cx), b = eval(n->ch[1], row, cx);
        if (a.type == TY_MISSING || b.type == TY_MISSING) return V_MISSING;
        if (a.type == TY_NUMBER && b.type == TY_NUMBER) return n->kind == E_IDIV ? num_idiv(a, b) : num_imod(a, b);
        return V_NULL;
    }
    case E_EQ: { /* expression/comp_eq.go:76-78 */
        val a = eval(n->ch[0], row, cx), b = eval(n->ch[1], row, cx);
        return equals(a, b, &cx->unsupported);
    }
    case E_LT:   /* expression/comp_lt.go:57-65 */
    case E_LE: { /* expression/comp_le.go:57-65 */
        val a = eval(n->ch[0], row, cx), b = eval(n->ch[1], row, cx);
        val cmp = compare(a, b, &cx->unsupported);
        if (cmp.type != TY_NUMBER) return cmp;
        return v_bool(n->kind == E_LT ? cmp.i < 0 : cmp.i <= 0);
    }
    case E_BETWEEN: { /* expression/comp_between.go:58-78 */
        val it = eval(n->ch[0], row, cx), lo = eval(n->ch[1], row, cx), hi = eval(n->ch[2], row, cx);
        val lc = compare(it, lo, &cx->unsupported);
        if (lc.type == TY_MISSING) return lc;
        val hc = compare(it, hi, &cx->unsupported);
        if (hc.type == TY_MISSING) return hc;
        if (lc.type == TY_NUMBER && hc.type == TY_NUMBER) return v_bool(lc.i >= 0 && hc.i <= 0);
        return V_NULL;
    }
    case E_AND: { /* expression/logic_and.go:64-89 */
        int missing = 0, null = 0;
        for (int i = 0; i < n->nch; i++) {
            val a = eval(n->ch[i], row, cx);
            if (a.type == TY_NULL) null = 1;
            else if (a.type == TY_MISSING) missing = 1;
            else if (!truth(a, &cx->unsupported)) return v_bool(0);
        }
        return missing ? V_MISSING : (null ? V_NULL : v_bool(1));
    }
    case E_OR: { /* expression/logic_or.go:98-123 */
        int missing = 0, null = 0;
        for (int i = 0; i < n->nch; i++) {
            val a = eval(n->ch[i], row, cx);
            if (a.type == TY_NULL) null = 1;
            else if (a.type == TY_MISSING) missing = 1;
            else if (truth(a, &cx->unsupported)) return v_bool(1);
        }
        return null ? V_NULL : (missing ? V_MISSING : v_bool(0));
    }
    case E_NOT: { /* expression/logic_not.go:57-69 */
        val a = eval(n->ch[0], row, cx);
        if (a.type == TY_MISSING || a.type == TY_NULL) return a;
        return v_bool(!truth(a, &cx->unsupported));
    }
    case E_ISNULL: { /* expression/comp_null.go:58-67 */
        val a = eval(n->ch[0], row, cx);
        return a.type == TY_NULL ? v_bool(1) : (a.type == TY_MISSING ? V_MISSING : v_bool(0));
    }
    case E_ISNOTNULL: { /* expression/comp_null.go:116-125 */
        val a = eval(n->ch[0], row, cx);
        return a.type == TY_NULL ? v_bool(0) : (a.type == TY_MISSING ? V_MISSING : v_bool(1));
    }
    case E_ISMISSING: /* expression/comp_missing.go:62-69 */
        return v_bool(eval(n->ch[0], row, cx).type == TY_MISSING);
    case E_ISNOTMISSING: /* expression/comp_missing.go:125-132 */
        return v_bool(eval(n->ch[0], row, cx).type != TY_MISSING);
    case E_ISVALUED: { /* expression/comp_valued.go:61-68 */
        val a = eval(n->ch[0], row, cx);
        return v_bool(!(a.type == TY_NULL || a.type == TY_MISSING));
    }
    case E_ISNOTVALUED: { /* expression/comp_valued.go:124-131 */
        val a = eval(n->ch[0], row, cx);
        return v_bool(a.type == TY_NULL || a.type == TY_MISSING);
    }
    }
    return V_NULL;
}

/* ------------------------------------------------------------------- sets */

/* value/set.go:22-35: one hash set per type; only the members the path can produce */
typedef struct u64set {
    uint64_t *slots;
    uint8_t *used;
    size_t cap, n;
} u64set;
static uint64_t mix64(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}
static void u64set_add(u64set *s, uint64_t k);
static void u64set_grow(u64set *s) {
    u64set o = *s;
    s->cap = o.cap ? o.cap * 2 : 16;
    s->slots = calloc(s->cap, 8);
    s->used = calloc(s->cap, 1);
    s->n = 0;
    for (size_t i = 0; i < o.cap; i++)
        if (o.used[i]) u64set_add(s, o.slots[i]);
    free(o.slots);
    free(o.used);
}
static void u64set_add(u64set *s, uint64_t k) {
    if ((s->n + 1) * 2 > s->cap) u64set_grow(s);
    size_t i = mix64(k) & (s->cap - 1);
    while (s->used[i]) {
        if (s->slots[i] == k) return;
        i = (i + 1) & (s->cap - 1);
    }
    s->used[i] = 1;
    s->slots[i] = k;
    s->n++;
}
static void u64set_free(u64set *s) {
    free(s->slots);
    free(s->used);
    memset(s, 0, sizeof *s);
}

typedef struct vset {
    int has_false, has_true; /* booleans map[bool] */
    u64set ints;             /* ints map[int64] */
    u64set floats;           /* floats map[float64] keyed by bits (NaN never produced on this path) */
    u64set strings;          /* strings map[string]: dictionary codes (equal bytes <=> equal codes) */
    u64set arrays, objects;  /* arrays/objects map[string]: keyed by canonical text code */
} vset;

/* value/set.go:65-110 Put */
static void set_add(vset *s, val v, ectx *cx) {
    switch (v.type) {
    case TY_BOOLEAN:
        if (v.b) s->has_true = 1; else s->has_false = 1;
        break;
    case TY_NUMBER:
        if (v.isf) {
            if (is_int(v.f)) u64set_add(&s->ints, (uint64_t)go_f2i(v.f));
            else {
                uint64_t b;
                memcpy(&b, &v.f, 8);
                u64set_add(&s->floats, b);
            }
        } else
            u64set_add(&s->ints, (uint64_t)v.i);
        break;
    case TY_STRING:
        if (v.code == UINT32_MAX) cx->unsupported = 1; /* constant strings never reach a set on this path */
        u64set_add(&s->strings, v.code);
        break;
    case TY_ARRAY: u64set_add(&s->arrays, v.code); break;
    case TY_OBJECT: u64set_add(&s->objects, v.code); break;
    default: break;
    }
}
/* value/set.go:198-215 Len */
static int64_t set_len(const vset *s) {
    return (int64_t)(s->has_false + s->has_true) + (int64_t)s->ints.n + (int64_t)s->floats.n +
           (int64_t)s->strings.n + (int64_t)s->arrays.n + (int64_t)s->objects.n;
}
/* algebra/agg_util.go:52-81 cumulateSets: add the smaller into the bigger (order irrelevant for membership) */
static void set_union(vset *dst, const vset *src) {
    dst->has_false |= src->has_false;
    dst->has_true |= src->has_true;
    const u64set *ss[5] = {&src->ints, &src->floats, &src->strings, &src->arrays, &src->objects};
    u64set *dd[5] = {&dst->ints, &dst->floats, &dst->strings, &dst->arrays, &dst->objects};
    for (int k = 0; k < 5; k++)
        for (size_t i = 0; i < ss[k]->cap; i++)
            if (ss[k]->used[i]) u64set_add(dd[k], ss[k]->slots[i]);
}
static void set_free(vset *s) {
    u64set_free(&s->ints);
    u64set_free(&s->floats);
    u64set_free(&s->strings);
    u64set_free(&s->arrays);
    u64set_free(&s->objects);
}
/* sum of the NUMBER members (SumDistinct/AvgDistinct.ComputeFinal, algebra/agg_sum_distinct.go:113-133) */
static val set_sum_numbers(const vset *s) {
    val sum = v_int(0);
    for (size_t i = 0; i < s->ints.cap; i++)
        if (s->ints.used[i]) sum = num_add(sum, v_int((int64_t)s->ints.slots[i]));
    for (size_t i = 0; i < s->floats.cap; i++)
        if (s->floats.used[i]) {
            double d;
            memcpy(&d, &s->floats.slots[i], 8);
            sum = num_add(sum, v_float(d));
        }
    return sum;
}

/* ------------------------------------------------------------- aggregates */

typedef struct aggstate {
    val cum;        /* SUM/COUNT/COUNTN/MIN/MAX cumulative; AVG: sum */
    val avg_count;  /* AVG: count */
    uint8_t is_null;/* AVG: cumulative == NULL_VALUE; DISTINCT: no set yet (ZERO/NULL default) */
    vset *set;
    val *items;     /* ARRAY_AGG: the cumulative array ([]interface{}), in arrival order until ComputeFinal sorts it */
    size_t nitems, capitems;
} aggstate;

static void items_push(aggstate *st, val v) {
    if (st->nitems == st->capitems) {
        st->capitems = st->capitems ? st->capitems * 2 : 8;
        st->items = realloc(st->items, st->capitems * sizeof(val));
    }
    st->items[st->nitems++] = v;
}

/* Default(): algebra/agg_sum.go:77, agg_count.go:95, agg_countn.go:77, agg_avg.go:77, agg_min.go:76, agg_max.go:76,
 * agg_count_distinct.go:76, agg_sum_distinct.go:79 */
static void agg_default(const aggdef *a, aggstate *st) {
    memset(st, 0, sizeof *st);
    st->is_null = 1;
    if (a->distinct) {
        st->cum = (a->kind == A_COUNT || a->kind == A_COUNTN) ? v_int(0) : V_NULL;
        return;
    }
    switch (a->kind) {
    case A_COUNT:
    case A_COUNTN: st->cum = v_int(0); break;
    default: st->cum = V_NULL;
    }
}

/* CumulateInitial */
static void agg_cumulate_initial(const aggdef *a, aggstate *st, uint64_t row, ectx *cx) {
    val item = V_NULL;
    if (a->operand) item = eval(a->operand, row, cx);
    if (a->kind == A_ARRAY) {
        /* ArrayAgg.CumulateInitial algebra/agg_array.go:86-97 / ArrayAggDistinct agg_array_distinct.go:86-97: every
         * operand but MISSING (BINARY does not exist on this path) joins; DISTINCT is applied by ComputeFinal here
         * (value.Set membership: the sorted members that collate equal are one) */
        if (item.type <= TY_MISSING) return;
        st->is_null = 0;
        items_push(st, item);
        return;
    }
    if (a->distinct) {
        /* agg_count_distinct.go:84-95 (type <= NULL skipped); agg_countn_distinct.go / agg_sum_distinct.go:85-97 /
         * agg_avg_distinct.go:86-98 (non-NUMBER skipped); setAdd: agg_util.go:30-47 */
        if (a->kind == A_COUNT) {
            if (item.type <= TY_NULL) return;
        } else if (item.type != TY_NUMBER)
            return;
        if (!st->set) st->set = calloc(1, sizeof(vset));
        st->is_null = 0;
        set_add(st->set, item, cx);
        return;
    }
    switch (a->kind) {
    case A_SUM: /* agg_sum.go:86-97,118-136 */
        if (item.type != TY_NUMBER) return;
        st->cum = st->cum.type == TY_NULL ? item : num_add(st->cum, item);
        break;
    case A_COUNT: /* agg_count.go:102-116 */
        if (a->operand && item.type <= TY_NULL) return;
        st->cum = num_add(st->cum, v_int(1));
        break;
    case A_COUNTN: /* agg_countn.go:84-97 */
        if (item.type != TY_NUMBER) return;
        st->cum = num_add(st->cum, v_int(1));
        break;
    case A_AVG: /* agg_avg.go:85-97,136-157 */
        if (item.type != TY_NUMBER) return;
        if (st->is_null) {
            st->cum = item;
            st->avg_count = v_int(1);
            st->is_null = 0;
        } else {
            st->cum = num_add(st->cum, item);
            st->avg_count = num_add(st->avg_count, v_int(1));
        }
        break;
    case A_MIN: /* agg_min.go:83-94,117-127 */
        if (item.type <= TY_NULL) return;
        if (st->cum.type == TY_NULL || collate(item, st->cum, &cx->unsupported) < 0) st->cum = item;
        break;
    case A_MAX: /* agg_max.go:83-94,117-127 */
        if (item.type <= TY_NULL) return;
        if (st->cum.type == TY_NULL || collate(item, st->cum, &cx->unsupported) > 0) st->cum = item;
        break;
    }
}

/* CumulateIntermediate(part, cumulative) */
static void agg_cumulate_intermediate(const aggdef *a, const aggstate *part, aggstate *cum, ectx *cx) {
    if (a->kind == A_ARRAY) { /* cumulatePart agg_array.go:118-145: append; cumulateSets for DISTINCT */
        for (size_t i = 0; i < part->nitems; i++) items_push(cum, part->items[i]);
        if (part->nitems) cum->is_null = 0;
        return;
    }
    if (a->distinct) {
        /* agg_count_distinct.go:103-111; a partial without a set is the ZERO/NULL default. For SUM/AVG DISTINCT the
         * reference would raise "Invalid DISTINCT" on a NULL partial (agg_util.go:87-101); it is treated as the
         * empty set here. */
        if (!part->set) return;
        if (!cum->set) cum->set = calloc(1, sizeof(vset));
        cum->is_null = 0;
        set_union(cum->set, part->set);
        return;
    }
    switch (a->kind) {
    case A_SUM: /* agg_sum.go:102-104,118-136 */
        if (part->cum.type == TY_NULL) return;
        cum->cum = cum->cum.type == TY_NULL ? part->cum : num_add(cum->cum, part->cum);
        break;
    case A_COUNT:
    case A_COUNTN: /* agg_count.go:121-123,137-149 */
        cum->cum = num_add(cum->cum, part->cum);
        break;
    case A_AVG: /* agg_avg.go:102-104,136-157 */
        if (part->is_null) return;
        if (cum->is_null) {
            cum->cum = part->cum;
            cum->avg_count = part->avg_count;
            cum->is_null = 0;
        } else {
            cum->cum = num_add(cum->cum, part->cum);
            cum->avg_count = num_add(cum->avg_count, part->avg_count);
        }
        break;
    case A_MIN:
        if (part->cum.type == TY_NULL) return;
        if (cum->cum.type == TY_NULL || collate(part->cum, cum->cum, &cx->unsupported) < 0) cum->cum = part->cum;
        break;
    case A_MAX:
        if (part->cum.type == TY_NULL) return;
        if (cum->cum.type == TY_NULL || collate(part->cum, cum->cum, &cx->unsupported) > 0) cum->cum = part->cum;
        break;
    }
}

/* ComputeFinal */
static val agg_compute_final(const aggdef *a, const aggstate *st) {
    if (a->distinct) {
        switch (a->kind) {
        case A_COUNT:
        case A_COUNTN: /* agg_count_distinct.go:118-126 */
            return st->set ? v_int(set_len(st->set)) : v_int(0);
        case A_SUM: /* agg_sum_distinct.go:113-133 */
            if (!st->set || set_len(st->set) == 0) return V_NULL;
            return set_sum_numbers(st->set);
        case A_AVG: /* agg_avg_distinct.go:114-134 */
            if (!st->set || set_len(st->set) == 0) return V_NULL;
            return new_value_f64(num_actual(set_sum_numbers(st->set)) / (double)set_len(st->set));
        }
    }
    if (a->kind == A_AVG) { /* agg_avg.go:111-129 */
        if (st->is_null) return V_NULL;
        double c = num_actual(st->avg_count);
        if (c > 0.0) return new_value_f64(num_actual(st->cum) / c);
        return V_NULL;
    }
    return st->cum; /* agg_sum.go:109-111, agg_count.go:128-130, agg_min.go:107-109 */
}

/* -------------------------------------------------------------- group key */

typedef struct strbuf {
    char *p;
    size_t n, cap;
} strbuf;
static void sb_put(strbuf *b, const void *d, size_t n) {
    if (b->n + n > b->cap) {
        b->cap = (b->n + n) * 2 + 32;
        b->p = realloc(b->p, b->cap);
    }
    memcpy(b->p + b->n, d, n);
    b->n += n;
}

/* strconv.FormatFloat(f, 'f', -1, 64): shortest digits that round-trip, positional notation
 * (value/float.go:31-48; -0 prints as 0) */
static void format_float_f(double f, strbuf *b) {
    if (isnan(f)) { sb_put(b, "\"NaN\"", 5); return; }
    if (isinf(f)) { if (f > 0) sb_put(b, "\"+Infinity\"", 11); else sb_put(b, "\"-Infinity\"", 11); return; }
    if (f == 0) { sb_put(b, "0", 1); return; }
    char e[40];
    int prec;
    for (prec = 0; prec < 17; prec++) {
        snprintf(e, sizeof e, "%.*e", prec, f);
        if (strtod(e, NULL) == f) break;
    }
    /* e = [-]d.ddddde[+-]xx -> digits + exponent */
    char digits[24];
    int nd = 0, neg = 0;
    const char *q = e;
    if (*q == '-') { neg = 1; q++; }
    for (; *q && *q != 'e'; q++)
        if (*q != '.') digits[nd++] = *q;
    int ex = atoi(q + 1);
    while (nd > 1 && digits[nd - 1] == '0') nd--; /* shortest */
    if (neg) sb_put(b, "-", 1);
    int pointpos = ex + 1; /* digits before the decimal point */
    if (pointpos <= 0) {
        sb_put(b, "0.", 2);
        for (int i = 0; i < -pointpos; i++) sb_put(b, "0", 1);
        sb_put(b, digits, (size_t)nd);
    } else if (pointpos >= nd) {
        sb_put(b, digits, (size_t)nd);
        for (int i = nd; i < pointpos; i++) sb_put(b, "0", 1);
    } else {
        sb_put(b, digits, (size_t)pointpos);
        sb_put(b, ".", 1);
        sb_put(b, digits + pointpos, (size_t)(nd - pointpos));
    }
}

/* groupKey(): execution/group_util.go:18-35 — the canonical JSON of {" ":k0,"":k1,...} without the
 * MISSING keys.  Any injective re-encoding of that JSON gives the same grouping; numbers keep the exact
 * reference text (value/integer.go:34-37, value/float.go:31-48) because int 5 and float 5.0 must collide. */
static void group_key(node *const *keys, uint32_t nkeys, uint64_t row, ectx *cx, strbuf *b, val *keyvals) {
    b->n = 0;
    for (uint32_t i = 0; i < nkeys; i++) {
        val k = eval(keys[i], row, cx);
        keyvals[i] = k;
        if (k.type == TY_MISSING) continue;
        char name = (char)i;
        sb_put(b, &name, 1);
        switch (k.type) {
        case TY_NULL: sb_put(b, "n", 1); break;
        case TY_BOOLEAN: sb_put(b, k.b ? "t" : "f", 1); break;
        case TY_NUMBER: {
            if (k.isf && (isnan(k.f) || isinf(k.f))) {
                /* value/float.go:31-48: NaN / +Inf / -Inf marshal as the JSON STRINGS "NaN" / "+Infinity" / "-Infinity": such a
                 * key is the same map key as that string */
                const char *t = isnan(k.f) ? "NaN" : (k.f > 0 ? "+Infinity" : "-Infinity");
                uint32_t l = (uint32_t)strlen(t);
                sb_put(b, "s", 1);
                sb_put(b, &l, 4);
                sb_put(b, t, l);
                break;
            }
            sb_put(b, "#", 1);
            if (k.isf) format_float_f(k.f, b);
            else {
                char t[24];
                int n = snprintf(t, sizeof t, "%lld", (long long)k.i);
                sb_put(b, t, (size_t)n);
            }
            break;
        }
        default: { /* STRING / ARRAY / OBJECT: type char + length-prefixed bytes */
            char ty = k.type == TY_STRING ? 's' : (k.type == TY_ARRAY ? 'a' : 'o');
            sb_put(b, &ty, 1);
            uint32_t l = k.slen;
            sb_put(b, &l, 4);
            sb_put(b, k.s, k.slen);
        }
        }
        sb_put(b, ",", 1);
    }
}

/* ----------------------------------------------------------- group tables */

typedef struct group {
    char *key;
    uint32_t keylen;
    uint64_t hash;
    uint64_t first_row;
    val *keyvals;
    aggstate *aggs;
} group;

typedef struct gmap { /* map[string]AnnotatedValue  (execution/group_initial.go:22-26) */
    group **slots;
    size_t cap, n;
    group **list;
    size_t list_cap;
} gmap;

static uint64_t hash_bytes(const char *p, size_t n) {
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n; i++) {
        h ^= (unsigned char)p[i];
        h *= 0x100000001b3ull;
    }
    return mix64(h);
}
static void gmap_insert_raw(gmap *m, group *g) {
    size_t i = g->hash & (m->cap - 1);
    while (m->slots[i]) i = (i + 1) & (m->cap - 1);
    m->slots[i] = g;
}
static group *gmap_find(gmap *m, const char *key, uint32_t len, uint64_t h) {
    if (!m->cap) return NULL;
    size_t i = h & (m->cap - 1);
    while (m->slots[i]) {
        group *g = m->slots[i];
        if (g->hash == h && g->keylen == len && memcmp(g->key, key, len) == 0) return g;
        i = (i + 1) & (m->cap - 1);
    }
    return NULL;
}
static void gmap_add(gmap *m, group *g) {
    if ((m->n + 1) * 2 > m->cap) {
        size_t nc = m->cap ? m->cap * 2 : 64;
        free(m->slots);
        m->slots = calloc(nc, sizeof(group *));
        m->cap = nc;
        for (size_t i = 0; i < m->n; i++) gmap_insert_raw(m, m->list[i]);
    }
    if (m->n == m->list_cap) {
        m->list_cap = m->list_cap ? m->list_cap * 2 : 64;
        m->list = realloc(m->list, m->list_cap * sizeof(group *));
    }
    m->list[m->n++] = g;
    gmap_insert_raw(m, g);
}
static void group_free(group *g, uint32_t naggs) {
    for (uint32_t i = 0; i < naggs; i++) {
        free(g->aggs[i].items);
        if (g->aggs[i].set) {
            set_free(g->aggs[i].set);
            free(g->aggs[i].set);
        }
    }
    free(g->key);
    free(g->keyvals);
    free(g->aggs);
    free(g);
}
static void gmap_free(gmap *m, uint32_t naggs, int free_groups) {
    if (free_groups)
        for (size_t i = 0; i < m->n; i++) group_free(m->list[i], naggs);
    free(m->slots);
    free(m->list);
    memset(m, 0, sizeof *m);
}

/* ----------------------------------------------------------------- engine */

typedef struct plan {
    node *cond;
    uint32_t nkeys, naggs;
    node **keys;
    aggdef *aggs;
    const n1o_table *t;
} plan;

typedef struct worker {
    const plan *pl;
    pthread_t th;
    gmap groups; /* one private map per Parallel copy: execution/group_initial.go:43-50 */
    uint64_t *next_chunk;
    uint64_t chunk_rows;
    uint64_t rows_passed;
    int unsupported;
} worker;

/* One Parallel copy: Sequence[Filter, InitialGroup] pulling items from the shared input
 * (execution/parallel.go:67-83, execution/base.go:533-540). */
static void *run_worker(void *arg) {
    worker *w = arg;
    const plan *pl = w->pl;
    ectx cx = {pl->t, 0};
    strbuf kb = {0};
    val *kv = malloc(sizeof(val) * (pl->nkeys ? pl->nkeys : 1));
    for (;;) {
        uint64_t c = __atomic_fetch_add(w->next_chunk, 1, __ATOMIC_RELAXED);
        uint64_t r0 = c * w->chunk_rows;
        if (r0 >= pl->t->nrows) break;
        uint64_t r1 = r0 + w->chunk_rows;
        if (r1 > pl->t->nrows) r1 = pl->t->nrows;
        for (uint64_t row = r0; row < r1; row++) {
            /* Filter.processItem: execution/filter.go:49-61 */
            if (pl->cond) {
                val v = eval(pl->cond, row, &cx);
                if (!truth(v, &cx.unsupported)) continue;
            }
            w->rows_passed++;
            /* InitialGroup.processItem: execution/group_initial.go:56-100 */
            kb.n = 0;
            if (pl->nkeys) group_key(pl->keys, pl->nkeys, row, &cx, &kb, kv);
            uint64_t h = hash_bytes(kb.p, kb.n);
            group *g = gmap_find(&w->groups, kb.p, (uint32_t)kb.n, h);
            if (!g) {
                g = calloc(1, sizeof *g);
                g->key = malloc(kb.n ? kb.n : 1);
                memcpy(g->key, kb.p, kb.n);
                g->keylen = (uint32_t)kb.n;
                g->hash = h;
                g->first_row = row; /* the first row met is the carrier (:69-72) */
                g->keyvals = malloc(sizeof(val) * (pl->nkeys ? pl->nkeys : 1));
                memcpy(g->keyvals, kv, sizeof(val) * pl->nkeys);
                g->aggs = malloc(sizeof(aggstate) * (pl->naggs ? pl->naggs : 1));
                for (uint32_t a = 0; a < pl->naggs; a++) agg_default(&pl->aggs[a], &g->aggs[a]);
                gmap_add(&w->groups, g);
            }
            for (uint32_t a = 0; a < pl->naggs; a++) agg_cumulate_initial(&pl->aggs[a], &g->aggs[a], row, &cx);
        }
    }
    w->unsupported = cx.unsupported;
    free(kb.p);
    free(kv);
    return NULL;
}

static void out_value(val v, n1k_value *o, int *bad) {
    memset(o, 0, sizeof *o);
    switch (v.type) {
    case TY_MISSING: o->tag = N1K_T_MISSING; break;
    case TY_NULL: o->tag = N1K_T_NULL; break;
    case TY_BOOLEAN: o->tag = v.b ? N1K_T_TRUE : N1K_T_FALSE; break;
    case TY_NUMBER:
        if (v.isf) {
            o->tag = N1K_T_FLOAT;
            o->v.f = v.f;
        } else {
            o->tag = N1K_T_INT;
            o->v.i = v.i;
        }
        break;
    case TY_STRING:
    case TY_ARRAY:
    case TY_OBJECT:
        o->tag = v.type == TY_STRING ? N1K_T_STRING : (v.type == TY_ARRAY ? N1K_T_ARRAY : N1K_T_OBJECT);
        o->v.code = v.code;
        if (v.code == UINT32_MAX) *bad = 1;
        break;
    default: *bad = 1;
    }
}

/* ArrayAgg.ComputeFinal (algebra/agg_array.go:105-112): NULL when nothing joined, else the array sorted with
 * value.NewSorter (Collate).  ArrayAggDistinct.ComputeFinal (agg_array_distinct.go:111-127): the set's members,
 * sorted; an empty set is NULL.  The array leaves as its canonical JSON text (value.MarshalJSON: numbers as
 * integer.go:34-37 / float.go:31-48, strings without HTML escaping) in the result's extra strings; its code is
 * dict_n + index. */
static int cmp_collate_ctx_unsupported;
static int cmp_collate(const void *x, const void *y) {
    return collate(*(const val *)x, *(const val *)y, &cmp_collate_ctx_unsupported);
}
static void sb_put_json_string(strbuf *b, const char *s, uint32_t n) {
    sb_put(b, "\"", 1);
    for (uint32_t i = 0; i < n; i++) {
        unsigned char c = (unsigned char)s[i];
        char esc[8];
        switch (c) {
        case '"': sb_put(b, "\\\"", 2); break;
        case '\\': sb_put(b, "\\\\", 2); break;
        case '\n': sb_put(b, "\\n", 2); break;
        case '\r': sb_put(b, "\\r", 2); break;
        case '\t': sb_put(b, "\\t", 2); break;
        case '\b': sb_put(b, "\\b", 2); break;
        case '\f': sb_put(b, "\\f", 2); break;
        default:
            if (c < 0x20) { snprintf(esc, sizeof esc, "\\u%04x", c); sb_put(b, esc, 6); }
            else sb_put(b, &s[i], 1);
        }
    }
    sb_put(b, "\"", 1);
}
static void array_agg_final(const aggdef *a, aggstate *st, const n1o_table *t, ectx *cx, strbuf *extra, uint64_t **off,
                            uint32_t *n, uint32_t *cap, n1k_value *out) {
    memset(out, 0, sizeof *out);
    out->tag = N1K_T_NULL;
    if (st->nitems == 0) return;
    cmp_collate_ctx_unsupported = 0;
    /* sort.Sort is not stable, Collate is total on this path's scalars: any stable choice gives the same JSON */
    qsort(st->items, st->nitems, sizeof(val), cmp_collate);
    if (cmp_collate_ctx_unsupported) cx->unsupported = 1;
    if (*n + 2 > *cap) {
        *cap = *cap ? *cap * 2 : 64;
        *off = realloc(*off, ((size_t)*cap + 1) * sizeof(uint64_t));
    }
    if (*n == 0) (*off)[0] = 0;
    sb_put(extra, "[", 1);
    int first = 1;
    for (size_t i = 0; i < st->nitems; i++) {
        val v = st->items[i];
        if (a->distinct && i > 0) {
            int u = 0;
            if (collate(st->items[i - 1], v, &u) == 0) continue; /* one member (value/set.go:65-110) */
        }
        if (!first) sb_put(extra, ",", 1);
        first = 0;
        char num[32];
        switch (v.type) {
        case TY_NULL: sb_put(extra, "null", 4); break;
        case TY_BOOLEAN: if (v.b) sb_put(extra, "true", 4); else sb_put(extra, "false", 5); break;
        case TY_NUMBER:
            if (v.isf) format_float_f(v.f, extra);
            else { int k = snprintf(num, sizeof num, "%lld", (long long)v.i); sb_put(extra, num, (size_t)k); }
            break;
        case TY_STRING: sb_put_json_string(extra, v.s, v.slen); break;
        default: sb_put(extra, v.s, v.slen); break; /* arrays / objects: canonical text */
        }
    }
    sb_put(extra, "]", 1);
    (*off)[++*n] = extra->n;
    out->tag = N1K_T_ARRAY;
    out->v.code = (uint64_t)t->dict_n + (*n - 1);
}

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void plan_free(plan *pl) {
    free_node(pl->cond);
    for (uint32_t i = 0; i < pl->nkeys; i++) free_node(pl->keys[i]);
    free(pl->keys);
    for (uint32_t i = 0; i < pl->naggs; i++) free_node(pl->aggs[i].operand);
    free(pl->aggs);
}

int n1o_run(const char *condition, const char *const *keys, uint32_t nkeys, const char *const *aggs,
            uint32_t naggs, int has_group, const n1o_table *t, int threads, n1o_result *out) {
    memset(out, 0, sizeof *out);
    plan pl;
    memset(&pl, 0, sizeof pl);
    pl.t = t;
    if (condition && condition[0]) {
        pl.cond = parse_full(condition, out->err, sizeof out->err);
        if (!pl.cond) return 1;
        if (bind(pl.cond, t, out->err, sizeof out->err)) { plan_free(&pl); return 1; }
    }
    pl.nkeys = nkeys;
    pl.naggs = naggs;
    pl.keys = calloc(nkeys ? nkeys : 1, sizeof(node *));
    pl.aggs = calloc(naggs ? naggs : 1, sizeof(aggdef));
    for (uint32_t i = 0; i < nkeys; i++) {
        pl.keys[i] = parse_full(keys[i], out->err, sizeof out->err);
        if (!pl.keys[i] || bind(pl.keys[i], t, out->err, sizeof out->err)) { plan_free(&pl); return 1; }
    }
    for (uint32_t i = 0; i < naggs; i++) {
        if (parse_aggregate(aggs[i], &pl.aggs[i], out->err, sizeof out->err) ||
            bind(pl.aggs[i].operand, t, out->err, sizeof out->err)) {
            plan_free(&pl);
            return 1;
        }
    }
    out->nkeys = nkeys;
    out->naggs = naggs;
    double t0 = now_s();

    if (!has_group) {
        /* Filter alone: the rows sendItem() forwards (execution/filter.go:56-57), in input order */
        ectx cx = {t, 0};
        uint64_t cap = 1024, n = 0;
        uint64_t *sel = malloc(cap * 8);
        for (uint64_t row = 0; row < t->nrows; row++) {
            if (pl.cond) {
                val v = eval(pl.cond, row, &cx);
                if (!truth(v, &cx.unsupported)) continue;
            }
            if (n == cap) {
                cap *= 2;
                sel = realloc(sel, cap * 8);
            }
            sel[n++] = row;
        }
        out->nselected = n;
        out->selected = sel;
        out->rows_filtered_in = n;
        out->seconds = now_s() - t0;
        plan_free(&pl);
        if (cx.unsupported) {
            snprintf(out->err, sizeof out->err, "value outside the restated subset met at run time");
            return 2;
        }
        return 0;
    }

    if (threads < 1) threads = 1;
    worker *ws = calloc((size_t)threads, sizeof(worker));
    uint64_t next_chunk = 0;
    for (int i = 0; i < threads; i++) {
        ws[i].pl = &pl;
        ws[i].next_chunk = &next_chunk;
        ws[i].chunk_rows = threads == 1 ? (t->nrows ? t->nrows : 1) : 4096;
    }
    if (threads == 1) run_worker(&ws[0]);
    else {
        for (int i = 0; i < threads; i++) pthread_create(&ws[i].th, NULL, run_worker, &ws[i]);
        for (int i = 0; i < threads; i++) pthread_join(ws[i].th, NULL);
    }

    /* IntermediateGroup (serial): execution/group_intermediate.go:56-104 — the first partial met for a key is
     * kept, later ones are merged with CumulateIntermediate.  InitialGroup.afterItems emits in map order
     * (random in Go); worker order is used here. */
    ectx cx = {t, 0};
    gmap inter;
    memset(&inter, 0, sizeof inter);
    for (int i = 0; i < threads; i++) {
        cx.unsupported |= ws[i].unsupported;
        out->rows_filtered_in += ws[i].rows_passed;
        for (size_t k = 0; k < ws[i].groups.n; k++) {
            group *g = ws[i].groups.list[k];
            group *c = gmap_find(&inter, g->key, g->keylen, g->hash);
            if (!c) {
                gmap_add(&inter, g);
            } else {
                for (uint32_t a = 0; a < naggs; a++) agg_cumulate_intermediate(&pl.aggs[a], &g->aggs[a], &c->aggs[a], &cx);
                if (g->first_row < c->first_row) c->first_row = g->first_row;
                group_free(g, naggs);
            }
        }
        gmap_free(&ws[i].groups, naggs, 0);
    }
    free(ws);

    /* FinalGroup: execution/group_final.go:55-118 — ComputeFinal per aggregate; with no keys and no input one
     * row of Default() values is emitted (:108-117). */
    int bad = 0;
    strbuf extra;
    memset(&extra, 0, sizeof extra);
    uint64_t *extra_off = NULL;
    uint32_t extra_n = 0, extra_cap = 0;
    uint64_t ng = inter.n;
    int default_row = (nkeys == 0 && ng == 0);
    uint64_t nout = default_row ? 1 : ng;
    out->ngroups = nout;
    out->keys = calloc(nout * (nkeys ? nkeys : 1), sizeof(n1k_value));
    out->aggs = calloc(nout * (naggs ? naggs : 1), sizeof(n1k_value));
    if (default_row) {
        for (uint32_t a = 0; a < naggs; a++) {
            aggstate st;
            agg_default(&pl.aggs[a], &st);
            out_value(st.cum, &out->aggs[a], &bad);
        }
    } else {
        for (uint64_t gi = 0; gi < ng; gi++) {
            group *g = inter.list[gi];
            for (uint32_t k = 0; k < nkeys; k++) out_value(g->keyvals[k], &out->keys[gi * nkeys + k], &bad);
            for (uint32_t a = 0; a < naggs; a++) {
                if (pl.aggs[a].kind == A_ARRAY) {
                    array_agg_final(&pl.aggs[a], &g->aggs[a], t, &cx, &extra, &extra_off, &extra_n, &extra_cap,
                                    &out->aggs[gi * naggs + a]);
                    continue;
                }
                out_value(agg_compute_final(&pl.aggs[a], &g->aggs[a]), &out->aggs[gi * naggs + a], &bad);
            }
        }
    }
    out->extra_bytes = extra.p;
    out->extra_offsets = extra_off;
    out->extra_n = extra_n;
    out->seconds = now_s() - t0;
    gmap_free(&inter, naggs, 1);
    plan_free(&pl);
    if (cx.unsupported || bad) {
        snprintf(out->err, sizeof out->err, "value outside the restated subset met at run time");
        return 2;
    }
    return 0;
}

void n1o_free_result(n1o_result *r) {
    free(r->keys);
    free(r->aggs);
    free(r->selected);
    free(r->extra_bytes);
    free(r->extra_offsets);
    r->extra_bytes = NULL;
    r->extra_offsets = NULL;
    r->extra_n = 0;
    r->keys = r->aggs = NULL;
    r->selected = NULL;
}

int n1o_eval(const char *expr, const n1o_table *t, n1k_value *outv, char *err, size_t errlen) {
    node *n = parse_full(expr, err, errlen);
    if (!n) return 1;
    if (bind(n, t, err, errlen)) {
        free_node(n);
        return 1;
    }
    ectx cx = {t, 0};
    int bad = 0;
    for (uint64_t r = 0; r < t->nrows; r++) {
        val v = eval(n, r, &cx);
        if ((v.type == TY_STRING) && v.code == UINT32_MAX) {
            /* constant string result (GREATEST / LEAST may return one): its code if the table's dictionary holds the same
             * bytes, else the tag only */
            memset(&outv[r], 0, sizeof outv[r]);
            outv[r].tag = N1K_T_STRING;
            outv[r].v.code = UINT32_MAX;
            for (uint32_t c = 0; c < t->dict_n; c++)
                if (t->dict_offsets[c + 1] - t->dict_offsets[c] == v.slen && !memcmp(t->dict_bytes + t->dict_offsets[c], v.s, v.slen)) {
                    outv[r].v.code = c;
                    break;
                }
        } else
            out_value(v, &outv[r], &bad);
    }
    free_node(n);
    if (cx.unsupported || bad) {
        snprintf(err, errlen, "value outside the restated subset met at run time");
        return 2;
    }
    return 0;
}

/* keyspace.Count(): datastore/file/file.go:296-302 — len(ioutil.ReadDir(path)) */
int64_t n1o_count_scan(const char *dir) {
    DIR *d = opendir(dir);
    if (!d) return -1;
    int64_t n = 0;
    struct dirent *e;
    while ((e = readdir(d))) {
        if (!strcmp(e->d_name, ".") || !strcmp(e->d_name, "..")) continue;
        n++;
    }
    closedir(d);
    return n;
}
