// n1k_device.h — device-side value semantics of the N1QL hot path (gfx950).
//
// Tagged scalars (tag + 64-bit payload) with the reference's collation,
// equality, truth and 4-valued logic rules.  Each helper cites the Go source
// it reproduces (paths relative to the reference tree).
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif
#include "n1k_types.h"

namespace n1k {

#define N1K_DEV __device__ __forceinline__

// type-order class of a tag (value/value.go:69-79):
// MISSING 0, NULL 1, BOOLEAN 2, NUMBER 3, STRING 4, ARRAY 5, OBJECT 6
N1K_DEV uint32_t cls_of(uint32_t tag) { return (uint32_t)((0x654332210ull >> (tag * 4u)) & 0xFu); }

N1K_DEV double as_f64(uint64_t bits) { return __longlong_as_double((long long)bits); }
N1K_DEV uint64_t f64_bits(double d) { return (uint64_t)__double_as_longlong(d); }

// Go float64 -> int64 on amd64 (cvttsd2si): out of range / NaN -> MinInt64
N1K_DEV int64_t go_f2i(double f) {
    if (!(f >= -9223372036854775808.0 && f < 9223372036854775808.0)) return INT64_MIN;
    return (int64_t)f;
}
// value.IsInt (value/integer.go:354-356)
N1K_DEV bool is_int_f64(double x) { return x == (double)go_f2i(x); }

// monotone u64 image of a double; NaN -> 0 ("NaN sorts first", value/float.go:123-172)
N1K_DEV uint64_t f64_sortable(double d) {
    if (d != d) return 0ull;
    uint64_t b = f64_bits(d);
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}
N1K_DEV double f64_unsortable(uint64_t s) {
    if (s == 0ull) return __longlong_as_double(0x7FF8000000000000ll);
    uint64_t b = (s & 0x8000000000000000ull) ? (s & 0x7FFFFFFFFFFFFFFFull) : ~s;
    return as_f64(b);
}

// collateFloat (value/float.go:123-172)
N1K_DEV int collate_f64(double t, double o) {
    if (t != t) return (o != o) ? 0 : -1;
    if (o != o) return 1;
    return t < o ? -1 : (t > o ? 1 : 0);
}

N1K_DEV double num_actual(uint32_t tag, uint64_t p) { return tag == T_INT ? (double)(int64_t)p : as_f64(p); }

// result of X.Compare(Y): -1/0/1, or these two
constexpr int CMP_NULL = 2, CMP_MISSING = 3;

// X.Collate(Y) for two non-MISSING/NULL values of any type
// (value/integer.go:100-118, float.go:106-121, string.go:116-130, boolean.go:99-113)
N1K_DEV int collate(uint32_t ta, uint64_t pa, uint32_t tb, uint64_t pb, const uint32_t* rank, uint32_t* unsupported) {
    uint32_t ca = cls_of(ta), cb = cls_of(tb);
    if (ca != cb) return ca < cb ? -1 : 1;
    if (ca == 3) {
        if (ta == T_INT && tb == T_INT) {
            int64_t x = (int64_t)pa, y = (int64_t)pb;
            return x < y ? -1 : (x > y ? 1 : 0);
        }
        return collate_f64(num_actual(ta, pa), num_actual(tb, pb));
    }
    if (ca == 2) return ta < tb ? -1 : (ta > tb ? 1 : 0);
    if (ca == 4) {
        if (pa == pb) return 0;
        uint32_t ra = rank[(uint32_t)pa], rb = rank[(uint32_t)pb];
        return ra < rb ? -1 : 1;
    }
    if (ca >= 5) {
        if (pa == pb) return 0;
        *unsupported = 1;  // element-wise ordering of arrays/objects is outside the device subset
        return 0;
    }
    return 0;
}

// X.Compare(Y): MISSING if either is MISSING, else NULL if either is NULL, else Collate
N1K_DEV int compare(uint32_t ta, uint64_t pa, uint32_t tb, uint64_t pb, const uint32_t* rank, uint32_t* unsupported) {
    if (ta == T_MISSING || tb == T_MISSING) return CMP_MISSING;
    if (ta == T_NULL || tb == T_NULL) return CMP_NULL;
    return collate(ta, pa, tb, pb, rank, unsupported);
}

// X.Equals(Y) as a 4-valued logic value
// (value/integer.go:68-87, float.go:74-93, string.go:82-96, boolean.go:74-88, null.go:72-80, missing.go:88-90)
N1K_DEV uint32_t equals_l(uint32_t ta, uint64_t pa, uint32_t tb, uint64_t pb, uint32_t* unsupported) {
    if (ta == T_MISSING || tb == T_MISSING) return L_MISSING;
    if (ta == T_NULL || tb == T_NULL) return L_NULL;
    uint32_t ca = cls_of(ta), cb = cls_of(tb);
    if (ca != cb) return L_FALSE;
    if (ca == 3) {
        if (ta == T_INT && tb == T_INT) return pa == pb ? L_TRUE : L_FALSE;
        return num_actual(ta, pa) == num_actual(tb, pb) ? L_TRUE : L_FALSE;
    }
    if (ca == 2) return ta == tb ? L_TRUE : L_FALSE;
    if (ca == 4) return pa == pb ? L_TRUE : L_FALSE;
    if (pa == pb) return L_TRUE;
    *unsupported = 1;
    return L_FALSE;
}

// type + Truth() of a value used as a condition
// (value/integer.go:136-138, float.go:190-192, string.go:148, boolean.go:125, null.go:106, missing.go:113)
N1K_DEV uint32_t truth_l(uint32_t tag, uint64_t p, uint32_t empty_str, uint32_t empty_arr, uint32_t empty_obj) {
    switch (tag) {
        case T_MISSING: return L_MISSING;
        case T_NULL: return L_NULL;
        case T_FALSE: return L_FALSE;
        case T_TRUE: return L_TRUE;
        case T_INT: return p != 0 ? L_TRUE : L_FALSE;
        case T_FLOAT: {
            double d = as_f64(p);
            return (d == d && d != 0.0) ? L_TRUE : L_FALSE;
        }
        case T_STRING: return (uint32_t)p != empty_str ? L_TRUE : L_FALSE;
        case T_ARRAY: return (uint32_t)p != empty_arr ? L_TRUE : L_FALSE;
        default: return (uint32_t)p != empty_obj ? L_TRUE : L_FALSE;
    }
}

// And.Apply over `n` stacked logic values (expression/logic_and.go:64-89)
N1K_DEV uint32_t logic_and(uint64_t& st, uint32_t n) {
    bool any_false = false, any_missing = false, any_null = false;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t v = (uint32_t)(st & 3ull);
        st >>= 2;
        any_false |= (v == L_FALSE);
        any_missing |= (v == L_MISSING);
        any_null |= (v == L_NULL);
    }
    return any_false ? L_FALSE : (any_missing ? L_MISSING : (any_null ? L_NULL : L_TRUE));
}
// Or.Apply (expression/logic_or.go:98-123)
N1K_DEV uint32_t logic_or(uint64_t& st, uint32_t n) {
    bool any_true = false, any_missing = false, any_null = false;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t v = (uint32_t)(st & 3ull);
        st >>= 2;
        any_true |= (v == L_TRUE);
        any_missing |= (v == L_MISSING);
        any_null |= (v == L_NULL);
    }
    return any_true ? L_TRUE : (any_null ? L_NULL : (any_missing ? L_MISSING : L_FALSE));
}
// Not.Apply (expression/logic_not.go:57-69)
N1K_DEV uint32_t logic_not(uint32_t v) { return v >= L_NULL ? v : (v ^ 1u); }

// ---- NumberValue arithmetic (value/integer.go:266-352, value/float.go:331-385); a number is (tag INT|FLOAT, payload)
struct Num {
    uint32_t tag;
    uint64_t p;
};
N1K_DEV Num num_int(int64_t v) { return Num{T_INT, (uint64_t)v}; }
N1K_DEV Num num_flt(double v) { return Num{T_FLOAT, f64_bits(v)}; }
// value.NewValue(float64): integral values fold to int64 (value/value.go:377-382)
N1K_DEV Num num_new_value(double d) { return is_int_f64(d) ? num_int(go_f2i(d)) : num_flt(d); }
// intValue.Add keeps int64 only for same-sign operands without overflow (value/integer.go:266-277)
N1K_DEV Num num_add(Num a, Num b) {
    if (a.tag == T_INT && b.tag == T_INT) {
        int64_t x = (int64_t)a.p, y = (int64_t)b.p;
        int64_t rv = (int64_t)((uint64_t)x + (uint64_t)y);
        if ((x >= 0 && y >= 0 && rv >= 0) || (x < 0 && y < 0 && rv < 0)) return num_int(rv);
        return num_flt((double)x + (double)y);
    }
    return num_flt(num_actual(a.tag, a.p) + num_actual(b.tag, b.p));
}
N1K_DEV Num num_neg(Num a) {  // value/integer.go:331-335
    if (a.tag == T_FLOAT) return num_flt(-as_f64(a.p));
    int64_t x = (int64_t)a.p;
    if (x == INT64_MIN) return num_flt(-(double)x);
    return num_int(-x);
}
N1K_DEV Num num_sub(Num a, Num b) {  // value/integer.go:337-346
    if (a.tag == T_INT && b.tag == T_INT && (int64_t)b.p > INT64_MIN) return num_add(a, num_int(-(int64_t)b.p));
    return num_flt(num_actual(a.tag, a.p) - num_actual(b.tag, b.p));
}
N1K_DEV Num num_mult(Num a, Num b) {  // value/integer.go:318-329
    if (a.tag == T_INT && b.tag == T_INT) {
        int64_t x = (int64_t)a.p, y = (int64_t)b.p;
        int64_t rv = (int64_t)((uint64_t)x * (uint64_t)y);
        // Go keeps the int64 when `x == 0 || rv / x == y`.  For a wrapped product that holds exactly when the 128-bit
        // product fits (rv = x * y - k * 2^64 and rv = x * y + r with |r| < |x| force k = 0), plus the one case where Go's
        // own division wraps: x == -1, y == MinInt64 (MinInt64 / -1 == MinInt64).  The high half of the product decides
        // it without a 64-bit division (~100 instructions per row on the device).
        const bool fits = __mul64hi((long long)x, (long long)y) == (long long)(rv >> 63);
        if (fits || (x == -1 && y == INT64_MIN)) return num_int(rv);
        return num_flt((double)x * (double)y);
    }
    return num_flt(num_actual(a.tag, a.p) * num_actual(b.tag, b.p));
}
// IDiv / IMod (value/integer.go:279-316, value/float.go:335-367): NULL (tag T_NULL) on a zero divisor
N1K_DEV Num num_idiv_imod(Num a, Num b, bool mod) {
    int64_t x = a.tag == T_FLOAT ? go_f2i(as_f64(a.p)) : (int64_t)a.p;
    int64_t d;
    if (b.tag == T_INT) d = (int64_t)b.p;
    else {
        if (as_f64(b.p) == 0.0) return Num{T_NULL, 0};
        d = go_f2i(as_f64(b.p));
    }
    if (d == 0) return Num{T_NULL, 0};
    if (x == INT64_MIN && d == -1) return num_int(mod ? 0 : INT64_MIN);
    return num_int(mod ? x % d : x / d);
}

// math.Pow(10, n) for the digit counts ROUND / TRUNC see: exact powers of ten up to 1e22, their correctly rounded
// reciprocals for negative n (Go's Pow inverts the positive power), the library pow beyond
N1K_DEV double pow10_go(int n) {
    const int a = n < 0 ? -n : n;
    if (a > 22) return pow(10.0, (double)n);
    double p = 1.0;
    for (int i = 0; i < a; i++) p *= 10.0;
    return n < 0 ? 1.0 / p : p;
}
// roundFloat (expression/func_num.go:1715-1736): half away from zero, except that an exact .5 goes to the even neighbour
N1K_DEV double round_float(double x, int prec) {
    if (x != x || x == __longlong_as_double(0x7FF0000000000000ll) || x == __longlong_as_double((long long)0xFFF0000000000000ull)) return x;
    double sign = 1.0;
    if (x < 0) { sign = -1.0; x = -x; }
    const double pw = pow10_go(prec);
    const double intermed = x * pw + 0.5;
    double rounder = floor(intermed);
    if (rounder == intermed && fmod(rounder, 2.0) != 0.0) rounder -= 1.0;
    return sign * rounder / pw;
}

// One arithmetic node over up to four operand values (shared by the element-wise arith_kernel and by the plan-specialised
// scan, which evaluates the nodes in registers).  Add/Mult.Apply (expression/arith_add.go:51-70, arith_mult.go:51-70): any
// MISSING operand -> MISSING, else any non-number -> NULL, else the fold from int 0 / int 1.  Sub/Div/Mod/IDiv/IMod/Neg:
// arith_sub.go:53-61, arith_div.go:46-64, arith_mod.go:48-66, arith_idiv.go:46-56, arith_imod.go, arith_neg.go:51-59.
N1K_DEV void arith_apply(uint32_t op, uint32_t nops, const uint32_t (&tg)[4], const uint64_t (&pv)[4], uint32_t& rt_out, uint64_t& rp_out) {
    auto is_num = [](uint32_t t) { return t == T_INT || t == T_FLOAT; };
    uint32_t rt = T_NULL;
    uint64_t rp = 0;
    if (op == AR_ADD || op == AR_MULT) {
        bool null = false, missing = false;
        Num acc = num_int(op == AR_ADD ? 0 : 1);
        for (uint32_t k = 0; k < nops; k++) {
            if (!null && is_num(tg[k])) {
                if (k == 0) {
                    // the fold's first step, from the identity (Add.Apply starts at int 0, Mult.Apply at int 1), spelled out:
                    // 1 * x is x (an int stays the same int, 1.0 * f is f bit for bit); 0 + x is x for a non-negative int,
                    // float64(x) for a negative one (the same-sign rule of intValue.Add) and 0.0 + f for a float
                    if (op == AR_MULT) acc = Num{tg[0], pv[0]};
                    else if (tg[0] == T_INT) acc = (int64_t)pv[0] >= 0 ? num_int((int64_t)pv[0]) : num_flt((double)(int64_t)pv[0]);
                    else acc = num_flt(0.0 + as_f64(pv[0]));
                } else
                    acc = op == AR_ADD ? num_add(acc, Num{tg[k], pv[k]}) : num_mult(acc, Num{tg[k], pv[k]});
            } else if (tg[k] == T_MISSING) missing = true;
            else null = true;
        }
        if (missing) rt = T_MISSING;
        else if (null) rt = T_NULL;
        else { rt = acc.tag; rp = acc.p; }
    } else if (op == AR_NEG) {
        if (is_num(tg[0])) { Num r = num_neg(Num{tg[0], pv[0]}); rt = r.tag; rp = r.p; }
        else rt = tg[0] == T_MISSING ? (uint32_t)T_MISSING : (uint32_t)T_NULL;
    } else if (op >= AR_ROUND) {
        // expression/func_num.go: the argument goes through float64 (intValue.Actual() is float64(this),
        // value/integer.go:57-59) and the result through value.NewValue (integral results fold back to int)
        if (tg[0] == T_MISSING) rt = T_MISSING;
        else if (is_num(tg[0])) {
            const double v = num_actual(tg[0], pv[0]);
            int prec = 0;
            bool ok = true;
            if ((op == AR_ROUND || op == AR_TRUNC) && nops == 2) {  // Round.Apply / Trunc.Apply: the digits argument
                if (tg[1] == T_MISSING) { rt = T_MISSING; ok = false; }
                else if (!is_num(tg[1])) ok = false;  // NULL
                else {
                    const double pf = num_actual(tg[1], pv[1]);
                    if (pf != trunc(pf)) ok = false;  // NULL
                    else prec = pf > 400.0 ? 400 : (pf < -400.0 ? -400 : (int)pf);
                }
            }
            if (ok) {
                double r;
                switch (op) {
                    case AR_ROUND: r = round_float(v, prec); break;
                    case AR_TRUNC: { const double pw = pow10_go(prec); r = trunc(v * pw) / pw; break; }  // truncateFloat
                    case AR_ABS: r = fabs(v); break;
                    case AR_CEIL: r = ceil(v); break;
                    case AR_FLOOR: r = floor(v); break;
                    case AR_SIGN: r = v < 0.0 ? -1.0 : (v > 0.0 ? 1.0 : 0.0); break;
                    default: r = sqrt(v); break;
                }
                Num n = num_new_value(r);
                rt = n.tag;
                rp = n.p;
            }
        }
    } else {
        bool both = is_num(tg[0]) && is_num(tg[1]);
        if (tg[0] == T_MISSING || tg[1] == T_MISSING) rt = T_MISSING;
        else if (op == AR_SUB) {
            if (both) { Num r = num_sub(Num{tg[0], pv[0]}, Num{tg[1], pv[1]}); rt = r.tag; rp = r.p; }
        } else if (op == AR_DIV || op == AR_MOD) {
            if (is_num(tg[1])) {
                double d = num_actual(tg[1], pv[1]);
                if (d != 0.0 && is_num(tg[0])) {
                    double x = num_actual(tg[0], pv[0]);
                    Num r = num_new_value(op == AR_DIV ? x / d : fmod(x, d));
                    rt = r.tag;
                    rp = r.p;
                }
            }
        } else if (both) {  // IDIV / IMOD
            Num r = num_idiv_imod(Num{tg[0], pv[0]}, Num{tg[1], pv[1]}, op == AR_IMOD);
            rt = r.tag;
            rp = r.p;
        }
    }
    rt_out = rt;
    rp_out = rp;
}

N1K_DEV uint64_t mix64(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}

// 32-bit hash of a packed key / member word for the radix partitions (its bytes, from the top, are the digits of the
// passes): a fold of the two halves and murmur3's 32-bit finalizer — three 32-bit multiplies where mix64 needs the
// equivalent of eight (integer multiplies issue at a quarter of the VALU rate, and every row is hashed in every pass)
N1K_DEV uint32_t part_hash(uint64_t x) {
    uint32_t h = (uint32_t)x ^ ((uint32_t)(x >> 32) * 0x9E3779B1u);
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}

N1K_DEV uint64_t zigzag(int64_t x) { return ((uint64_t)x << 1) ^ (uint64_t)(x >> 63); }
N1K_DEV int64_t unzigzag(uint64_t z) { return (int64_t)(z >> 1) ^ -(int64_t)(z & 1); }

// One group-key value -> its bit field (see KeySpec).  Equal fields <=> equal canonical JSON of the
// key (execution/group_util.go:18-35): MISSING and NULL are distinct, FALSE/TRUE, numbers by value
// (integral floats fold to the int they equal: value/float.go:31-48 prints 5.0 as "5"), strings /
// arrays / objects by dictionary code of their (canonical) text.
// Code of a wide key value: its slot in the handle's value table (find-or-insert, single-word CAS).  The codes
// are local to the handle (they depend on insertion order), so packed keys holding them never leave the device:
// finalize_kernel turns them back into values, and the multi-GPU exchange hashes the value itself.
N1K_DEV bool wide_code(uint64_t* tab, uint32_t bits, unsigned long long* count, uint64_t payload, uint64_t& code) {
    if (!tab) return false;
    const uint64_t mask = (1ull << bits) - 1ull;
    uint64_t h = mix64(payload) & mask;
    for (uint64_t probe = 0; probe <= mask; probe++) {
        unsigned long long cur = __hip_atomic_load((unsigned long long*)&tab[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == kEmptyKey) {
            cur = atomicCAS((unsigned long long*)&tab[h], (unsigned long long)kEmptyKey, (unsigned long long)payload);
            if (cur == kEmptyKey) {
                atomicAdd(count, 1ull);
                cur = payload;
            }
        }
        if (cur == payload) {
            code = h;
            return true;
        }
        h = (h + 1ull) & mask;
    }
    return false;
}

// The INT whose decimal text is strconv.FormatFloat(d, 'f', -1, 64) for an integral |d| in [2^53, 2^63): the shortest run of
// significant digits that parses back to d, then zeros (value/float.go:31-48 writes group keys with it, so the float and that
// INT are one group).  false: the text does not fit an int64 (the float then keeps a key of its own).
N1K_DEV bool float_key_text_int(double d, int64_t& out) {
    const double a = d < 0 ? -d : d;
    if (!(a >= 9007199254740992.0 && a < 9223372036854775808.0)) return false;
    const unsigned long long v = (unsigned long long)a;  // exact: an integer below 2^63
    int n = 1;
    for (unsigned long long t = v; t >= 10ull; t /= 10ull) n++;  // 16 .. 19 digits
    for (int p = 1; p <= 17 && p <= n; p++) {
        unsigned long long scale = 1;
        for (int i = 0; i < n - p; i++) scale *= 10ull;
        const unsigned long long lo = v / scale * scale, hi = lo + scale;  // the two p-digit decimals around v
        const bool rlo = (double)lo == a, rhi = (double)hi == a;
        if (!rlo && !rhi) continue;
        const unsigned long long c = rlo && rhi ? (v - lo <= hi - v ? lo : hi) : (rlo ? lo : hi);
        if (c > 9223372036854775807ull) return false;
        out = d < 0 ? -(int64_t)c : (int64_t)c;
        return true;
    }
    return false;
}

// field classes of a TAGGED key: 0 MISSING 1 NULL 2 FALSE 3 TRUE 4 INT (zigzag in the field) 5 STRING 6 ARRAY
// 7 OBJECT (dictionary code) 8 wide INT 9 FLOAT (code of the value table).  `canon` is a handle-independent
// image of the field (what the multi-GPU row exchange hashes).
N1K_DEV bool pack_key_field(const Program& P, const KeySpec& k, uint32_t tag, uint64_t p, uint64_t& field, uint64_t& canon) {
    if (k.mode == KEYM_DICT) {
        // DICT32 columns / pure string keys: 0 MISSING, 1 NULL, code+2
        if (tag == T_MISSING) field = 0;
        else if (tag == T_NULL) field = 1;
        else if (tag == T_STRING) field = p + 2;
        else return false;
        canon = field;
        return k.bits >= 64 || field < (1ull << k.bits);
    }
    const uint32_t sb = k.bits - 4;
    uint64_t cls, sub = 0;
    bool num = false;
    int64_t iv = 0;
    switch (tag) {
        case T_MISSING: cls = 0; break;
        case T_NULL: cls = 1; break;
        case T_FALSE: cls = 2; break;
        case T_TRUE: cls = 3; break;
        case T_INT: cls = 4; num = true; iv = (int64_t)p; break;
        case T_FLOAT: {
            double d = as_f64(p);
            // 5.0 and 5 are one group: both print "5" (value/float.go:31-48).  From 2^53 up FormatFloat's shortest
            // digits are no longer the integer's own ("-9223372036854776000" for -2^63), so such floats stay floats.
            int64_t shortest;
            if (is_int_f64(d) && d > -9007199254740992.0 && d < 9007199254740992.0) {
                cls = 4;
                num = true;
                iv = go_f2i(d);
            } else if (is_int_f64(d) && float_key_text_int(d, shortest)) {
                // From 2^53 up the key's text is FormatFloat's SHORTEST digits followed by zeros — float 2^60 prints
                // "1152921504606847000" — and that text is the map key: the float groups with the INT of that text.
                cls = 4;
                num = true;
                iv = shortest;
            } else if (d != d || d == __longlong_as_double(0x7FF0000000000000ll) || d == __longlong_as_double((long long)0xFFF0000000000000ull)) {
                // NaN / +Inf / -Inf marshal as JSON STRINGS (value/float.go:31-48): one group with the string of that text
                cls = 5;
                sub = d != d ? P.nan_code : (d > 0 ? P.pinf_code : P.ninf_code);
            } else {
                cls = 9;
                uint64_t bits = d != d ? 0x7FF8000000000000ull : p;  // one NaN
                canon = mix64(bits) ^ 9ull;
                if (P.wide_bits > sb || !wide_code(P.wide_flt, P.wide_bits, P.wide_count, bits, sub)) return false;
            }
            break;
        }
        case T_STRING: cls = 5; sub = p; break;
        case T_ARRAY: cls = 6; sub = p; break;
        default: cls = 7; sub = p; break;
    }
    if (num) {
        sub = zigzag(iv);
        if (sb < 64 && sub >= (1ull << sb)) {
            cls = 8;
            canon = mix64((uint64_t)iv) ^ 8ull;
            if (P.wide_bits > sb || !wide_code(P.wide_int, P.wide_bits, P.wide_count, (uint64_t)iv, sub)) return false;
        }
    }
    if (sb < 64 && sub >= (1ull << sb)) return false;
    field = (sub << 4) | cls;
    if (cls < 8) canon = field;
    return true;
}

N1K_DEV void unpack_key_field(const Program& P, uint32_t mode, uint64_t field, uint32_t& tag, uint64_t& p) {
    if (mode == KEYM_DICT) {
        if (field == 0) { tag = T_MISSING; p = 0; }
        else if (field == 1) { tag = T_NULL; p = 0; }
        else { tag = T_STRING; p = field - 2; }
        return;
    }
    uint64_t cls = field & 15ull, sub = field >> 4;
    p = 0;
    switch (cls) {
        case 0: tag = T_MISSING; break;
        case 1: tag = T_NULL; break;
        case 2: tag = T_FALSE; break;
        case 3: tag = T_TRUE; break;
        case 4: tag = T_INT; p = (uint64_t)unzigzag(sub); break;
        case 5: tag = T_STRING; p = sub; break;
        case 6: tag = T_ARRAY; p = sub; break;
        case 7: tag = T_OBJECT; p = sub; break;
        case 8: tag = T_INT; p = P.wide_int[sub]; break;
        default: tag = T_FLOAT; p = P.wide_flt[sub]; break;
    }
}

}  // namespace n1k
