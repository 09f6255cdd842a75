// n1k_tables.h — the workgroup (LDS) and global group tables: probing, accumulators, merges.
//
// Shared by the ahead-of-time kernels (n1k_kernels.hip) and by kernels compiled at run time for one plan shape
// (n1k_jit.cpp through hiprtc), so it must stay a plain header.
#pragma once
#include "n1k_device.h"

namespace n1k {

// LDS is addressed through explicit address-space-3 pointers and workgroup-scope atomics so that every access
// is a ds_* instruction (generic pointers make hipcc fall back to flat_* loads for volatile reads).
typedef __attribute__((address_space(3))) unsigned long long lds_u64;
typedef __attribute__((address_space(3))) long long lds_i64;
typedef __attribute__((address_space(3))) double lds_f64;
typedef __attribute__((address_space(3))) unsigned int lds_u32;

N1K_DEV lds_u64* lds_word(uint64_t* lds, uint32_t index) { return (lds_u64*)lds + index; }
N1K_DEV unsigned long long lds_peek(lds_u64* p) { return *(volatile lds_u64*)p; }
N1K_DEV void lds_add_u64(lds_u64* p, unsigned long long v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
N1K_DEV void lds_add_f64(lds_u64* p, double v) { (void)__hip_atomic_fetch_add((lds_f64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
N1K_DEV void lds_or_u64(lds_u64* p, unsigned long long v) { (void)__hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
N1K_DEV void lds_min_u64(lds_u64* p, unsigned long long v) { (void)__hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
N1K_DEV void lds_max_u64(lds_u64* p, unsigned long long v) { (void)__hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
N1K_DEV void lds_min_i64(lds_u64* p, long long v) { (void)__hip_atomic_fetch_min((lds_i64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
N1K_DEV void lds_max_i64(lds_u64* p, long long v) { (void)__hip_atomic_fetch_max((lds_i64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// ------------------------------------------------------------------ hash tables

// The workgroup's LDS table is word-major: word w of slot s lives at lds[w * S + s] (consecutive slots fall
// into consecutive banks).  Word 0 is the packed key (HASH mode) or a "touched" marker (DIRECT mode).
N1K_DEV uint32_t lds_hash(uint64_t key, uint32_t S) {
    uint32_t x = (uint32_t)key ^ ((uint32_t)(key >> 32) * 0x9E3779B1u);
    x *= 0x85EBCA6Bu;
    x ^= x >> 15;
    x *= 0xC2B2AE35u;
    return __umulhi(x, S);  // S need not be a power of two
}

N1K_DEV int lds_find_or_insert(uint64_t* lds, uint32_t S, uint64_t key, uint32_t* fill, uint32_t max_fill) {
    uint32_t h = lds_hash(key, S);
    lds_u32* fillp = (lds_u32*)fill;
#pragma unroll 1  // (unrolled, 32 copies of the probe per record were most of the kernels' code)
    for (int probe = 0; probe < 32; probe++) {
        lds_u64* p = lds_word(lds, h);
        unsigned long long cur = lds_peek(p);
        if (cur == key) return (int)h;
        if (cur == kEmptyKey) {
            if (*(volatile lds_u32*)fillp >= max_fill) return -1;  // table is kept sparse: new keys bypass LDS
            unsigned long long expected = kEmptyKey;
            bool won = __hip_atomic_compare_exchange_strong(p, &expected, (unsigned long long)key, __ATOMIC_RELAXED,
                                                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (won) {
                (void)__hip_atomic_fetch_add(fillp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                return (int)h;
            }
            if (expected == key) return (int)h;
        }
        h = h + 1 == S ? 0 : h + 1;
    }
    return -1;
}

// global open-addressed table (keys never change once set, so a stale cached EMPTY only costs a CAS)
N1K_DEV long long global_find_or_insert(const GlobalTable& G, uint64_t key, uint32_t* err_flags,
                                         unsigned long long* ngroups) {
    uint64_t mask = G.capacity - 1;
    uint64_t h = mix64(key) & mask;
    for (int probe = 0; probe < 8192; probe++) {
        unsigned long long cur = __hip_atomic_load((unsigned long long*)&G.keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == key) return (long long)h;
        if (cur == kEmptyKey) {
            unsigned long long old = atomicCAS((unsigned long long*)&G.keys[h], (unsigned long long)kEmptyKey, (unsigned long long)key);
            if (old == kEmptyKey) {
                // one atomic on the group counter per wave, not per new group: the lanes that are here together count
                // themselves (millions of same-address atomics cost ~10 ns each: 6.4 M new groups were 40+ ms)
                const unsigned long long active = __ballot(1);
                if (__builtin_amdgcn_mbcnt_hi((uint32_t)(active >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)active, 0u)) == 0u)
                    atomicAdd(ngroups, (unsigned long long)__popcll(active));
                return (long long)h;
            }
            if (old == key) return (long long)h;
        }
        h = (h + 1) & mask;
    }
    atomicOr(err_flags, (uint32_t)ERR_TABLE_FULL);
    return -1;
}

// the same without touching the group counter: the caller adds up `fresh` and reports once per workgroup
N1K_DEV long long global_find_or_insert_quiet(const GlobalTable& G, uint64_t key, uint32_t* err_flags, bool& fresh) {
    uint64_t mask = G.capacity - 1;
    uint64_t h = mix64(key) & mask;
    for (int probe = 0; probe < 8192; probe++) {
        unsigned long long cur = __hip_atomic_load((unsigned long long*)&G.keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == key) return (long long)h;
        if (cur == kEmptyKey) {
            unsigned long long old = atomicCAS((unsigned long long*)&G.keys[h], (unsigned long long)kEmptyKey, (unsigned long long)key);
            if (old == kEmptyKey) {
                fresh = true;
                return (long long)h;
            }
            if (old == key) return (long long)h;
        }
        h = (h + 1) & mask;
    }
    atomicOr(err_flags, (uint32_t)ERR_TABLE_FULL);
    return -1;
}

// ------------------------------------------------------------------ accumulators

template <int BLOCK>
N1K_DEV void lds_table_init(const Program& P, uint64_t* lds, uint32_t S, uint32_t tid) {
    for (uint32_t s = tid; s < S; s += BLOCK) lds[s] = kEmptyKey;
    if (P.want_rep_row)
        for (uint32_t s = tid; s < S; s += BLOCK) lds[(size_t)P.rep_lds_word * S + s] = ~0ull;
    for (uint32_t a = 0; a < P.naggs; a++) {
        const AggSpec& ag = P.aggs[a];
        uint64_t* w = lds + (size_t)ag.lds_off * S;
        if (ag.distinct) {
            for (uint32_t i = 0; i < ag.lds_n; i++)
                for (uint32_t s = tid; s < S; s += BLOCK) w[(size_t)i * S + s] = 0;
            continue;
        }
        uint32_t nw = (ag.kind == AGG_COUNT || ag.kind == AGG_COUNTN) ? 1u : (ag.kind == AGG_SUM ? kLdsWordsSum : (ag.kind == AGG_AVG ? kLdsWordsAvg : kWordsMinMax));
        for (uint32_t i = 0; i < nw; i++) {
            uint64_t ident = 0;
            if (ag.kind == AGG_MIN) ident = i == 1 ? (uint64_t)INT64_MAX : (i >= 2 ? ~0ull : 0ull);
            if (ag.kind == AGG_MAX) ident = i == 1 ? (uint64_t)INT64_MIN : 0ull;
            for (uint32_t s = tid; s < S; s += BLOCK) w[(size_t)i * S + s] = ident;
        }
    }
}

__device__ __forceinline__ void glob_row_init(const Program& P, uint64_t* g) {
    for (uint32_t a = 0; a < P.naggs; a++) {
        const AggSpec& ag = P.aggs[a];
        uint64_t* w = g + ag.glob_off;
        if (ag.distinct) { for (int i = 0; i < (int)kGlobWordsDistinctSum; i++) if (i < (int)kGlobWordsDistinct || ag.kind == AGG_SUM || ag.kind == AGG_AVG) w[i] = 0; continue; }
        switch (ag.kind) {
            case AGG_COUNT:
            case AGG_COUNTN: w[0] = 0; break;
            case AGG_SUM: for (int i = 0; i < (int)kGlobWordsSum; i++) w[i] = 0; break;
            case AGG_AVG: for (int i = 0; i < (int)kGlobWordsAvg; i++) w[i] = 0; break;
            case AGG_MIN: w[0] = 0; w[1] = (uint64_t)INT64_MAX; w[2] = ~0ull; w[3] = ~0ull; break;
            default: w[0] = 0; w[1] = (uint64_t)INT64_MIN; w[2] = 0; w[3] = 0; break;
        }
    }
}

// CumulateInitial of one aggregate for one row straight into a GLOBAL row (bypass path: LDS table full, or an
// integer too large for the narrow LDS sum).  algebra/agg_sum.go:86-97, agg_count.go:102-116, agg_countn.go:84-97,
// agg_avg.go:85-97, agg_min.go:83-94, agg_max.go:83-94.
N1K_DEV void acc_global(const Program& P, const AggSpec& ag, uint64_t* g, uint32_t tag, uint64_t p) {
    unsigned long long* w = (unsigned long long*)(g + ag.glob_off);
    switch (ag.kind) {
        case AGG_COUNT:
            if (!ag.has_operand || tag > T_NULL) atomicAdd(&w[0], 1ull);
            break;
        case AGG_COUNTN:
            if (tag == T_INT || tag == T_FLOAT) atomicAdd(&w[0], 1ull);
            break;
        case AGG_SUM:
        case AGG_AVG:
            if (tag == T_INT) {
                int64_t x = (int64_t)p;
                atomicAdd(&w[0], (unsigned long long)(uint32_t)x);
                atomicAdd(&w[1], (unsigned long long)(x >> 32));
                atomicOr(&w[3], x < 0 ? (unsigned long long)SF_NEG_INT : (unsigned long long)SF_NONNEG_INT);
                if (ag.kind == AGG_AVG) atomicAdd(&w[4], 1ull);
            } else if (tag == T_FLOAT) {
                atomicAdd((double*)&w[2], as_f64(p));
                atomicOr(&w[3], (unsigned long long)SF_FLOAT);
                if (ag.kind == AGG_AVG) atomicAdd(&w[4], 1ull);
            }
            break;
        case AGG_MIN:
        case AGG_MAX: {
            if (tag <= T_NULL) break;
            bool mn = ag.kind == AGG_MIN;
            if (tag == T_FALSE || tag == T_TRUE) {
                atomicOr(&w[0], tag == T_TRUE ? (unsigned long long)MM_TRUE : (unsigned long long)MM_FALSE);
            } else if (tag == T_INT) {
                atomicOr(&w[0], (unsigned long long)MM_INT);
                if (mn) atomicMin((long long*)&w[1], (long long)p); else atomicMax((long long*)&w[1], (long long)p);
            } else if (tag == T_FLOAT) {
                atomicOr(&w[0], (unsigned long long)MM_FLOAT);
                unsigned long long s = f64_sortable(as_f64(p));
                if (mn) atomicMin(&w[2], s); else atomicMax(&w[2], s);
            } else if (tag == T_STRING) {
                atomicOr(&w[0], (unsigned long long)MM_STRING);
                unsigned long long s = ((unsigned long long)P.str_rank[(uint32_t)p] << 32) | (uint32_t)p;
                if (mn) atomicMin(&w[3], s); else atomicMax(&w[3], s);
            } else {
                atomicOr(&w[0], (unsigned long long)MM_OTHER);
            }
            break;
        }
    }
}

// set a read-mostly flag bit in LDS: after the first row of a kind the atomic is skipped
N1K_DEV void lds_set_flag(lds_u64* w, unsigned long long bit) {
    if (!(lds_peek(w) & bit)) lds_or_u64(w, bit);
}

// CumulateInitial into the workgroup's LDS slot.  Returns false when the value must take the global path
// (|int| >= 2^40: the 64-bit LDS sum of a workgroup's share could overflow).
N1K_DEV bool acc_lds(const Program& P, const AggSpec& ag, uint64_t* lds, uint32_t S, uint32_t slot, uint32_t tag, uint64_t p) {
    lds_u64* w = lds_word(lds, ag.lds_off * S + slot);  // word i at w[i * S]
    switch (ag.kind) {
        case AGG_COUNT:
            if (!ag.has_operand || tag > T_NULL) lds_add_u64(w, 1ull);
            return true;
        case AGG_COUNTN:
            if (tag == T_INT || tag == T_FLOAT) lds_add_u64(w, 1ull);
            return true;
        case AGG_SUM:
        case AGG_AVG:
            if (tag == T_INT) {
                int64_t x = (int64_t)p;
                if (x >= (1ll << 40) || x <= -(1ll << 40)) return false;
                lds_add_u64(w, (unsigned long long)x);
                lds_set_flag(w + 2 * S, x < 0 ? (unsigned long long)SF_NEG_INT : (unsigned long long)SF_NONNEG_INT);
                if (ag.kind == AGG_AVG) lds_add_u64(w + 3 * S, 1ull);
            } else if (tag == T_FLOAT) {
                lds_add_f64(w + S, as_f64(p));
                lds_set_flag(w + 2 * S, (unsigned long long)SF_FLOAT);
                if (ag.kind == AGG_AVG) lds_add_u64(w + 3 * S, 1ull);
            }
            return true;
        default: {  // MIN / MAX
            if (tag <= T_NULL) return true;
            bool mn = ag.kind == AGG_MIN;
            if (tag == T_FALSE || tag == T_TRUE) {
                lds_set_flag(w, tag == T_TRUE ? (unsigned long long)MM_TRUE : (unsigned long long)MM_FALSE);
            } else if (tag == T_INT) {
                lds_set_flag(w, (unsigned long long)MM_INT);
                long long x = (long long)p, cur = (long long)lds_peek(w + S);
                if (mn ? x < cur : x > cur) { if (mn) lds_min_i64(w + S, x); else lds_max_i64(w + S, x); }
            } else if (tag == T_FLOAT) {
                lds_set_flag(w, (unsigned long long)MM_FLOAT);
                unsigned long long x = f64_sortable(as_f64(p)), cur = lds_peek(w + 2 * S);
                if (mn ? x < cur : x > cur) { if (mn) lds_min_u64(w + 2 * S, x); else lds_max_u64(w + 2 * S, x); }
            } else if (tag == T_STRING) {
                lds_set_flag(w, (unsigned long long)MM_STRING);
                unsigned long long x = ((unsigned long long)P.str_rank[(uint32_t)p] << 32) | (uint32_t)p;
                unsigned long long cur = lds_peek(w + 3 * S);
                if (mn ? x < cur : x > cur) { if (mn) lds_min_u64(w + 3 * S, x); else lds_max_u64(w + 3 * S, x); }
            } else {
                lds_set_flag(w, (unsigned long long)MM_OTHER);
            }
            return true;
        }
    }
}

// The VALUE part of CumulateInitial for COUNT / COUNTN / SUM / AVG (no returning LDS operation: nothing to wait for).  Returns
// the bit the slot's SUM / AVG flags word has to carry (0: none; lds_set_flag on word lds_off + 2 is the caller's, so that it
// can read the flags of several records with one wait), or ~0u when the value must take the global path (|int| >= 2^40).
N1K_DEV uint32_t acc_lds_value(const AggSpec& ag, uint64_t* lds, uint32_t S, uint32_t slot, uint32_t tag, uint64_t p) {
    lds_u64* w = lds_word(lds, ag.lds_off * S + slot);  // word i at w[i * S]
    if (ag.kind == AGG_COUNT) {
        if (!ag.has_operand || tag > T_NULL) lds_add_u64(w, 1ull);
        return 0u;
    }
    if (ag.kind == AGG_COUNTN) {
        if (tag == T_INT || tag == T_FLOAT) lds_add_u64(w, 1ull);
        return 0u;
    }
    if (tag == T_INT) {
        const int64_t x = (int64_t)p;
        if (x >= (1ll << 40) || x <= -(1ll << 40)) return ~0u;
        lds_add_u64(w, (unsigned long long)x);
        if (ag.kind == AGG_AVG) lds_add_u64(w + 3 * S, 1ull);
        return x < 0 ? (uint32_t)SF_NEG_INT : (uint32_t)SF_NONNEG_INT;
    }
    if (tag == T_FLOAT) {
        lds_add_f64(w + S, as_f64(p));
        if (ag.kind == AGG_AVG) lds_add_u64(w + 3 * S, 1ull);
        return (uint32_t)SF_FLOAT;
    }
    return 0u;
}

// CumulateIntermediate: fold one LDS slot into its global row (algebra/agg_*.go CumulateIntermediate)
N1K_DEV void merge_slot(const Program& P, const uint64_t* lds, uint32_t S, uint32_t slot, uint64_t* g) {
    for (uint32_t a = 0; a < P.naggs; a++) {
        const AggSpec& ag = P.aggs[a];
        const uint64_t* l = lds + (size_t)ag.lds_off * S + slot;  // word i at l[i * S]
        unsigned long long* w = (unsigned long long*)(g + ag.glob_off);
        if (ag.distinct) {
            for (uint32_t i = 0; i < ag.lds_n; i++)
                if (l[(size_t)i * S]) atomicAdd(&w[1 + i], (unsigned long long)l[(size_t)i * S]);
            continue;
        }
        switch (ag.kind) {
            case AGG_COUNT:
            case AGG_COUNTN:
                if (l[0]) atomicAdd(&w[0], (unsigned long long)l[0]);
                break;
            case AGG_SUM:
            case AGG_AVG: {
                uint64_t fl = l[2 * (size_t)S];
                if (!fl) break;
                if (fl & (SF_NONNEG_INT | SF_NEG_INT)) {
                    int64_t x = (int64_t)l[0];
                    atomicAdd(&w[0], (unsigned long long)(uint32_t)x);
                    atomicAdd(&w[1], (unsigned long long)(x >> 32));
                }
                if (fl & SF_FLOAT) atomicAdd((double*)&w[2], as_f64(l[(size_t)S]));
                atomicOr(&w[3], (unsigned long long)fl);
                if (ag.kind == AGG_AVG) atomicAdd(&w[4], (unsigned long long)l[3 * (size_t)S]);
                break;
            }
            default: {
                uint64_t fl = l[0];
                if (!fl) break;
                bool mn = ag.kind == AGG_MIN;
                atomicOr(&w[0], (unsigned long long)fl);
                if (fl & MM_INT) { if (mn) atomicMin((long long*)&w[1], (long long)l[(size_t)S]); else atomicMax((long long*)&w[1], (long long)l[(size_t)S]); }
                if (fl & MM_FLOAT) { if (mn) atomicMin(&w[2], (unsigned long long)l[2 * (size_t)S]); else atomicMax(&w[2], (unsigned long long)l[2 * (size_t)S]); }
                if (fl & MM_STRING) { if (mn) atomicMin(&w[3], (unsigned long long)l[3 * (size_t)S]); else atomicMax(&w[3], (unsigned long long)l[3 * (size_t)S]); }
                break;
            }
        }
    }
}

// ------------------------------------------------------------------ DISTINCT members
//
// value.Set (value/set.go:22-110) keeps one hash map per type: ints and integral floats share the int map, other
// floats their own, strings / arrays / objects are keyed by text (here: dictionary code), booleans by value.
// CountDistinct adds every operand of type > NULL (algebra/agg_count_distinct.go:84-95), CountnDistinct every
// NUMBER.  Returns false when the operand does not enter the set.
N1K_DEV bool distinct_classify(uint32_t kind, uint32_t tag, uint64_t p, uint32_t& cls, uint64_t& val) {
    if (tag <= T_NULL) return false;
    if (tag == T_INT) { cls = DC_INT; val = p; return true; }
    if (tag == T_FLOAT) {
        double d = as_f64(p);
        if (is_int_f64(d)) { cls = DC_INT; val = (uint64_t)go_f2i(d); }
        else { cls = DC_FLOAT; val = p; }
        return true;
    }
    if (kind != AGG_COUNT) return false;  // COUNTN / SUM / AVG (DISTINCT): NUMBER operands only
    cls = DC_OTHER;
    val = ((uint64_t)tag << 40) | (p & 0xFFFFFFFFFFull);
    return true;
}

// One-word form of a (group key, class, value) member of a COUNT(DISTINCT) set, when both parts are small enough:
//   [key : key_bits][class : 3][value : val_bits]      (key_bits + 3 + val_bits == 64)
// Non-integral floats never are; DC_OTHER values (tag << 40 | code) are re-packed as code << 4 | tag.
N1K_DEV bool member_word_bits(uint32_t key_bits, uint32_t val_bits, uint64_t key, uint32_t cls, uint64_t val, uint64_t& word) {
    if (cls == DC_FLOAT) return false;
    uint64_t v = cls == DC_INT ? zigzag((int64_t)val) : (((val & 0xFFFFFFFFFFull) << 4) | (val >> 40));
    if ((key >> key_bits) != 0ull || (v >> val_bits) != 0ull) return false;
    word = (key << (val_bits + 3)) | ((uint64_t)cls << val_bits) | v;
    return true;
}

// radix digit of a member word / record key: 8 bits of its hash (equal words always share every digit); `shift` counts
// as if the hash had 64 bits (first digit: 56)
N1K_DEV uint32_t radix_bin(uint64_t w, uint32_t shift) { return (part_hash(w) >> (shift - 32u)) & 255u; }  // shift: 56, 48, 40, 32

// ------------------------------------------------------------------ 16-byte records of the partitioned GROUP BY
//
// (packed group key, one aggregate operand).  The key's bit 63 is free (packed keys use 63 bits) and says "the operand
// is an INT"; everything else travels in the operand word: a FLOAT as its bits (any NaN as the canonical quiet NaN), the
// other tags boxed into a negative quiet NaN that no float of the path ever has: 0xFFF8 << 48 | tag << 40 | payload (a
// dictionary code: < 2^40).
struct Rec16 {
    uint64_t k, v;
};
constexpr uint64_t kRecIntFlag = 1ull << 63, kRecBox = 0xFFF8000000000000ull;
N1K_DEV Rec16 rec16_encode(uint64_t key, uint32_t tag, uint64_t p) {
    Rec16 r;
    r.k = key;
    if (tag == T_INT) {
        r.k |= kRecIntFlag;
        r.v = p;
    } else if (tag == T_FLOAT) {
        const double d = as_f64(p);
        r.v = d != d ? 0x7FF8000000000000ull : p;
    } else
        r.v = kRecBox | ((uint64_t)tag << 40) | (p & 0xFFFFFFFFFFull);
    return r;
}
N1K_DEV void rec16_decode(const Rec16& r, uint64_t& key, uint32_t& tag, uint64_t& p) {
    key = r.k & ~kRecIntFlag;
    if (r.k & kRecIntFlag) {
        tag = T_INT;
        p = r.v;
    } else if ((r.v >> 48) == 0xFFF8ull) {
        tag = (uint32_t)(r.v >> 40) & 0xFFu;
        p = r.v & 0xFFFFFFFFFFull;
    } else {
        tag = T_FLOAT;
        p = r.v;
    }
}

// perfect-hash slot -> packed group key (inverse of slot = sum(field_k * stride_k))
N1K_DEV uint64_t fast_slot_key(const FastArgs& F, uint32_t slot) {
    uint64_t key = 0;
#pragma unroll
    for (int k = kFastKeys - 1; k >= 0; k--) {
        if (k < (int)F.nkeys) {
            uint32_t f = slot / F.keys[k].stride;
            slot -= f * F.keys[k].stride;
            key |= (uint64_t)f << F.keys[k].shift;
        }
    }
    return key;
}

}  // namespace n1k
