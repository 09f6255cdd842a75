// n1k_spec.h — plan-specialised scan kernel.
//
// scan_spec_kernel<Spec, R, BLOCK, WIDE> is the same algorithm as scan_fast_kernel (Filter -> perfect-hash
// InitialGroup in LDS -> atomic merge into the global table), but the SHAPE of the plan is a compile-time
// `Spec`: which columns exist and their kinds, which cheap predicate terms are ANDed, which dictionary columns
// are the keys, which aggregates run over which columns.  Everything else (pointers, constants, dictionary
// radix, table sizes) stays a run-time argument, so one instantiation serves every query of that shape.
// The compiler sees straight-line code per row: no interpretation, no uniform branches, no descriptor loads.
//
// WIDE: two adjacent rows per lane and load: payload 16 B, tags 2 B, codes 8 B per lane (needs aligned bases).
#pragma once
#include "n1k_tables.h"

namespace n1k {

// A Spec type provides:
//   static constexpr int ncols, nterms, nkeys, naggs;
//   static constexpr uint32_t col_kind[kFastCols];
//   static constexpr SpecTerm terms[kFastTerms];
//   static constexpr uint32_t key_col[kFastKeys];
//   static constexpr SpecAgg aggs[kFastAggs];
#define N1K_DEFINE_SPEC(NAME, NCOLS, K0, K1, K2, NTERMS, T0, T1, NKEYS, KC0, KC1, NAGGS, A0, A1, A2, A3, A4) \
    struct NAME {                                                                                            \
        static constexpr int ncols = NCOLS, nterms = NTERMS, nkeys = NKEYS, naggs = NAGGS;                    \
        static constexpr uint32_t col_kind[kFastCols] = {K0, K1, K2};                                         \
        static constexpr SpecTerm terms[kFastTerms] = {T0, T1};                                               \
        static constexpr uint32_t key_col[kFastKeys] = {KC0, KC1};                                            \
        static constexpr SpecAgg aggs[kFastAggs] = {A0, A1, A2, A3, A4};                                      \
    }

// flag bits: fire-and-forget ds_or (no LDS read, so the row loop never waits on lgkmcnt)
#ifndef SPEC_FLAG
#define SPEC_FLAG(ptr, bit) lds_or_u64((ptr), (bit))
#endif

// a Spec whose keys are not all dictionary columns uses the open-addressed LDS table (decided at compile time, so
// the perfect-hash kernels carry none of the hashing code)
template <class Spec>
constexpr bool spec_hashed() {
    bool h = false;
    for (int k = 0; k < Spec::nkeys; k++) h = h || Spec::col_kind[Spec::key_col[k]] != COLK_DICT32;
    return h;
}

template <class Spec>
N1K_DEV bool spec_term_true(int t, const FastArgs& F, uint32_t tg, uint64_t p) {
    constexpr int kT = kFastTerms;
    (void)kT;
    const uint32_t op = Spec::terms[t].op;
    switch (op) {
        case TERM_IS_NULL: return tg == T_NULL;
        case TERM_IS_NOT_NULL: return tg > T_NULL;
        case TERM_IS_MISSING: return tg == T_MISSING;
        case TERM_IS_NOT_MISSING: return tg != T_MISSING;
        case TERM_IS_VALUED: return tg > T_NULL;
        case TERM_IS_NOT_VALUED: return tg <= T_NULL;
        case TERM_STR_EQ: return tg == T_STRING && p == F.terms[t].cpayload;
        default: {
            // collation of the row value against the NUMBER constant (value/integer.go:100-118, float.go:106-121)
            int c;
            const uint64_t cp = F.terms[t].cpayload;
            if (tg == T_INT) {
                if (Spec::terms[t].const_int) {
                    int64_t x = (int64_t)p, y = (int64_t)cp;
                    c = x < y ? -1 : (x > y ? 1 : 0);
                } else {
                    c = collate_f64((double)(int64_t)p, as_f64(cp));
                }
            } else if (tg == T_FLOAT) {
                c = collate_f64(as_f64(p), Spec::terms[t].const_int ? (double)(int64_t)cp : as_f64(cp));
            } else if (tg <= T_NULL) {
                return false;  // MISSING / NULL are never TRUE
            } else {
                c = tg < T_INT ? -1 : 1;  // BOOLEAN below NUMBER, STRING/ARRAY/OBJECT above
            }
            return op == TERM_NUM_LT ? c < 0 : op == TERM_NUM_LE ? c <= 0 : op == TERM_NUM_GT ? c > 0
                   : op == TERM_NUM_GE ? c >= 0 : (c == 0 && (tg == T_INT || tg == T_FLOAT));
        }
    }
}

// CumulateInitial of aggregate `a` (compile-time kind) into the LDS slot; false -> take the global path
template <class Spec>
N1K_DEV bool spec_acc(int a, const Program& P, uint64_t* lds, uint32_t S, uint32_t slot, uint32_t tag, uint64_t p) {
    const uint32_t kind = Spec::aggs[a].kind;
    lds_u64* w = lds_word(lds, P.aggs[a].lds_off * S + slot);
    if (kind == AGG_COUNT) {
        if (!Spec::aggs[a].has_operand || tag > T_NULL) lds_add_u64(w, 1ull);
        return true;
    }
    if (kind == AGG_COUNTN) {
        if (tag == T_INT || tag == T_FLOAT) lds_add_u64(w, 1ull);
        return true;
    }
    if (kind == AGG_SUM || kind == AGG_AVG) {
        if (tag == T_INT) {
            int64_t x = (int64_t)p;
            if (x >= (1ll << 40) || x <= -(1ll << 40)) return false;
            lds_add_u64(w, (unsigned long long)x);
            SPEC_FLAG(w + 2 * S, x < 0 ? (unsigned long long)SF_NEG_INT : (unsigned long long)SF_NONNEG_INT);
            if (kind == AGG_AVG) lds_add_u64(w + 3 * S, 1ull);
        } else if (tag == T_FLOAT) {
            lds_add_f64(w + S, as_f64(p));
            SPEC_FLAG(w + 2 * S, (unsigned long long)SF_FLOAT);
            if (kind == AGG_AVG) lds_add_u64(w + 3 * S, 1ull);
        }
        return true;
    }
    // MIN / MAX
    if (tag <= T_NULL) return true;
    const bool mn = kind == AGG_MIN;
    if (tag == T_INT) {
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
    } else if (tag == T_FALSE || tag == T_TRUE) {
        lds_set_flag(w, tag == T_TRUE ? (unsigned long long)MM_TRUE : (unsigned long long)MM_FALSE);
    } else {
        lds_set_flag(w, (unsigned long long)MM_OTHER);
    }
    return true;
}

// one row, everything about the plan shape folded at compile time
template <class Spec>
N1K_DEV void spec_row(const Program& P, const FastArgs& F, const GlobalTable& G, unsigned long long* ngroups,
                      uint64_t* lds, uint32_t S, uint32_t* lds_fill, const uint32_t (&tg)[kFastCols],
                      const uint64_t (&pv)[kFastCols], uint32_t& selected, uint32_t& unpackable) {
    bool pass = true;
#pragma unroll
    for (int t = 0; t < Spec::nterms; t++) pass = pass && spec_term_true<Spec>(t, F, tg[Spec::terms[t].col], pv[Spec::terms[t].col]);
    if (!pass) return;
    uint32_t slot = 0;
    uint64_t key = 0;
    long long grow = -1;
    constexpr bool kHashed = spec_hashed<Spec>();
    if (kHashed) {
        // open-addressed LDS table on the packed key (integer / mixed keys, or a dictionary domain beyond the LDS)
#pragma unroll
        for (int k = 0; k < Spec::nkeys; k++) {
            uint64_t f = 0, canon;
            if (!pack_key_field(P, P.keys[k], tg[Spec::key_col[k]], pv[Spec::key_col[k]], f, canon)) {
                unpackable = 1;
                return;
            }
            key |= f << P.keys[k].shift;
        }
        int sl = lds_find_or_insert(lds, S, key, lds_fill, F.lds_max_fill);
        if (sl < 0) {
            grow = global_find_or_insert(G, key, F.err_flags, ngroups);
            if (grow < 0) return;
        }
        slot = (uint32_t)sl;
    } else {
#pragma unroll
        for (int k = 0; k < Spec::nkeys; k++) {
            const uint32_t t = tg[Spec::key_col[k]];
            uint32_t f = t == T_MISSING ? 0u : (t == T_NULL ? 1u : (uint32_t)pv[Spec::key_col[k]] + 2u);
            if ((t > T_NULL && t != T_STRING) || f >= F.keys[k].radix) {
                unpackable = 1;
                return;
            }
            slot += f * F.keys[k].stride;
        }
        *(volatile lds_u64*)lds_word(lds, slot) = 1ull;  // "touched": every writer stores the same value, nobody reads it here
    }
    selected++;
#pragma unroll
    for (int a = 0; a < Spec::naggs; a++) {
        const uint32_t c = Spec::aggs[a].has_operand ? Spec::aggs[a].col : 0u;
        const uint32_t t = Spec::aggs[a].has_operand ? tg[c] : (uint32_t)T_NULL;
        const uint64_t p = Spec::aggs[a].has_operand ? pv[c] : 0ull;
        if (kHashed && grow >= 0) {  // the LDS table is full: this group lives in the global table only
            acc_global(P, P.aggs[a], &G.acc[(size_t)grow * P.glob_words], t, p);
        } else if (!spec_acc<Spec>(a, P, lds, S, slot, t, p)) {  // rare: an integer too large for the narrow LDS sum
            long long g = global_find_or_insert(G, kHashed ? key : fast_slot_key(F, slot), F.err_flags, ngroups);
            if (g >= 0) acc_global(P, P.aggs[a], &G.acc[(size_t)g * P.glob_words], t, p);
        }
    }
}

template <class Spec, int R, int BLOCK, bool WIDE>
N1K_DEV void scan_spec_body(const Program& P, const FastArgs& F, const GlobalTable& G, unsigned long long* ngroups) {
    extern __shared__ uint64_t lds[];
    __shared__ uint32_t lds_fill;
    const uint32_t S = F.lds_slots;
    const uint32_t tid = threadIdx.x;
    lds_table_init<BLOCK>(P, lds, S, tid);
    if (tid == 0) lds_fill = 0;
    __syncthreads();

    uint32_t unpackable = 0, selected = 0;
    constexpr uint32_t kRowsPerItem = WIDE ? 2u : 1u;
    const uint32_t nitems = WIDE ? F.nrows / 2u : F.nrows;  // the engine passes an even row count to WIDE launches
    const uint32_t tile = BLOCK * R;

    for (uint32_t base = blockIdx.x * tile; base < nitems; base += gridDim.x * tile) {
        // issue every load of the tile first (R items x ncols columns), then compute
        uint32_t tg[R][kRowsPerItem][kFastCols];
        uint64_t pv[R][kRowsPerItem][kFastCols];
        bool valid[R];
#pragma unroll
        for (int j = 0; j < R; j++) {
            const uint32_t i = base + (uint32_t)j * BLOCK + tid;
            valid[j] = i < nitems;
#pragma unroll
            for (int c = 0; c < kFastCols; c++) {
                if (c < Spec::ncols) {
                    if (Spec::col_kind[c] == COLK_DICT32) {
                        if (WIDE) {
                            typedef uint32_t n1k_u32x2 __attribute__((ext_vector_type(2)));
                            n1k_u32x2 cc = {0xFFFFFFFFu, 0xFFFFFFFFu};
                            if (valid[j]) cc = __builtin_nontemporal_load((const n1k_u32x2*)F.cols[c].codes + i);
                            pv[j][0][c] = cc.x;
                            pv[j][WIDE ? 1 : 0][c] = cc.y;
                        } else {
                            pv[j][0][c] = valid[j] ? F.cols[c].codes[i] : 0xFFFFFFFFu;
                        }
#pragma unroll
                        for (int h = 0; h < (int)kRowsPerItem; h++) {
                            uint32_t code = (uint32_t)pv[j][h][c];
                            tg[j][h][c] = code == 0xFFFFFFFFu ? (uint32_t)T_MISSING : (code == 0xFFFFFFFEu ? (uint32_t)T_NULL : (uint32_t)T_STRING);
                        }
                    } else {
                        if (WIDE) {
                            typedef unsigned long long n1k_u64x2 __attribute__((ext_vector_type(2)));
                            n1k_u64x2 pp = {0ull, 0ull};
                            if (valid[j]) pp = __builtin_nontemporal_load((const n1k_u64x2*)F.cols[c].payload + i);
                            uint32_t tt = valid[j] ? (uint32_t)__builtin_nontemporal_load((const uint16_t*)F.cols[c].tags + i) : 0u;
                            pv[j][0][c] = pp.x;
                            pv[j][WIDE ? 1 : 0][c] = pp.y;
                            tg[j][0][c] = tt & 255u;
                            tg[j][WIDE ? 1 : 0][c] = tt >> 8;
                        } else {
                            pv[j][0][c] = valid[j] ? F.cols[c].payload[i] : 0ull;
                            tg[j][0][c] = valid[j] ? (uint32_t)F.cols[c].tags[i] : (uint32_t)T_MISSING;
                        }
                    }
                } else {
#pragma unroll
                    for (int h = 0; h < (int)kRowsPerItem; h++) { tg[j][h][c] = T_MISSING; pv[j][h][c] = 0; }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < R; j++) {
            if (valid[j]) {
#pragma unroll
                for (int h = 0; h < (int)kRowsPerItem; h++)
                    spec_row<Spec>(P, F, G, ngroups, lds, S, &lds_fill, tg[j][h], pv[j][h], selected, unpackable);
            }
        }
    }

    if (unpackable) atomicOr(F.err_flags, (uint32_t)ERR_UNPACKABLE_KEY);
    // rows that passed the Filter (≙ Filter #itemsOut): wave shuffle, then ONE LDS counter per workgroup — thousands of
    // same-address global atomics at the end of the kernel cost ~10 % of its time
    __shared__ unsigned int block_selected;
    if (tid == 0) block_selected = 0;
    __syncthreads();
    for (int off = 32; off > 0; off >>= 1) selected += __shfl_down(selected, off, 64);
    if ((tid & 63) == 0 && selected) atomicAdd(&block_selected, selected);
    __syncthreads();
    if (F.slabs) {
        // hand the workgroup's partial groups to merge_slabs_kernel: plain coalesced stores, no atomics
        uint64_t* slab = F.slabs + (size_t)blockIdx.x * P.lds_words * S;
        for (uint32_t i = tid; i < P.lds_words * S; i += BLOCK) slab[i] = lds[i];
        if (tid == 0) F.block_selected[blockIdx.x] = block_selected;
        return;
    }
    if (tid == 0 && block_selected) atomicAdd(F.rows_selected, (unsigned long long)block_selected);
    // K4: merge this workgroup's partial groups into the global table (≙ IntermediateGroup)
    for (uint32_t s = tid; s < S; s += BLOCK) {
        if (lds[s] == kEmptyKey) continue;
        long long g = global_find_or_insert(G, spec_hashed<Spec>() ? lds[s] : fast_slot_key(F, s), F.err_flags, ngroups);
        if (g < 0) continue;
        merge_slot(P, lds, S, s, &G.acc[(size_t)g * P.glob_words]);
    }
}

// ahead-of-time instantiations use this kernel; kernels compiled at run time wrap scan_spec_body themselves
template <class Spec, int R, int BLOCK, bool WIDE>
__global__ __launch_bounds__(BLOCK) void scan_spec_kernel(const Program P, const FastArgs F, const GlobalTable G,
                                                         unsigned long long* ngroups) {
    scan_spec_body<Spec, R, BLOCK, WIDE>(P, F, G, ngroups);
}

}  // namespace n1k
