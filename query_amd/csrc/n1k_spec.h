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
#include "n1k_scatter.h"
#include "n1k_tables.h"

namespace n1k {

// A Spec type provides:
//   static constexpr int ncols, nterms, nkeys, naggs;
//   static constexpr uint32_t col_kind[kFastCols];
//   static constexpr SpecTerm terms[kFastTerms];
//   static constexpr uint32_t key_col[kFastKeys];
//   static constexpr SpecAgg aggs[kFastAggs];
//   static constexpr int nderived; static constexpr SpecDerived derived[kFastDerived];   (fused arithmetic nodes: column
//   slots ncols .. ncols + nderived - 1, evaluated in registers from the loaded columns — run-time-built shapes only)
#define N1K_DEFINE_SPEC(NAME, NCOLS, K0, K1, K2, NTERMS, T0, T1, NKEYS, KC0, KC1, NAGGS, A0, A1, A2, A3, A4) \
    struct NAME {                                                                                            \
        static constexpr int ncols = NCOLS, nterms = NTERMS, nkeys = NKEYS, naggs = NAGGS, nderived = 0;      \
        static constexpr SpecDerived derived[kFastDerived] = {};                                              \
        static constexpr uint32_t col_kind[kFastCols] = {K0, K1, K2};                                         \
        static constexpr SpecTerm terms[kFastTerms] = {T0, T1};                                               \
        static constexpr uint32_t key_col[kFastKeys] = {KC0, KC1};                                            \
        static constexpr SpecAgg aggs[kFastAggs] = {A0, A1, A2, A3, A4};                                      \
    }

// flag bits: fire-and-forget ds_or (no LDS read, so the row loop never waits on lgkmcnt)
// Two tiles in flight for every shape (0: only for those whose tiles end in barriers — the A/B switch).  With the tiles issued
// unconditionally and decoded when they are processed (spec_issue_tile / spec_decode_tile) the second tile really is in flight
// while the first is processed: config 2 at 100 M rows 0.270 -> 0.258 ms (scan + merge, three alternations on one box).
#ifndef N1K_SPEC_PIPE_ALL
#define N1K_SPEC_PIPE_ALL 1
#endif
#ifndef SPEC_FLAG
#define SPEC_FLAG(ptr, bit) lds_or_u64((ptr), (bit))
#endif

// DISTINCT aggregates of a Spec (COUNT(DISTINCT col) over one-word members, see WordLogArgs)
template <class Spec>
constexpr int spec_ndistinct() {
    int n = 0;
    for (int a = 0; a < Spec::naggs; a++) n += Spec::aggs[a].distinct ? 1 : 0;
    return n;
}
template <class Spec>
constexpr int spec_distinct_index(int a) {
    int n = 0;
    for (int i = 0; i < a; i++) n += Spec::aggs[i].distinct ? 1 : 0;
    return n;
}

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
N1K_DEV void spec_flag(lds_u64* w, unsigned long long bit) {
    // Shapes with COUNT(DISTINCT) run at the CU's LDS-atomic rate (about one lane per clock: four atomics per row were
    // 0.65 ms per 100 M rows), so there the flag is read first and set once; the other shapes are bound by HBM and
    // prefer the fire-and-forget atomic, which never makes the row loop wait for an LDS read.
    if (spec_ndistinct<Spec>() > 0) lds_set_flag(w, bit);
    else SPEC_FLAG(w, bit);
}

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
            spec_flag<Spec>(w + 2 * S, x < 0 ? (unsigned long long)SF_NEG_INT : (unsigned long long)SF_NONNEG_INT);
            if (kind == AGG_AVG) lds_add_u64(w + 3 * S, 1ull);
        } else if (tag == T_FLOAT) {
            lds_add_f64(w + S, as_f64(p));
            spec_flag<Spec>(w + 2 * S, (unsigned long long)SF_FLOAT);
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
// a member that does not fit one word (non-integral float, wide value): the (key, value, class) pair log and the
// per-class operand counts of the group, as the interpreter kernel keeps them — rare, so plain global atomics
N1K_DEV void spec_pair_log(const Program& P, const GlobalTable& G, const WordLogArgs& L, int d, const AggSpec& ag, uint64_t gkey,
                           uint32_t cls, uint64_t val, uint32_t* err_flags, unsigned long long* ngroups) {
    const unsigned long long pos = atomicAdd(&L.log_cursor[L.log_index[d]], 1ull);
    if (pos < L.log_capacity) {
        L.log_key[d][pos] = gkey;
        L.log_val[d][pos] = val;
        L.log_cls[d][pos] = (uint8_t)cls;
    } else
        atomicOr(err_flags, (uint32_t)ERR_TABLE_FULL);
    const long long g = global_find_or_insert(G, gkey, err_flags, ngroups);
    if (g >= 0) atomicAdd((unsigned long long*)&G.acc[(size_t)g * P.glob_words + ag.glob_off + 1 + cls], 1ull);
}

template <class Spec>
N1K_DEV void spec_row(const Program& P, const FastArgs& F, const GlobalTable& G, unsigned long long* ngroups,
                      uint64_t* lds, uint32_t S, uint32_t* lds_fill, const uint32_t (&tg)[kSpecCols],
                      const uint64_t (&pv)[kSpecCols], uint32_t& selected, uint32_t& unpackable, const WordLogArgs& L,
                      uint64_t* dcache, uint64_t (&words)[kSpecDistinct], uint32_t (&bins)[kSpecDistinct]) {
    constexpr int kND = spec_ndistinct<Spec>();
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
            if (kND) key |= (uint64_t)f << F.keys[k].shift;  // the packed key, for the members of the DISTINCT sets
        }
        *(volatile lds_u64*)lds_word(lds, slot) = 1ull;  // "touched": every writer stores the same value, nobody reads it here
    }
    selected++;
#pragma unroll
    for (int a = 0; a < Spec::naggs; a++) {
        const uint32_t c = Spec::aggs[a].has_operand ? Spec::aggs[a].col : 0u;
        const uint32_t t = Spec::aggs[a].has_operand ? tg[c] : (uint32_t)T_NULL;
        const uint64_t p = Spec::aggs[a].has_operand ? pv[c] : 0ull;
        if (Spec::aggs[a].distinct) {
            // setAdd (algebra/agg_util.go:30-47): the (group, value) member leaves as one word; n1k_finish builds the sets
            const int d = spec_distinct_index<Spec>(a);
            uint32_t cls;
            uint64_t val, word;
            if (!distinct_classify(AGG_COUNT, t, p, cls, val)) continue;
            if (!member_word_bits(L.nw_key_bits, L.nw_val_bits, key, cls, val, word)) {
                spec_pair_log(P, G, L, d, P.aggs[a], key, cls, val, F.err_flags, ngroups);
                continue;
            }
            const uint32_t hw = part_hash(word);  // top byte: the hash region; low bits: the slot of the workgroup's cache
            if (L.dcache_slots) {  // skip a word this workgroup logged already (direct-mapped; a race costs a redundant word)
                lds_u64* c = lds_word(dcache, (uint32_t)d * L.dcache_slots + (hw & (L.dcache_slots - 1u)));
                if (lds_peek(c) == word) continue;
                *(volatile lds_u64*)c = word;
            }
            words[d] = word;
            bins[d] = hw >> 24;
            continue;
        }
        if (kHashed && grow >= 0) {  // the LDS table is full: this group lives in the global table only
            acc_global(P, P.aggs[a], &G.acc[(size_t)grow * P.glob_words], t, p);
        } else if (!spec_acc<Spec>(a, P, lds, S, slot, t, p)) {  // rare: an integer too large for the narrow LDS sum
            long long g = global_find_or_insert(G, kHashed ? key : fast_slot_key(F, slot), F.err_flags, ngroups);
            if (g >= 0) acc_global(P, P.aggs[a], &G.acc[(size_t)g * P.glob_words], t, p);
        }
    }
}

// The loads of one tile (R items per thread x the Spec's columns), ALL issued before anything is computed from them: the
// tags are derived in a second phase behind a scheduling barrier.  (Without it the compiler interleaved the first item's
// tag arithmetic with the loads and waited for the first item's data — s_waitcnt vmcnt(0) — before it had issued the
// second item's loads: two memory latencies per tile instead of one, config 2's scan 237 -> 283 us.)
// Issue: R items per thread x the Spec's columns, every load unconditional — an item beyond the tile re-reads the first item
// (of the segment) and is dropped through valid[] — and NOTHING is computed from what was loaded: `tt` holds the tag bytes as
// they come (WIDE: two rows' tags in one 16-bit load), `pv` payloads / codes.  A tile issued this way can stay in flight while
// another one is processed: with predicated loads the compiler cannot count what is outstanding and waits for everything
// (s_waitcnt vmcnt(0)), and any arithmetic on a loaded value is a wait at that spot.
template <class Spec, int R, int BLOCK, bool WIDE>
N1K_DEV void spec_issue_tile(const FastArgs& F, uint32_t base, uint32_t nitems, uint32_t tid, uint32_t (&tt)[R][kFastCols],
                             uint64_t (&pv)[R][WIDE ? 2 : 1][kSpecCols], bool (&valid)[R],
                             uint32_t item0 = 0) {  // item0: first item of the segment the tile belongs to (segmented batches)
#pragma unroll
    for (int j = 0; j < R; j++) {
        const uint32_t il = base + (uint32_t)j * BLOCK + tid;
        valid[j] = il < nitems;
        const uint32_t i = (valid[j] ? il : 0u) + item0;
#pragma unroll
        for (int c = 0; c < kFastCols; c++) {
            tt[j][c] = 0;
            if (c < Spec::ncols) {
                if (Spec::col_kind[c] == COLK_DICT32) {
                    if (WIDE) {
                        typedef uint32_t n1k_u32x2 __attribute__((ext_vector_type(2)));
                        const n1k_u32x2 cc = __builtin_nontemporal_load((const n1k_u32x2*)F.cols[c].codes + i);
                        pv[j][0][c] = cc.x;
                        pv[j][WIDE ? 1 : 0][c] = cc.y;
                    } else {
                        pv[j][0][c] = __builtin_nontemporal_load(F.cols[c].codes + i);
                    }
                } else {
                    if (WIDE) {
                        typedef unsigned long long n1k_u64x2 __attribute__((ext_vector_type(2)));
                        const n1k_u64x2 pp = __builtin_nontemporal_load((const n1k_u64x2*)F.cols[c].payload + i);
                        tt[j][c] = (uint32_t)__builtin_nontemporal_load((const uint16_t*)F.cols[c].tags + i);
                        pv[j][0][c] = pp.x;
                        pv[j][WIDE ? 1 : 0][c] = pp.y;
                    } else {
                        pv[j][0][c] = __builtin_nontemporal_load(F.cols[c].payload + i);
                        tt[j][c] = (uint32_t)__builtin_nontemporal_load(F.cols[c].tags + i);
                    }
                }
            }
        }
    }
}

// "The tile has arrived": an empty statement every register spec_issue_tile loaded passes through.  Placed right behind the
// issue of the NEXT tile — straight-line code — it makes the compiler wait there with a count (everything but the loads just
// issued); met first inside the rows' divergent code, the same registers cost a wait for everything outstanding.
template <class Spec, int R, bool WIDE>
N1K_DEV void spec_tile_arrived(uint32_t (&tt)[R][kFastCols], uint64_t (&pv)[R][WIDE ? 2 : 1][kSpecCols]) {
#pragma unroll
    for (int j = 0; j < R; j++) {
#pragma unroll
        for (int c = 0; c < kFastCols; c++) {
            if (c >= Spec::ncols) continue;
            if (Spec::col_kind[c] != COLK_DICT32) asm volatile("" : "+v"(tt[j][c]));
#pragma unroll
            for (int h = 0; h < (WIDE ? 2 : 1); h++) asm volatile("" : "+v"(pv[j][h][c]));
        }
    }
}

// Decode: the rows' tags from what spec_issue_tile loaded, and the fused arithmetic nodes (expression/arith_*.go,
// func_num.go): column slot ncols + d from the slots before it, in registers — no derived column in HBM (the element-wise
// arith_kernel writes 9 B per row and node and the scan reads them back).
template <class Spec, int R, bool WIDE>
N1K_DEV void spec_decode_tile(const FastArgs& F, const uint32_t (&tt)[R][kFastCols], uint64_t (&pv)[R][WIDE ? 2 : 1][kSpecCols],
                              uint32_t (&tg)[R][WIDE ? 2 : 1][kSpecCols]) {
    constexpr uint32_t kRowsPerItem = WIDE ? 2u : 1u;
#pragma unroll
    for (int j = 0; j < R; j++) {
#pragma unroll
        for (int c = 0; c < kSpecCols; c++) {
#pragma unroll
            for (int h = 0; h < (int)kRowsPerItem; h++) {
                if (c >= Spec::ncols) {
                    tg[j][h][c] = T_MISSING;
                    pv[j][h][c] = 0;
                } else if (Spec::col_kind[c] == COLK_DICT32) {
                    const uint32_t code = (uint32_t)pv[j][h][c];
                    tg[j][h][c] = code == 0xFFFFFFFFu ? (uint32_t)T_MISSING : (code == 0xFFFFFFFEu ? (uint32_t)T_NULL : (uint32_t)T_STRING);
                } else {
                    tg[j][h][c] = WIDE ? (h == 0 ? (tt[j][c] & 255u) : (tt[j][c] >> 8)) : tt[j][c];
                }
            }
        }
#pragma unroll
        for (int d = 0; d < Spec::nderived; d++) {
#pragma unroll
            for (int h = 0; h < (int)kRowsPerItem; h++) {
                uint32_t ot[4] = {T_MISSING, T_MISSING, T_MISSING, T_MISSING};
                uint64_t op[4] = {0, 0, 0, 0};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (k < (int)Spec::derived[d].nops) {
                        if (Spec::derived[d].ops[k].is_const) {
                            ot[k] = Spec::derived[d].ops[k].v;
                            op[k] = F.dconst[d][k];
                        } else {
                            ot[k] = tg[j][h][Spec::derived[d].ops[k].v];
                            op[k] = pv[j][h][Spec::derived[d].ops[k].v];
                        }
                    }
                }
                uint32_t rt;
                uint64_t rp;
                arith_apply(Spec::derived[d].op, Spec::derived[d].nops, ot, op, rt, rp);
                tg[j][h][Spec::ncols + d] = rt;
                pv[j][h][Spec::ncols + d] = rp;
            }
        }
    }
}

// Both at once (one tile in flight per workgroup: the loads are all issued before anything is computed from them — a
// scheduling barrier keeps the first item's tag arithmetic, and the wait it implies, from moving up between them).
template <class Spec, int R, int BLOCK, bool WIDE>
N1K_DEV void spec_load_tile(const FastArgs& F, uint32_t base, uint32_t nitems, uint32_t tid,
                            uint32_t (&tg)[R][WIDE ? 2 : 1][kSpecCols], uint64_t (&pv)[R][WIDE ? 2 : 1][kSpecCols], bool (&valid)[R],
                            uint32_t item0 = 0) {
    uint32_t tt[R][kFastCols];
    spec_issue_tile<Spec, R, BLOCK, WIDE>(F, base, nitems, tid, tt, pv, valid, item0);
    __builtin_amdgcn_sched_barrier(0);  // nothing below moves up between the loads
    spec_decode_tile<Spec, R, WIDE>(F, tt, pv, tg);
}

// ---- member words -> hash regions -------------------------------------------------------------------------------
//
// The member words of a DISTINCT aggregate leave the scan already partitioned by the first radix digit of part_hash(word):
// 256 hash regions, each split into kRecSubs sub-regions, one per workgroup label blockIdx.x % 8 (below: why).  A tile's
// words are ranked, staged in LDS in region order and written in runs (scatter_tile, n1k_scatter.h).  A word whose
// sub-region is full (many copies of few words) goes to the plain word log instead; n1k_finish then takes the exact path.
N1K_DEV void word_over_append(const WordLogArgs& L, int d, uint32_t b, uint64_t word, uint32_t* err_flags) {
    const uint32_t li = L.log_index[d];
    const unsigned long long q = atomicAdd(&L.over_cursor[li], 1ull);
    if (q < L.over_capacity) {
        L.over_word[d][q] = word;
        atomicAdd(&L.over_hist[(size_t)li * 256 + b], 1ull);
    } else
        atomicOr(err_flags, (uint32_t)ERR_TABLE_FULL);
}

// ---- records mode: Filter + group key of the plan shape, each surviving row leaves as a 16-byte record ----------
//
// High-cardinality GROUP BY (N1K_MODE_PARTITIONED): no workgroup table would absorb anything, so the specialised scan
// only evaluates the Filter, packs the group key and scatters (key, operand) records by the first radix digit of
// part_hash(key) into 256 hash regions — the projection and the first partition pass of the partitioned path in one
// kernel, reading the columns once and writing 16 bytes per surviving row (scatter_tile, n1k_scatter.h: a tile of
// BLOCK x kNW rows is ranked, staged in LDS in region order and written in runs).
//
// Every region is split into 8 sub-regions, one per workgroup label blockIdx.x % 8: workgroups are dealt round-robin
// over the 8 XCDs, whose L2s are private, so the partial cache lines at the ends of the runs — the next run of the same
// sub-region completes them — are merged in ONE L2 before they leave for HBM (with all workgroups appending to one
// tail the kernel wrote 1.5x its bytes).  Speed only: any placement gives the same records.  Sub-region (r, x) holds
// region_cap records at (r * 8 + x) * region_cap, its count at region_cursor[0][(r * 8 + x) * kCursorStride].

template <class Spec>
constexpr int spec_operand_col() {
    for (int a = 0; a < Spec::naggs; a++)
        if (Spec::aggs[a].has_operand) return (int)Spec::aggs[a].col;
    return -1;
}

N1K_DEV uint32_t rec16_region(const Rec16& r) { return part_hash(r.k & ~kRecIntFlag) >> 24; }

template <class Spec>
N1K_DEV void spec_row_record(const Program& P, const FastArgs& F, const uint32_t (&tg)[kSpecCols], const uint64_t (&pv)[kSpecCols],
                             uint32_t& selected, uint32_t& unpackable, Rec16& rec, uint32_t& bin) {
    bool pass = true;
#pragma unroll
    for (int t = 0; t < Spec::nterms; t++) pass = pass && spec_term_true<Spec>(t, F, tg[Spec::terms[t].col], pv[Spec::terms[t].col]);
    if (!pass) return;
    uint64_t key = 0;
#pragma unroll
    for (int k = 0; k < Spec::nkeys; k++) {
        uint64_t f = 0, canon;
        if (!pack_key_field(P, P.keys[k], tg[Spec::key_col[k]], pv[Spec::key_col[k]], f, canon)) {
            unpackable = 1;
            return;
        }
        key |= f << P.keys[k].shift;
    }
    selected++;
    constexpr int oc = spec_operand_col<Spec>();
    rec = rec16_encode(key, oc >= 0 ? tg[oc >= 0 ? oc : 0] : (uint32_t)T_NULL, oc >= 0 ? pv[oc >= 0 ? oc : 0] : 0ull);
    bin = rec16_region(rec);
}

template <class Spec, int R, int BLOCK, bool WIDE>
N1K_DEV void scan_spec_records_body(const Program& P, const FastArgs& F, const WordLogArgs& L) {
    constexpr uint32_t kRowsPerItem = WIDE ? 2u : 1u;
    constexpr int kNW = R * (int)kRowsPerItem;
    extern __shared__ uint64_t lds[];
    ScatterLds<Rec16, BLOCK, kNW>& S = *(ScatterLds<Rec16, BLOCK, kNW>*)lds;
    const uint32_t tid = threadIdx.x;
    scatter_init<BLOCK>(S.cnt);
    __syncthreads();
    const uint32_t sub = blockIdx.x % kRecSubs;
    unsigned long long* const cursor = L.region_cursor[0] + (size_t)sub * kCursorStride;
    Rec16* const dst = (Rec16*)L.region[0] + (size_t)sub * L.region_cap;
    uint32_t par = 0, unpackable = 0, selected = 0;
    const uint32_t nrows = F.nrows;
    const uint32_t nitems = WIDE ? (nrows + 1u) / 2u : nrows;
    const uint32_t tile = BLOCK * R;
    uint32_t* const flag = L.rec_overflow;
    auto process = [&](uint32_t (&tt)[R][kFastCols], uint64_t (&pv)[R][kRowsPerItem][kSpecCols], const bool (&valid)[R],
                       uint32_t base) {
        spec_tile_arrived<Spec, R, WIDE>(tt, pv);
        uint32_t tg[R][kRowsPerItem][kSpecCols];
        spec_decode_tile<Spec, R, WIDE>(F, tt, pv, tg);
        Rec16 recs[kNW];
        uint32_t bins[kNW];
#pragma unroll
        for (int j = 0; j < R; j++) {
#pragma unroll
            for (int h = 0; h < (int)kRowsPerItem; h++) {
                const int at = j * (int)kRowsPerItem + h;
                recs[at].k = kEmptyKey;
                recs[at].v = 0;
                bins[at] = kScatterNone;
                const bool row_ok = valid[j] && (!WIDE || h == 0 || 2u * (base + (uint32_t)j * BLOCK + tid) + 1u < nrows);
                if (row_ok) spec_row_record<Spec>(P, F, tg[j][h], pv[j][h], selected, unpackable, recs[at], bins[at]);
            }
        }
        scatter_tile<BLOCK, kNW>(S, par, recs, bins, cursor, kRecSubs * kCursorStride, dst, (uint64_t)kRecSubs * L.region_cap,
                                 L.region_cap, [=](uint32_t, const Rec16&) { *(volatile uint32_t*)flag = 1u; });
        par ^= 1u;
    };
    // Two tiles in flight: the columns of tile t + 1 are requested (spec_issue_tile: loads only) before tile t is decoded and
    // its records go through LDS (three barriers and the stores to the regions).
    uint32_t ttA[R][kFastCols], ttB[R][kFastCols];
    uint64_t pvA[R][kRowsPerItem][kSpecCols], pvB[R][kRowsPerItem][kSpecCols];
    bool vA[R], vB[R];
    const uint32_t stride = gridDim.x * tile;
    uint32_t base = blockIdx.x * tile;
    if (base < nitems) {
        spec_issue_tile<Spec, R, BLOCK, WIDE>(F, base, nitems, tid, ttA, pvA, vA);
        spec_tile_arrived<Spec, R, WIDE>(ttA, pvA);  // (nothing in flight at the loop's head on the way in)
    }
    while (base < nitems) {
        spec_issue_tile<Spec, R, BLOCK, WIDE>(F, base + stride, nitems, tid, ttB, pvB, vB);  // (beyond the end: dropped)
        process(ttA, pvA, vA, base);
        base += stride;
        if (base >= nitems) break;
        spec_issue_tile<Spec, R, BLOCK, WIDE>(F, base + stride, nitems, tid, ttA, pvA, vA);
        process(ttB, pvB, vB, base);
        base += stride;
    }
    if (unpackable) atomicOr(F.err_flags, (uint32_t)ERR_UNPACKABLE_KEY);
    __shared__ unsigned int block_selected;
    if (tid == 0) block_selected = 0;
    __syncthreads();
    for (int off = 32; off > 0; off >>= 1) selected += __shfl_down(selected, off, 64);
    if ((tid & 63) == 0 && selected) atomicAdd(&block_selected, selected);
    __syncthreads();
    if (tid == 0 && block_selected) atomicAdd(F.rows_selected, (unsigned long long)block_selected);
}

template <class Spec, int R, int BLOCK, bool WIDE>
__global__ __launch_bounds__(BLOCK) void scan_spec_records_kernel(const Program P, const FastArgs F, const WordLogArgs L) {
    scan_spec_records_body<Spec, R, BLOCK, WIDE>(P, F, L);
}

// ---- partition mode: Filter + destination of the plan shape, the survivors' input columns leave in destination order
//
// The multi-GPU row exchange (n1k_exchange_rows; no reference analogue: the reference fans in through one in-memory
// queue, execution/exchange.go:161-251).  Same contract as the interpreting partition_kernel (n1k_kernels.hip): row r of
// destination d = hash of the group key VALUES; survivors appended to d's region, one count per region.  Here the tile's
// survivors are ranked per destination (wave ballots, one LDS atomic per wave and destination, ONE global atomic per tile
// and destination), staged in LDS in destination order, and every shipped column is then written in runs: consecutive
// lanes store consecutive rows of one destination (the interpreter kernel reads every column twice and stores row by row).
template <class Spec>
constexpr int spec_ncols_of(uint32_t kind) {
    int n = 0;
    for (int c = 0; c < Spec::ncols; c++) n += Spec::col_kind[c] == kind ? 1 : 0;
    return n;
}
template <class Spec>
constexpr int spec_col_slot(int c) {  // index of column c among the columns of its kind
    int n = 0;
    for (int i = 0; i < c; i++) n += Spec::col_kind[i] == Spec::col_kind[c] ? 1 : 0;
    return n;
}

template <class Spec, int TILE>
struct PartLds {
    uint64_t pay[spec_ncols_of<Spec>(COLK_TAGGED64) ? spec_ncols_of<Spec>(COLK_TAGGED64) : 1][TILE];
    uint32_t code[spec_ncols_of<Spec>(COLK_DICT32) ? spec_ncols_of<Spec>(COLK_DICT32) : 1][TILE];
    uint8_t tag[spec_ncols_of<Spec>(COLK_TAGGED64) ? spec_ncols_of<Spec>(COLK_TAGGED64) : 1][TILE];
    uint8_t sdest[TILE];
    uint32_t cnt[2][64];  // survivors per destination in this tile (the other copy is zeroed for the next tile)
    uint32_t pre[64];     // first staged position of the destination
    unsigned long long gbase[64];  // first row of the tile's run in the destination's region
    // PartArgs::per_dest, per destination: rows a sub-region holds, and where the region's arrays start (payload / codes, then
    // tags, of every column) — worked out once per workgroup, not per row
    uint32_t dsub[64];
    uint64_t doff[2 * (spec_ncols_of<Spec>(COLK_TAGGED64) + spec_ncols_of<Spec>(COLK_DICT32)) ? 2 * (spec_ncols_of<Spec>(COLK_TAGGED64) + spec_ncols_of<Spec>(COLK_DICT32)) : 1][64];
    uint32_t total;
};

// PIPE: two tiles in flight (the next tile's columns requested before this one goes through LDS) — or one, with fewer
// registers and more workgroups per CU (measured, 100 M rows of config 2: 512 threads x 2 per CU piped 0.58 ms, 256 x 5
// piped 0.49 ms, 256 x 6 unpiped 0.43 ms).
template <class Spec, int R, int BLOCK, bool WIDE, bool PIPE = true>
N1K_DEV void scan_spec_partition_body(const Program& P, const FastArgs& F, const PartArgs& A) {
    constexpr uint32_t kRowsPerItem = WIDE ? 2u : 1u;
    constexpr int kNW = R * (int)kRowsPerItem;
    constexpr int TILE = BLOCK * kNW;
    __shared__ PartLds<Spec, TILE> S;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    for (uint32_t i = tid; i < 128; i += BLOCK) (&S.cnt[0][0])[i] = 0;
    if (tid < 64 && A.per_dest) {
        const uint64_t capd = A.dest_cap[tid];
        S.dsub[tid] = (uint32_t)(capd / (A.nsub > 1u ? A.nsub : 1u));
        uint64_t off = A.hdr_bytes;
#pragma unroll
        for (int c = 0; c < Spec::ncols; c++) {
            S.doff[2 * c][tid] = off;
            if (Spec::col_kind[c] == COLK_DICT32) off = part_region_next(off, capd, 4);
            else {
                off = part_region_next(off, capd, 8);
                S.doff[2 * c + 1][tid] = off;
                off = part_region_next(off, capd, 1);
            }
        }
    }
    __syncthreads();
    uint32_t par = 0, unpackable = 0;
    const uint32_t nrows = F.nrows;
    const uint32_t nitems = WIDE ? (nrows + 1u) / 2u : nrows;
    const uint32_t tile = BLOCK * R;
    const uint32_t nparts = A.nparts;
    const uint32_t cstride = A.count_stride ? A.count_stride : 1u;
    // this workgroup's sub-region of every destination (PartArgs::nsub): its counter and its first row
    const uint32_t nsub = A.nsub > 1u ? A.nsub : 1u;
    const uint32_t sub = blockIdx.x % nsub;
    const uint64_t sub_cap = nsub > 1u ? A.sub_rows : A.capacity;
    const uint64_t sub_first = (uint64_t)sub * A.sub_rows * (nsub > 1u ? 1u : 0u);
    unsigned long long* const my_counts = A.counts + (size_t)sub * kCursorStride * (nsub > 1u ? 1u : 0u);
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    auto process = [&](uint32_t (&tt)[R][kFastCols], uint64_t (&pv)[R][kRowsPerItem][kSpecCols], const bool (&valid)[R], uint32_t base,
                       bool piped) {
        __builtin_amdgcn_sched_barrier(0);  // (nothing of the decode moves up between the loads of a tile issued just before)
        if (piped) spec_tile_arrived<Spec, R, WIDE>(tt, pv);  // (another tile in flight: wait for this one here, with a count)
        uint32_t tg[R][kRowsPerItem][kSpecCols];
        spec_decode_tile<Spec, R, WIDE>(F, tt, pv, tg);
        bool pass[kNW];
        uint32_t dest[kNW], rk[kNW];
#pragma unroll
        for (int j = 0; j < R; j++) {
#pragma unroll
            for (int h = 0; h < (int)kRowsPerItem; h++) {
                const int at = j * (int)kRowsPerItem + h;
                bool ok = valid[j] && (!WIDE || h == 0 || 2u * (base + (uint32_t)j * BLOCK + tid) + 1u < nrows);
#pragma unroll
                for (int t = 0; t < Spec::nterms; t++) ok = ok && spec_term_true<Spec>(t, F, tg[j][h][Spec::terms[t].col], pv[j][h][Spec::terms[t].col]);
                // the destination is a function of the key VALUES (the same function as partition_kernel's)
                uint64_t key = 0;
#pragma unroll
                for (int k = 0; k < Spec::nkeys; k++) {
                    uint64_t f = 0, canon = 0;
                    if (ok && !pack_key_field(P, P.keys[k], tg[j][h][Spec::key_col[k]], pv[j][h][Spec::key_col[k]], f, canon)) {
                        unpackable = 1;
                        ok = false;
                    }
                    key = mix64(key ^ canon) + (uint64_t)k;
                }
                pass[at] = ok;
                dest[at] = ok ? (uint32_t)(((key >> 32) * (uint64_t)nparts) >> 32) : 0u;
                rk[at] = 0;
            }
        }
        // ranks: per destination the wave's survivors in (item, lane) order; lane d keeps the wave's total for d
        uint32_t mytotal = 0;
        for (uint32_t d = 0; d < nparts; d++) {
            uint32_t run = 0;
#pragma unroll
            for (int at = 0; at < kNW; at++) {
                const bool mine = pass[at] && dest[at] == d;
                const unsigned long long m = __ballot(mine);
                if (mine) rk[at] = run + (uint32_t)__popcll(m & lt_mask);
                run += (uint32_t)__popcll(m);
            }
            if (lane == d) mytotal = run;
        }
        uint32_t mybase = 0;
        if (mytotal) mybase = atomicAdd(&S.cnt[par][lane], mytotal);
#pragma unroll
        for (int at = 0; at < kNW; at++) rk[at] += (uint32_t)__shfl((int)mybase, (int)dest[at], 64);
        lds_barrier();
        unsigned long long gb = 0;
        if (tid < 64) {
            const uint32_t c = tid < nparts ? S.cnt[par][tid] : 0u;
            if (c) gb = atomicAdd(&my_counts[(size_t)tid * cstride], (unsigned long long)c);
            uint32_t incl = c;
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t t = __shfl_up(incl, off, 64);
                if ((int)tid >= off) incl += t;
            }
            S.pre[tid] = incl - c;
            S.cnt[par ^ 1u][tid] = 0;
            if (tid == 63) S.total = incl;
        }
        lds_barrier();
#pragma unroll
        for (int j = 0; j < R; j++) {
#pragma unroll
            for (int h = 0; h < (int)kRowsPerItem; h++) {
                const int at = j * (int)kRowsPerItem + h;
                if (!pass[at]) continue;
                const uint32_t q = S.pre[dest[at]] + rk[at];
                S.sdest[q] = (uint8_t)dest[at];
#pragma unroll
                for (int c = 0; c < Spec::ncols; c++) {
                    if (Spec::col_kind[c] == COLK_DICT32) S.code[spec_col_slot<Spec>(c)][q] = (uint32_t)pv[j][h][c];
                    else {
                        S.pay[spec_col_slot<Spec>(c)][q] = pv[j][h][c];
                        S.tag[spec_col_slot<Spec>(c)][q] = (uint8_t)tg[j][h][c];
                    }
                }
            }
        }
        if (tid < 64) S.gbase[tid] = gb;
        lds_barrier();
        const uint32_t staged = S.total;
        for (uint32_t q = tid; q < staged; q += BLOCK) {
            const uint32_t d = S.sdest[q];
            const unsigned long long r = S.gbase[d] + (q - S.pre[d]);
            if (r >= (A.per_dest ? (uint64_t)S.dsub[d] : sub_cap)) {  // (per_dest: this destination's own capacity)
                if (!(atomicOr(A.err_flags, (uint32_t)ERR_TABLE_FULL) & ERR_TABLE_FULL) && A.region_bytes)
                    for (uint32_t z = 0; z < nparts; z++)  // every receiver reads the verdict in the header it gets
                        atomicOr(&A.counts[(size_t)z * A.count_stride + 1], 1ull);
                continue;
            }
            if (A.per_dest) {
                char* const reg = (char*)A.counts + (size_t)d * A.region_bytes;
                const uint64_t at = (uint64_t)sub * S.dsub[d] + r;
#pragma unroll
                for (int c = 0; c < Spec::ncols; c++) {
                    if (Spec::col_kind[c] == COLK_DICT32) ((uint32_t*)(reg + S.doff[2 * c][d]))[at] = S.code[spec_col_slot<Spec>(c)][q];
                    else {
                        ((uint64_t*)(reg + S.doff[2 * c][d]))[at] = S.pay[spec_col_slot<Spec>(c)][q];
                        ((uint8_t*)(reg + S.doff[2 * c + 1][d]))[at] = S.tag[spec_col_slot<Spec>(c)][q];
                    }
                }
                continue;
            }
            const size_t shift = (size_t)d * A.region_bytes;  // (0 without packed regions)
            const uint64_t pos = A.region_bytes ? sub_first + r : (uint64_t)d * A.capacity + r;
#pragma unroll
            for (int c = 0; c < Spec::ncols; c++) {
                if (Spec::col_kind[c] == COLK_DICT32) ((uint32_t*)((char*)A.out_codes[c] + shift))[pos] = S.code[spec_col_slot<Spec>(c)][q];
                else {
                    ((uint64_t*)((char*)A.out_payload[c] + shift))[pos] = S.pay[spec_col_slot<Spec>(c)][q];
                    ((uint8_t*)((char*)A.out_tags[c] + shift))[pos] = S.tag[spec_col_slot<Spec>(c)][q];
                }
            }
        }
        par ^= 1u;
        // (the next tile writes pre / gbase / total / the staging arrays only behind its first barrier, which every wave
        //  reaches after it has left this loop)
    };
    if constexpr (!PIPE) {
        for (uint32_t base = blockIdx.x * tile; base < nitems; base += gridDim.x * tile) {
            uint32_t tt[R][kFastCols];
            uint64_t pv[R][kRowsPerItem][kSpecCols];
            bool valid[R];
            spec_issue_tile<Spec, R, BLOCK, WIDE>(F, base, nitems, tid, tt, pv, valid);
            process(tt, pv, valid, base, false);
        }
        if (unpackable) atomicOr(A.err_flags, (uint32_t)ERR_UNPACKABLE_KEY);
        return;
    }
    uint32_t ttA[R][kFastCols], ttB[R][kFastCols];
    uint64_t pvA[R][kRowsPerItem][kSpecCols], pvB[R][kRowsPerItem][kSpecCols];
    bool vA[R], vB[R];
    const uint32_t stride = gridDim.x * tile;
    uint32_t base = blockIdx.x * tile;
    if (base < nitems) spec_issue_tile<Spec, R, BLOCK, WIDE>(F, base, nitems, tid, ttA, pvA, vA);
    while (base < nitems) {
        spec_issue_tile<Spec, R, BLOCK, WIDE>(F, base + stride, nitems, tid, ttB, pvB, vB);  // (beyond the end: dropped)
        process(ttA, pvA, vA, base, true);
        base += stride;
        if (base >= nitems) break;
        spec_issue_tile<Spec, R, BLOCK, WIDE>(F, base + stride, nitems, tid, ttA, pvA, vA);
        process(ttB, pvB, vB, base, true);
        base += stride;
    }
    if (unpackable) atomicOr(A.err_flags, (uint32_t)ERR_UNPACKABLE_KEY);
}

// SEG: the batch is segmented (FastArgs::nseg > 1; run-time-built kernels only: the receiving side of the row exchange) —
// a compile-time switch, so that plain batches carry none of the segment bookkeeping (its scalar registers alone cost the
// config-2 scan 12 %: 16 SGPR spills instead of 2).
template <class Spec, int R, int BLOCK, bool WIDE, bool SEG = false>
N1K_DEV void scan_spec_body(const Program& P, const FastArgs& F, const GlobalTable& G, unsigned long long* ngroups,
                            const WordLogArgs& L) {
    extern __shared__ uint64_t lds[];
    __shared__ uint32_t lds_fill;
    const uint32_t S = F.lds_slots;
    const uint32_t tid = threadIdx.x;
    constexpr uint32_t kRowsPerItem = WIDE ? 2u : 1u;
    // COUNT(DISTINCT): scratch of the word scatter (nothing when the shape has no DISTINCT aggregate)
    constexpr int kND = spec_ndistinct<Spec>();
    constexpr int kNW = R * (int)kRowsPerItem;  // member words a thread can produce per tile and aggregate
    __shared__ ScatterLds<uint64_t, BLOCK, kNW> w_lds[kND ? kND : 1];
    uint32_t w_par = 0;
    const uint32_t w_sub = blockIdx.x % kRecSubs;  // sub-region of every hash region this workgroup appends to
    uint64_t* dcache = lds + (size_t)S * P.lds_words;  // "already logged" caches of the DISTINCT aggregates
    if constexpr (kND > 0) {
        for (uint32_t i = tid; i < L.dcache_slots * (uint32_t)kND; i += BLOCK) *(volatile lds_u64*)lds_word(dcache, i) = kEmptyKey;
#pragma unroll
        for (int d = 0; d < kND; d++) scatter_init<BLOCK>(w_lds[d].cnt);
    }
    lds_table_init<BLOCK>(P, lds, S, tid);
    if (tid == 0) lds_fill = 0;
    __syncthreads();

    uint32_t unpackable = 0, selected = 0;
    // (the engine passes an even row count to WIDE launches; a row count that lives on the device may be odd: the second
    //  row of the last item is then masked)
    const uint32_t nrows = F.nrows_dev ? (uint32_t)(*F.nrows_dev < (unsigned long long)F.nrows ? *F.nrows_dev : F.nrows) : F.nrows;
    const uint32_t nitems = WIDE ? (nrows + 1u) / 2u : nrows;
    const uint32_t tile = BLOCK * R;
    // Segmented batch (FastArgs::nseg > 1): tiles are numbered segment by segment, `tps` per segment (its capacity); the
    // segments' row counts are read once into LDS.  A plain batch is one segment.
    const uint32_t nseg = SEG ? (F.nseg > 1u ? F.nseg : 1u) : 1u;
    uint32_t seg_n[SEG ? kMaxSegments : 1];  // (wave-uniform: scalar registers)
    if constexpr (SEG) {
#pragma unroll
        for (int i = 0; i < (int)kMaxSegments; i++) {
            seg_n[i] = 0;
            if (nseg > 1u && (uint32_t)i < nseg) {
                const unsigned long long c = F.seg_counts[(size_t)i * F.seg_count_stride];
                seg_n[i] = (uint32_t)(c < (unsigned long long)F.seg_rows ? c : F.seg_rows);
            }
        }
    }
    const uint32_t seg_items = WIDE ? F.seg_rows / 2u : F.seg_rows;
    const uint32_t tps = nseg > 1u ? (seg_items + tile - 1u) / tile : (nitems + tile - 1u) / tile;
    const uint32_t total_tiles = nseg * tps;
    // tile t -> first item inside its segment, the segment's items / rows, the segment's first item in the columns
    auto locate = [&](uint32_t t, uint32_t& base, uint32_t& ni, uint32_t& nr, uint32_t& item0) -> bool {
        base = 0; ni = 0; nr = 0; item0 = 0;
        if (t >= total_tiles) return false;
        if (!SEG || nseg == 1u) {
            base = t * tile; ni = nitems; nr = nrows;
            return true;
        }
        if constexpr (SEG) {
            const uint32_t seg = t / tps;
            base = (t - seg * tps) * tile;
#pragma unroll
            for (int i = 0; i < (int)kMaxSegments; i++)
                if (seg == (uint32_t)i) nr = seg_n[i];
            ni = WIDE ? (nr + 1u) / 2u : nr;
            item0 = seg * seg_items;
        }
        return base < ni;  // (false: a tile of the segment's unused capacity)
    };

    auto process = [&](uint32_t (&tt)[R][kFastCols], uint64_t (&pv)[R][kRowsPerItem][kSpecCols], const bool (&valid)[R],
                       uint32_t base, uint32_t nr, bool piped) {
        __builtin_amdgcn_sched_barrier(0);  // (nothing of the decode moves up between the loads of a tile issued just before)
        // (another tile in flight: wait for this one here, with a count; else the compiler's own waits, item by item, do better)
        if (piped) spec_tile_arrived<Spec, R, WIDE>(tt, pv);
        uint32_t tg[R][kRowsPerItem][kSpecCols];
        spec_decode_tile<Spec, R, WIDE>(F, tt, pv, tg);
        uint64_t mw[kSpecDistinct][kNW];  // this thread's member words of the tile (kEmptyKey = none) and their hash regions
        uint32_t mb[kSpecDistinct][kNW];
#pragma unroll
        for (int j = 0; j < R; j++) {
#pragma unroll
            for (int h = 0; h < (int)kRowsPerItem; h++) {
                uint64_t words[kSpecDistinct];
                uint32_t bins[kSpecDistinct];
#pragma unroll
                for (int d = 0; d < (int)kSpecDistinct; d++) { words[d] = kEmptyKey; bins[d] = kScatterNone; }
                const bool row_ok = valid[j] && (!WIDE || h == 0 || 2u * (base + (uint32_t)j * BLOCK + tid) + 1u < nr);
                if (row_ok) spec_row<Spec>(P, F, G, ngroups, lds, S, &lds_fill, tg[j][h], pv[j][h], selected, unpackable, L, dcache, words, bins);
#pragma unroll
                for (int d = 0; d < (int)kSpecDistinct; d++) { mw[d][j * (int)kRowsPerItem + h] = words[d]; mb[d][j * (int)kRowsPerItem + h] = bins[d]; }
            }
        }
        if constexpr (kND > 0) {
#pragma unroll
            for (int d = 0; d < kND; d++) {
                if (L.pad & 2u) continue;  // (timing experiments only: words dropped)
                uint32_t* const ef = F.err_flags;
                scatter_tile<BLOCK, kNW>(w_lds[d], w_par, mw[d], mb[d], L.region_cursor[d] + (size_t)w_sub * kCursorStride,
                                         kRecSubs * kCursorStride, L.region[d] + (size_t)w_sub * L.region_cap,
                                         (uint64_t)kRecSubs * L.region_cap, L.region_cap,
                                         [&, d, ef](uint32_t b, uint64_t w) { word_over_append(L, d, b, w, ef); });
            }
            w_par ^= 1u;
        }
    };
    if constexpr (!SEG) {
        // plain batch: tiles at base = blockIdx.x * tile, + gridDim.x * tile, ...
        const uint32_t stride = gridDim.x * tile;
        if constexpr (kND > 0 || (N1K_SPEC_PIPE_ALL && Spec::nderived == 0)) {  // (arithmetic nodes in registers: one tile, or the registers run out)
            // Two tiles in flight: the columns of tile t + 1 are requested before tile t's member words go through LDS (three
            // barriers and the stores to the regions), so the loads' latency hides behind that instead of adding to it.
            uint32_t ttA[R][kFastCols], ttB[R][kFastCols];
            uint64_t pvA[R][kRowsPerItem][kSpecCols], pvB[R][kRowsPerItem][kSpecCols];
            bool vA[R], vB[R];
            uint32_t base = blockIdx.x * tile;
            if (base < nitems) spec_issue_tile<Spec, R, BLOCK, WIDE>(F, base, nitems, tid, ttA, pvA, vA);
            while (base < nitems) {
                spec_issue_tile<Spec, R, BLOCK, WIDE>(F, base + stride, nitems, tid, ttB, pvB, vB);  // (beyond the end: dropped)
                process(ttA, pvA, vA, base, nrows, true);
                base += stride;
                if (base >= nitems) break;
                spec_issue_tile<Spec, R, BLOCK, WIDE>(F, base + stride, nitems, tid, ttA, pvA, vA);
                process(ttB, pvB, vB, base, nrows, true);
                base += stride;
            }
        } else {
            for (uint32_t base = blockIdx.x * tile; base < nitems; base += stride) {
                // issue every load of the tile first (R items x ncols columns), then compute
                uint32_t tt[R][kFastCols];
                uint64_t pv[R][kRowsPerItem][kSpecCols];
                bool valid[R];
                spec_issue_tile<Spec, R, BLOCK, WIDE>(F, base, nitems, tid, tt, pv, valid);
                process(tt, pv, valid, base, nrows, false);
            }
        }
    } else {
        // segmented batch: tiles numbered segment by segment (locate)
        const uint32_t gstride = gridDim.x;
        if constexpr (kND > 0 || (N1K_SPEC_PIPE_ALL && Spec::nderived == 0)) {  // (arithmetic nodes in registers: one tile, or the registers run out)
            // Two tiles in flight: the columns of tile t + 1 are requested before tile t's member words go through LDS (three
            // barriers and the stores to the regions), so the loads' latency hides behind that instead of adding to it.
            uint32_t ttA[R][kFastCols], ttB[R][kFastCols];
            uint64_t pvA[R][kRowsPerItem][kSpecCols], pvB[R][kRowsPerItem][kSpecCols];
            bool vA[R], vB[R];
            uint32_t baseA = 0, baseB = 0, nrA = 0, nrB = 0, ni, item0;
            uint32_t t = blockIdx.x;
            // (a tile that does not exist, or lies in unused capacity: nitems 0 -> no loads, every row invalid)
            if (t < total_tiles) {
                const bool ok = locate(t, baseA, ni, nrA, item0);
                spec_issue_tile<Spec, R, BLOCK, WIDE>(F, baseA, ok ? ni : 0u, tid, ttA, pvA, vA, item0);
            }
            while (t < total_tiles) {
                {
                    const bool ok = locate(t + gstride, baseB, ni, nrB, item0);
                    spec_issue_tile<Spec, R, BLOCK, WIDE>(F, baseB, ok ? ni : 0u, tid, ttB, pvB, vB, item0);
                }
                process(ttA, pvA, vA, baseA, nrA, true);
                t += gstride;
                if (t >= total_tiles) break;
                {
                    const bool ok = locate(t + gstride, baseA, ni, nrA, item0);
                    spec_issue_tile<Spec, R, BLOCK, WIDE>(F, baseA, ok ? ni : 0u, tid, ttA, pvA, vA, item0);
                }
                process(ttB, pvB, vB, baseB, nrB, true);
                t += gstride;
            }
        } else {
            for (uint32_t t = blockIdx.x; t < total_tiles; t += gstride) {
                uint32_t base, ni, nr, item0;
                if (!locate(t, base, ni, nr, item0)) continue;
                // issue every load of the tile first (R items x ncols columns), then compute
                uint32_t tt[R][kFastCols];
                uint64_t pv[R][kRowsPerItem][kSpecCols];
                bool valid[R];
                spec_issue_tile<Spec, R, BLOCK, WIDE>(F, base, ni, tid, tt, pv, valid, item0);
                process(tt, pv, valid, base, nr, false);
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
                                                         unsigned long long* ngroups, const WordLogArgs L) {
    scan_spec_body<Spec, R, BLOCK, WIDE>(P, F, G, ngroups, L);
}

}  // namespace n1k
