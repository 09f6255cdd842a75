// n1k_kernels.hip — hand-written gfx950 kernels of the Filter -> Group -> Aggregate path.
//
//   scan_group_kernel   K1+K2+K3: predicate (expression/comp_*, logic_*), key packing
//                       (execution/group_util.go:18-35) and the InitialGroup hash aggregate
//                       (execution/group_initial.go:56-100 + algebra/agg_*.CumulateInitial) into an
//                       open-addressed LDS table per workgroup, then K4: atomic merge of the
//                       workgroup's table into the global table (≙ IntermediateGroup,
//                       execution/group_intermediate.go:56-104 + CumulateIntermediate).
//   finalize_kernel     K5: FinalGroup (execution/group_final.go:55-98 + ComputeFinal).
//   filter_* kernels    Filter alone (execution/filter.go:49-61): ballot bit mask, scan, ordered compaction.
//
// All of it is HBM-bound integer/byte work (no MFMA): wave64, coalesced column loads, LDS atomics,
// scalar (wave-uniform) interpretation of the plan program.
#include <hip/hip_runtime.h>
#include "n1k_device.h"
#include "n1k_tables.h"
#include "n1k_scatter.h"
#include "n1k_kernels.h"
#include <algorithm>
#include <vector>

namespace n1k {

// ------------------------------------------------------------------ operand / term evaluation

template <int R>
N1K_DEV void load_operand(const Program& P, const Operand& o, const uint64_t (&row)[R], const bool (&valid)[R],
                          uint32_t (&tag)[R], uint64_t (&pay)[R]) {
    if (o.is_const) {
#pragma unroll
        for (int j = 0; j < R; j++) {
            tag[j] = o.ctag;
            pay[j] = o.cpayload;
        }
        return;
    }
    const DevCol& c = P.cols[o.col];
    if (c.kind == COLK_DICT32) {
        const uint32_t* codes = c.codes;
#pragma unroll
        for (int j = 0; j < R; j++) {
            uint32_t code = valid[j] ? codes[row[j]] : 0xFFFFFFFFu;
            tag[j] = code == 0xFFFFFFFFu ? T_MISSING : (code == 0xFFFFFFFEu ? T_NULL : T_STRING);
            pay[j] = code;
        }
    } else {
        const uint8_t* tags = c.tags;
        const uint64_t* payload = c.payload;
#pragma unroll
        for (int j = 0; j < R; j++) {
            tag[j] = valid[j] ? (uint32_t)tags[row[j]] : (uint32_t)T_MISSING;
            pay[j] = valid[j] ? payload[row[j]] : 0ull;
        }
    }
}

// one predicate term for R rows -> 4-valued logic
template <int R>
N1K_DEV void eval_term(const Program& P, const Term& t, const uint64_t (&row)[R], const bool (&valid)[R],
                       uint32_t (&out)[R], uint32_t& unsupported) {
    uint32_t ta[R];
    uint64_t pa[R];
    load_operand<R>(P, t.a, row, valid, ta, pa);
    switch (t.op) {
        case TERM_EQ:
        case TERM_LT:
        case TERM_LE: {
            uint32_t tb[R];
            uint64_t pb[R];
            load_operand<R>(P, t.b, row, valid, tb, pb);
#pragma unroll
            for (int j = 0; j < R; j++) {
                if (t.op == TERM_EQ) {
                    out[j] = equals_l(ta[j], pa[j], tb[j], pb[j], &unsupported);
                } else {
                    int c = compare(ta[j], pa[j], tb[j], pb[j], P.str_rank, &unsupported);
                    out[j] = c == CMP_MISSING ? L_MISSING
                                              : (c == CMP_NULL ? L_NULL : ((t.op == TERM_LT ? c < 0 : c <= 0) ? L_TRUE : L_FALSE));
                }
            }
            break;
        }
        case TERM_BETWEEN: {  // expression/comp_between.go:58-78
            uint32_t tb[R], tc[R];
            uint64_t pb[R], pc[R];
            load_operand<R>(P, t.b, row, valid, tb, pb);
            load_operand<R>(P, t.c, row, valid, tc, pc);
#pragma unroll
            for (int j = 0; j < R; j++) {
                int lo = compare(ta[j], pa[j], tb[j], pb[j], P.str_rank, &unsupported);
                int hi = compare(ta[j], pa[j], tc[j], pc[j], P.str_rank, &unsupported);
                uint32_t r;
                if (lo == CMP_MISSING || hi == CMP_MISSING) r = L_MISSING;
                else if (lo == CMP_NULL || hi == CMP_NULL) r = L_NULL;
                else r = (lo >= 0 && hi <= 0) ? L_TRUE : L_FALSE;
                out[j] = r;
            }
            break;
        }
        case TERM_IS_NULL:
#pragma unroll
            for (int j = 0; j < R; j++) out[j] = ta[j] == T_NULL ? L_TRUE : (ta[j] == T_MISSING ? L_MISSING : L_FALSE);
            break;
        case TERM_IS_NOT_NULL:
#pragma unroll
            for (int j = 0; j < R; j++) out[j] = ta[j] == T_NULL ? L_FALSE : (ta[j] == T_MISSING ? L_MISSING : L_TRUE);
            break;
        case TERM_IS_MISSING:
#pragma unroll
            for (int j = 0; j < R; j++) out[j] = ta[j] == T_MISSING ? L_TRUE : L_FALSE;
            break;
        case TERM_IS_NOT_MISSING:
#pragma unroll
            for (int j = 0; j < R; j++) out[j] = ta[j] == T_MISSING ? L_FALSE : L_TRUE;
            break;
        case TERM_IS_VALUED:
#pragma unroll
            for (int j = 0; j < R; j++) out[j] = ta[j] <= T_NULL ? L_FALSE : L_TRUE;
            break;
        case TERM_IS_NOT_VALUED:
#pragma unroll
            for (int j = 0; j < R; j++) out[j] = ta[j] <= T_NULL ? L_TRUE : L_FALSE;
            break;
        case TERM_TRUTH:
#pragma unroll
            for (int j = 0; j < R; j++) out[j] = truth_l(ta[j], pa[j], P.empty_str_code, P.empty_arr_code, P.empty_obj_code);
            break;
        default: {  // TERM_NUM_*: a <op> NUMBER constant; same result as LT/LE/Eq.Apply with the operands in this order
            const uint32_t ct = t.b.ctag;
            const uint64_t cp = t.b.cpayload;
            const double cf = num_actual(ct, cp);
#pragma unroll
            for (int j = 0; j < R; j++) {
                uint32_t tg = ta[j];
                int c;  // collation of a against the constant
                if (tg == T_INT && ct == T_INT) {
                    int64_t x = (int64_t)pa[j], y = (int64_t)cp;
                    c = x < y ? -1 : (x > y ? 1 : 0);
                } else if (tg == T_INT || tg == T_FLOAT) {
                    c = collate_f64(num_actual(tg, pa[j]), cf);
                } else {
                    c = tg < T_INT ? -1 : 1;  // BOOLEAN sorts below NUMBER, STRING/ARRAY/OBJECT above
                }
                bool r = t.op == TERM_NUM_LT ? c < 0 : t.op == TERM_NUM_LE ? c <= 0 : t.op == TERM_NUM_GT ? c > 0
                         : t.op == TERM_NUM_GE ? c >= 0 : (c == 0 && (tg == T_INT || tg == T_FLOAT));
                out[j] = tg == T_MISSING ? L_MISSING : (tg == T_NULL ? L_NULL : (r ? L_TRUE : L_FALSE));
            }
            break;
        }
    }
}

// Filter.processItem (execution/filter.go:49-61): pass iff Condition().Evaluate(item).Truth()
template <int R>
N1K_DEV void eval_predicate(const Program& P, const uint64_t (&row)[R], const bool (&valid)[R], bool (&pass)[R],
                            uint32_t& unsupported) {
    uint64_t st[R];
#pragma unroll
    for (int j = 0; j < R; j++) st[j] = 0;
    for (uint32_t i = 0; i < P.nlogic; i++) {
        LogicOp op = P.logic[i];
        if (op.op == LOGIC_PUSH) {
            uint32_t l[R];
            eval_term<R>(P, P.terms[op.arg], row, valid, l, unsupported);
#pragma unroll
            for (int j = 0; j < R; j++) st[j] = (st[j] << 2) | l[j];
        } else if (op.op == LOGIC_AND) {
#pragma unroll
            for (int j = 0; j < R; j++) {
                uint32_t r = logic_and(st[j], op.arg);
                st[j] = (st[j] << 2) | r;
            }
        } else if (op.op == LOGIC_OR) {
#pragma unroll
            for (int j = 0; j < R; j++) {
                uint32_t r = logic_or(st[j], op.arg);
                st[j] = (st[j] << 2) | r;
            }
        } else {
#pragma unroll
            for (int j = 0; j < R; j++) st[j] = (st[j] & ~3ull) | logic_not((uint32_t)(st[j] & 3ull));
        }
    }
#pragma unroll
    for (int j = 0; j < R; j++) pass[j] = valid[j] && (P.nlogic == 0 || (st[j] & 3ull) == L_TRUE);
}

// ------------------------------------------------------------------ derived columns (arithmetic operands)
//
// Add/Mult.Apply (expression/arith_add.go:51-70, arith_mult.go:51-70): any MISSING operand -> MISSING, else any
// non-number -> NULL, else the fold from int 0 / int 1.  Sub/Div/Mod/IDiv/IMod/Neg: arith_sub.go:53-61,
// arith_div.go:46-64, arith_mod.go:48-66, arith_idiv.go:46-56, arith_imod.go, arith_neg.go:51-59.
__global__ void arith_kernel(const ArithArgs A) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.nrows) return;
    uint32_t tg[4];
    uint64_t pv[4];
    for (uint32_t k = 0; k < A.nops; k++) {
        const Operand& o = A.ops[k];
        if (o.is_const) {
            tg[k] = o.ctag;
            pv[k] = o.cpayload;
        } else {
            const DevCol& c = A.cols[o.col];
            if (c.kind == COLK_DICT32) {
                uint32_t code = c.codes[i];
                tg[k] = code == 0xFFFFFFFFu ? (uint32_t)T_MISSING : (code == 0xFFFFFFFEu ? (uint32_t)T_NULL : (uint32_t)T_STRING);
                pv[k] = code;
            } else {
                tg[k] = c.tags[i];
                pv[k] = c.payload[i];
            }
        }
    }
    uint32_t rt;
    uint64_t rp;
    if (A.op == AR_GREATEST || A.op == AR_LEAST) {
        // Greatest.Apply / Least.Apply (expression/func_comp.go:54-67, 124-138): the largest (smallest) argument above NULL
        // in value.Collate order, the first one on ties; NULL when there is none
        rt = T_NULL;
        rp = 0;
        uint32_t unsupported = 0;
        for (uint32_t k = 0; k < A.nops; k++) {
            if (tg[k] <= T_NULL) continue;
            if (rt == T_NULL) { rt = tg[k]; rp = pv[k]; continue; }
            const int c = collate(tg[k], pv[k], rt, rp, A.str_rank, &unsupported);
            if (A.op == AR_GREATEST ? c > 0 : c < 0) { rt = tg[k]; rp = pv[k]; }
        }
        if (unsupported) atomicOr(A.err_flags, (uint32_t)ERR_UNSUPPORTED_VALUE);
    } else
        arith_apply(A.op, A.nops, tg, pv, rt, rp);
    A.out_tags[i] = (uint8_t)rt;
    A.out_payload[i] = rp;
}

// ------------------------------------------------------------------ K6: DISTINCT sets
// (distinct_classify / member_word_bits / radix_bin live in n1k_tables.h: the specialised scan uses them too)
N1K_DEV bool member_word(const ScanArgs& A, uint64_t key, uint32_t cls, uint64_t val, uint64_t& word) {
    return member_word_bits(A.nw_key_bits, A.nw_val_bits, key, cls, val, word);
}

// Workgroup-wide reservation of `mine` consecutive entries per thread behind *cursor: returns this thread's first
// position.  All threads of the block call it; the caller synchronises before the scratch words are reused.
template <int BLOCK>
N1K_DEV unsigned long long tile_reserve(uint32_t mine, unsigned long long* cursor, uint32_t* wave_cnt,
                                        unsigned long long* tile_base, uint32_t tid) {
    uint32_t incl = mine;
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t t = __shfl_up(incl, off, 64);
        if ((int)(tid & 63) >= off) incl += t;
    }
    if ((tid & 63) == 63) wave_cnt[tid >> 6] = incl;
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (int w = 0; w < BLOCK / 64; w++) {
            uint32_t c = wave_cnt[w];
            wave_cnt[w] = run;
            run += c;
        }
        *tile_base = run ? atomicAdd(cursor, (unsigned long long)run) : 0ull;
    }
    __syncthreads();
    return *tile_base + wave_cnt[tid >> 6] + (incl - mine);
}

// read-only probe of the global table
N1K_DEV long long global_find(const GlobalTable& G, uint64_t key) {
    uint64_t mask = G.capacity - 1;
    uint64_t h = mix64(key) & mask;
    for (int probe = 0; probe < 8192; probe++) {
        uint64_t cur = G.keys[h];
        if (cur == key) return (long long)h;
        if (cur == kEmptyKey) return -1;
        h = (h + 1) & mask;
    }
    return -1;
}

// finish step 1: give every (group, class) of one DISTINCT aggregate an open-addressed region of
// next_pow2(2 * operands) words inside one big table
__global__ void distinct_layout_kernel(const Program P, const GlobalTable G, const DistinctArgs D) {
    uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= G.capacity) return;
    uint64_t* reg = D.regions + s * 6;
    uint64_t* w = &G.acc[s * P.glob_words + D.glob_off];
    bool used = G.keys[s] != kEmptyKey;
    w[0] = 0;  // the set size (and the sum of its members) is recomputed from the whole log at every finish
    w[4] = 0;
    if (D.kind == AGG_SUM || D.kind == AGG_AVG) { w[5] = 0; w[6] = 0; w[7] = 0; w[8] = 0; }
    for (uint32_t c = 0; c < 3; c++) {
        uint64_t n = used ? w[1 + c] : 0;
        uint64_t cap = 0;
        if (n) {
            cap = 2;
            while (cap < 2 * n) cap <<= 1;
        }
        uint64_t off = cap ? atomicAdd(D.total_words, (unsigned long long)cap) : 0ull;
        reg[2 * c] = off;
        reg[2 * c + 1] = cap ? cap - 1 : 0;
    }
}

// finish step 2: insert every logged pair into its group's value set with ONE 64-bit compare-and-swap per probe
// (the group and the class are implied by the region, so the value alone identifies the member: no multi-word
// entries, no locks, no spinning).  First insertions are counted per group in an LDS table of counters (when the
// group table is small enough) and added to the global rows once per workgroup.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void distinct_insert_kernel(const Program P, const GlobalTable G, const DistinctArgs D,
                                                               uint32_t* err_flags, uint32_t lds_counters) {
    extern __shared__ uint32_t cnt[];  // lds_counters entries (== G.capacity) or none
    for (uint32_t i = threadIdx.x; i < lds_counters; i += BLOCK) cnt[i] = 0;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < D.npairs; i += (uint64_t)gridDim.x * BLOCK) {
        long long g = global_find(G, D.log_key[i]);
        if (g < 0) {
            atomicOr(err_flags, (uint32_t)ERR_TABLE_FULL);
            continue;
        }
        uint32_t cls = D.log_cls[i];
        uint64_t val = D.log_val[i];
        unsigned long long* w = (unsigned long long*)&G.acc[(size_t)g * P.glob_words + D.glob_off];
        bool fresh = false;
        if (val == kEmptyKey) {
            // the one value that collides with the free marker (int -1): tracked by a flag bit per class
            unsigned long long bit = 1ull << cls;
            unsigned long long old = atomicOr(&w[4], bit);
            fresh = !(old & bit);
        } else {
            const uint64_t* reg = D.regions + (size_t)g * 6 + 2 * cls;
            uint64_t off = reg[0], mask = reg[1];
            uint64_t h = mix64(val) & mask;
            bool done = false;
            for (uint64_t probe = 0; probe <= mask && !done; probe++) {
                unsigned long long* slot = (unsigned long long*)&D.set_table[off + h];
                unsigned long long old = atomicCAS(slot, (unsigned long long)kEmptyKey, (unsigned long long)val);
                if (old == kEmptyKey) { fresh = true; done = true; }
                else if (old == val) done = true;  // already a member
                h = (h + 1) & mask;
            }
            if (!done) atomicOr(err_flags, (uint32_t)ERR_TABLE_FULL);
        }
        if (fresh) {  // Set.Len() grows by one (value/set.go:198-215)
            if (lds_counters) atomicAdd(&cnt[g], 1u);
            else atomicAdd(&w[0], 1ull);
            if (D.kind == AGG_SUM || D.kind == AGG_AVG) {
                // SumDistinct/AvgDistinct.ComputeFinal add up the set's members (algebra/agg_sum_distinct.go:113-133)
                if (cls == DC_INT) {
                    int64_t x = (int64_t)val;
                    atomicAdd(&w[5], (unsigned long long)(uint32_t)x);
                    atomicAdd(&w[6], (unsigned long long)(x >> 32));
                    atomicOr(&w[8], x < 0 ? (unsigned long long)SF_NEG_INT : (unsigned long long)SF_NONNEG_INT);
                } else {
                    atomicAdd((double*)&w[7], as_f64(val));
                    atomicOr(&w[8], (unsigned long long)SF_FLOAT);
                }
            }
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < lds_counters; i += BLOCK)
        if (cnt[i]) atomicAdd((unsigned long long*)&G.acc[(size_t)i * P.glob_words + D.glob_off], (unsigned long long)cnt[i]);
}

// ------------------------------------------------------------------ COUNT(DISTINCT) over member words
//
// value.Set membership (value/set.go:22-110) for the pairs that fit one 64-bit word.  Random compare-and-swap into a
// table in HBM tops out near 20 G inserts/s on gfx950 whatever the table size (tools/ubench_cas.hip: 2 GB, 128 MB and
// 2 MB windows all land at 4.6-5.4 ms per 100 M), so the set is built in LDS instead: the word log is radix
// partitioned on bits of mix64(word) (one or two passes of 256 bins, LDS-staged so that every bin is written in
// runs) until a bin's distinct words fit an LDS open-addressed set, then one workgroup de-duplicates each bin and
// counts the new members per group.  Equal words always meet in the same bin, so the bins are independent.
constexpr int kRadixBlock = 512, kRadixPer = 16, kRadixTile = kRadixBlock * kRadixPer;  // 8192 words = 64 KB staged


// slice of segment `s` that workgroup blockIdx.x owns (the same in the histogram and the scatter pass)
N1K_DEV void radix_slice(const RadixArgs& A, uint32_t s, uint64_t& lo, uint64_t& hi) {
    uint64_t s0, s1;
    if (A.seg_count) {  // hash regions of fixed capacity
        const uint64_t c = A.seg_count[(size_t)s * kCursorStride];
        s0 = (uint64_t)s * A.seg_stride;
        s1 = s0 + (c < A.seg_stride ? c : A.seg_stride);
    } else {
        s0 = A.seg_start[s];
        s1 = A.seg_start[s + 1];
    }
    uint64_t chunk = (s1 - s0 + gridDim.x - 1) / gridDim.x;
    chunk = (chunk + kRadixTile - 1) / kRadixTile * kRadixTile;
    lo = s0 + (uint64_t)blockIdx.x * chunk;
    hi = lo + chunk < s1 ? lo + chunk : s1;
    if (lo > s1) lo = hi = s1;
}

__global__ __launch_bounds__(kRadixBlock) void radix_hist_kernel(const RadixArgs A) {
    __shared__ uint32_t h[256];
    const uint32_t tid = threadIdx.x, s = blockIdx.y;
    if (tid < 256) h[tid] = 0;
    __syncthreads();
    uint64_t lo, hi;
    radix_slice(A, s, lo, hi);
    for (uint64_t i = lo + tid; i < hi; i += kRadixBlock) atomicAdd(&h[radix_bin(A.src[i], A.shift)], 1u);
    __syncthreads();
    if (tid < 256 && h[tid]) atomicAdd(&A.hist[(size_t)s * 256 + tid], (unsigned long long)h[tid]);
}

// one workgroup per segment: exclusive scan of its 256 bin counts -> write cursors and the finer segment starts
__global__ __launch_bounds__(256) void radix_offsets_kernel(const RadixArgs A) {
    __shared__ unsigned long long sc[256];
    const uint32_t s = blockIdx.x, t = threadIdx.x;
    const unsigned long long v = A.hist[(size_t)s * 256 + t];
    sc[t] = v;
    __syncthreads();
    for (uint32_t off = 1; off < 256; off <<= 1) {
        unsigned long long x = t >= off ? sc[t - off] : 0ull;
        __syncthreads();
        sc[t] += x;
        __syncthreads();
    }
    const unsigned long long excl = sc[t] - v + A.seg_start[s];
    A.cursor[((size_t)s * 256 + t) * (A.cursor_stride ? A.cursor_stride : 1u)] = excl;
    A.out_start[(size_t)s * 256 + t] = excl;
    if (s == A.nseg - 1 && t == 255) A.out_start[(size_t)A.nseg * 256] = A.seg_start[A.nseg];
}

template <bool RECORDS>
__global__ __launch_bounds__(kRadixBlock) void radix_scatter_kernel(const RadixArgs A) {
    __shared__ uint64_t stage[kRadixTile];
    __shared__ uint8_t sbin[RECORDS ? kRadixTile : 1];  // bin of every staged position (records only)
    __shared__ uint32_t cnt[256], pre[256], wsum[4];
    __shared__ unsigned long long gbase[256];
    const uint32_t tid = threadIdx.x, s = blockIdx.y, cstride = A.cursor_stride ? A.cursor_stride : 1u;
    uint64_t lo, hi;
    radix_slice(A, s, lo, hi);
    for (uint64_t tile = lo; tile < hi; tile += kRadixTile) {
        const uint32_t n = (uint32_t)(hi - tile < (uint64_t)kRadixTile ? hi - tile : (uint64_t)kRadixTile);
        if (tid < 256) cnt[tid] = 0;
        __syncthreads();
        uint64_t w[kRadixPer];
        uint32_t rk[kRadixPer];  // bin << 16 | rank of the word inside (tile, bin)
        // (all loads of the tile first, no branch in between: sixteen of them in flight per thread)
#pragma unroll
        for (int j = 0; j < kRadixPer; j++) {
            const uint32_t p = (uint32_t)j * kRadixBlock + tid;
            w[j] = p < n ? A.src[tile + p] : kEmptyKey;
        }
#pragma unroll
        for (int j = 0; j < kRadixPer; j++) {
            const uint32_t p = (uint32_t)j * kRadixBlock + tid;
            rk[j] = 0xFFFFFFFFu;
            // (kEmptyKey among the words = "no word", the padding of the hash regions: dropped on the way into bins of
            //  fixed capacity; the exact pass moves it like any word — its histogram counted it — and the sets skip it)
            if (p < n && (RECORDS || !A.bin_cap || w[j] != kEmptyKey)) {
                const uint32_t b = radix_bin(w[j], A.shift);
                rk[j] = (b << 16) | atomicAdd(&cnt[b], 1u);
            }
        }
        __syncthreads();
        // exclusive scan of the 256 counts: shuffles inside each of the first four waves, their totals through LDS
        // (two barriers, not sixteen), and one global reservation per non-empty bin
        uint32_t mine = tid < 256 ? cnt[tid] : 0u;
        uint32_t incl = mine;
        if (tid < 256) {
            for (int off = 1; off < 64; off <<= 1) {
                uint32_t t = __shfl_up(incl, off, 64);
                if ((int)(tid & 63) >= off) incl += t;
            }
            if ((tid & 63) == 63) wsum[tid >> 6] = incl;
        }
        __syncthreads();
        if (tid < 256) {
            uint32_t before = 0;
            for (uint32_t w = 0; w < (tid >> 6); w++) before += wsum[w];
            pre[tid] = before + incl - mine;
            gbase[tid] = mine ? atomicAdd(&A.cursor[((size_t)s * 256 + tid) * cstride], (unsigned long long)mine) : 0ull;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kRadixPer; j++)
            if (rk[j] != 0xFFFFFFFFu) stage[pre[rk[j] >> 16] + (rk[j] & 0xFFFFu)] = w[j];
        __syncthreads();
        // the staged tile is ordered by bin: consecutive threads write consecutive words of a bin's run
        const uint32_t staged = RECORDS ? n : pre[255] + cnt[255];  // (padding words of the hash regions were dropped)
        for (uint32_t p = tid; p < staged; p += kRadixBlock) {
            const uint64_t x = stage[p];
            const uint32_t b = radix_bin(x, A.shift);
            const unsigned long long pos = gbase[b] + (p - pre[b]);
            if (!RECORDS && A.bin_cap) {  // bins of fixed capacity: no histogram pass came before
                if (pos < A.bin_cap) A.dst[((size_t)s * 256 + b) * A.bin_cap + pos] = x;
                else *(volatile uint32_t*)A.overflow = 1u;
            } else
                A.dst[pos] = x;
            if (RECORDS) sbin[p] = (uint8_t)b;
        }
        __syncthreads();
        // records: the other arrays follow the same permutation, one at a time through the same staging buffer
        for (uint32_t e = 0; RECORDS && e < A.nextra; e++) {
#pragma unroll
            for (int j = 0; j < kRadixPer; j++)
                if (rk[j] != 0xFFFFFFFFu) stage[pre[rk[j] >> 16] + (rk[j] & 0xFFFFu)] = A.src_pay[e][tile + (uint32_t)j * kRadixBlock + tid];
            __syncthreads();
            for (uint32_t p = tid; p < n; p += kRadixBlock) {
                const uint32_t b = sbin[p];
                A.dst_pay[e][gbase[b] + (p - pre[b])] = stage[p];
            }
            __syncthreads();
            uint8_t* stage8 = (uint8_t*)stage;
#pragma unroll
            for (int j = 0; j < kRadixPer; j++)
                if (rk[j] != 0xFFFFFFFFu) stage8[pre[rk[j] >> 16] + (rk[j] & 0xFFFFu)] = A.src_tag[e][tile + (uint32_t)j * kRadixBlock + tid];
            __syncthreads();
            for (uint32_t p = tid; p < n; p += kRadixBlock) {
                const uint32_t b = sbin[p];
                A.dst_tag[e][gbase[b] + (p - pre[b])] = stage8[p];
            }
            __syncthreads();
        }
    }
}

// one workgroup per bin (persistent over bins): LDS open-addressed set of the bin's words; every first insertion is
// one more member of its group's set (Set.Len(), value/set.go:198-215), counted in LDS per packed group key — by the
// key itself when the keys are small (direct_keys), else in a second LDS table keyed by the packed key — and handed
// to the groups once per workgroup (a look-up of the global group table per new member costs 0.8 ms per 100 M
// members in scattered loads).  (Tried: the next bin's words requested with unconditional loads and nothing looking at them
// before the copy at the loop's head — the pattern that repaired the other kernels' prefetch: 0.52 ms instead of 0.39, and 0.33
// instead of 0.24 with the words not even inserted; the predicated form below stays.)  The first words of the NEXT bin are loaded before the current bin is processed:
// a bin of a few thousand words is one memory latency, which would otherwise be paid bin after bin.
constexpr uint32_t kDedupeOwn = 256;  // bins per workgroup of distinct_dedupe_kernel, at most

template <int BLOCK, int U, bool TOGETHER>
__global__ __launch_bounds__(BLOCK) void distinct_dedupe_kernel(const Program P, const GlobalTable G, const DedupeArgs D) {
    extern __shared__ uint64_t dl[];
    uint64_t* set = dl;
    const uint32_t tid = threadIdx.x, mask = D.set_slots - 1, gcap = D.lds_counters, dk = D.direct_keys;
    uint64_t* gkeys = dl + D.set_slots;                                                 // hashed: gcap packed keys ...
    lds_u32* gcnt = (lds_u32*)(dl + D.set_slots + (dk ? 0u : gcap));                    // ... and the member counts
    if (dk) {
        for (uint32_t i = tid; i < dk; i += BLOCK) gcnt[i] = 0;
    } else {
        for (uint32_t i = tid; i < gcap; i += BLOCK) {
            *(volatile lds_u64*)lds_word(gkeys, i) = kEmptyKey;
            gcnt[i] = 0;
        }
    }
    // the bounds of the workgroup's bins (blockIdx.x + i * gridDim.x; the launch leaves it kDedupeOwn at most) go to LDS first:
    // read from global memory when the walk reached a bin, they were one exposed memory latency per bin — in front of the
    // request for the bin's words, which needs them
    __shared__ unsigned long long own_lo[kDedupeOwn], own_hi[kDedupeOwn];
    for (uint32_t i = tid; i < kDedupeOwn; i += BLOCK) {
        const uint64_t b = (uint64_t)blockIdx.x + (uint64_t)i * gridDim.x;
        unsigned long long l = 0, h = 0;
        if (b < D.nbins) {
            if (D.bin_count) {
                const uint64_t c = D.bin_count[(size_t)b * (D.count_stride ? D.count_stride : 1u)];
                l = b * D.bin_stride;
                h = l + (c < D.bin_stride ? c : D.bin_stride);
            } else {
                l = D.bin_start[b];
                h = D.bin_start[b + 1];
            }
        }
        own_lo[i] = l;
        own_hi[i] = h;
    }
    __syncthreads();
    uint32_t ord = 0;  // ordinal of the current bin among the workgroup's
    auto bounds = [&](uint32_t o, uint64_t& lo, uint64_t& hi) {
        lo = hi = 0;
        if (o >= kDedupeOwn) return;
        lo = own_lo[o];
        hi = own_hi[o];
    };
    uint32_t overflow = 0;
    uint32_t bin = blockIdx.x;
    uint64_t lo, hi;
    bounds(ord, lo, hi);
    uint64_t cur[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const uint64_t i = lo + (uint64_t)u * BLOCK + tid;
        cur[u] = i < hi ? D.words[i] : kEmptyKey;
    }
    while (bin < D.nbins) {
        uint64_t nlo, nhi, nxt[U];
        bounds(ord + 1, nlo, nhi);
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t i = nlo + (uint64_t)u * BLOCK + tid;
            nxt[u] = i < nhi ? D.words[i] : kEmptyKey;
        }
        if (lo != hi) {
            // (LDS-only barriers: __syncthreads() would also drain the loads of the next bin's words just issued)
            if (!(D.pad1 & 2u)) {
            lds_barrier();
            for (uint32_t i = tid; i < D.set_slots; i += BLOCK) *(volatile lds_u64*)lds_word(set, i) = kEmptyKey;
            lds_barrier();
            }
            for (uint64_t base = lo; base < hi; base += (uint64_t)BLOCK * U) {
                if (base != lo) {
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        const uint64_t i = base + (uint64_t)u * BLOCK + tid;
                        cur[u] = i < hi ? D.words[i] : kEmptyKey;
                    }
                }
                // all U words probe together, one compare-and-swap per word and round: the LDS latency of a round is paid
                // once for the U of them, and a slot that already holds the word answers without a separate read
                uint32_t hh[U];
                int state[U];  // 0 probing, 1 fresh, 2 already a member / no word
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const uint64_t w = cur[u];
                    // slot in the set: any spreading function will do (the partition hashed with mix64; the words of a
                    // bin differ in their value and key bits)
                    uint32_t h = ((uint32_t)w ^ ((uint32_t)(w >> 32) * 0x9E3779B1u)) * 0x85EBCA6Bu;
                    hh[u] = (h ^ (h >> 15)) & mask;
                    state[u] = (w == kEmptyKey || (D.pad1 & 1u)) ? 2 : 0;  // (pad1: timing experiments only)
                }
                if (!TOGETHER) {
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        for (int probe = 0; probe < 64 && !state[u]; probe++) {
                            // the swap IS the look-up: most words of a bin are new members (the sets are a quarter full at
                            // most), and what it returns tells a member from another word's slot — one LDS operation per
                            // step instead of a read and a swap (the kernel is bound by its LDS operations)
                            lds_u64* sp = lds_word(set, hh[u]);
                            unsigned long long expected = kEmptyKey;
                            if (__hip_atomic_compare_exchange_strong(sp, &expected, (unsigned long long)cur[u], __ATOMIC_RELAXED,
                                                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
                                state[u] = 1;
                            else if (expected == cur[u]) state[u] = 2;
                            hh[u] = (hh[u] + 1) & mask;
                        }
                    }
                }
                for (int probe = 0; TOGETHER && probe < 64; probe++) {
                    // a round: read the U slots, then swap into the ones found free (an atomic only where it can succeed)
                    unsigned long long seen[U];
                    bool busy = false;
#pragma unroll
                    for (int u = 0; u < U; u++) seen[u] = state[u] ? 0ull : lds_peek(lds_word(set, hh[u]));
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        if (state[u] || seen[u] != kEmptyKey) continue;
                        (void)__hip_atomic_compare_exchange_strong(lds_word(set, hh[u]), &seen[u], (unsigned long long)cur[u],
                                                                   __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (seen[u] == kEmptyKey) {  // the swap went through: a new member
                            state[u] = 1;
                            seen[u] = cur[u];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        if (state[u]) continue;
                        if (seen[u] == cur[u]) state[u] = 2;
                        else { hh[u] = (hh[u] + 1) & mask; busy = true; }
                    }
                    if (!busy) break;
                }
#pragma unroll
                for (int u = 0; u < U; u++) {
                    if (!state[u]) overflow = 1;  // more distinct words in the bin than the LDS set takes: the caller falls back
                    if (state[u] != 1) continue;
                    const uint64_t key = cur[u] >> D.key_shift;
                    if (dk) {
                        if (key < dk) (void)__hip_atomic_fetch_add(gcnt + (uint32_t)key, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        else overflow = 1;  // cannot happen: the key domain was sized from the dictionary
                    } else if (gcap) {
                        uint32_t gh = lds_hash(key, gcap);
                        bool placed = false;
                        for (uint32_t probe = 0; probe < gcap && !placed; probe++) {
                            lds_u64* kp = lds_word(gkeys, gh);
                            unsigned long long c = lds_peek(kp);
                            if (c == kEmptyKey) {
                                unsigned long long expected = kEmptyKey;
                                if (__hip_atomic_compare_exchange_strong(kp, &expected, (unsigned long long)key, __ATOMIC_RELAXED,
                                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
                                    c = key;
                                else
                                    c = expected;
                            }
                            if (c == key) {
                                (void)__hip_atomic_fetch_add(gcnt + gh, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                placed = true;
                            }
                            gh = gh + 1 == gcap ? 0 : gh + 1;
                        }
                        if (!placed) overflow = 1;  // cannot happen: gcap is the capacity of the group table itself
                    } else {
                        long long g = global_find(G, key);
                        if (g < 0) overflow = 1;
                        else atomicAdd(&D.counts[g], 1ull);
                    }
                }
            }
        }
        bin += gridDim.x;
        ord++;
        lo = nlo;
        hi = nhi;
#pragma unroll
        for (int u = 0; u < U; u++) cur[u] = nxt[u];
    }
    __syncthreads();
    const uint32_t ncount = dk ? dk : gcap;
    for (uint32_t i = tid; i < ncount; i += BLOCK) {
        const uint32_t c = gcnt[i];
        if (!c) continue;
        long long g = global_find(G, dk ? (uint64_t)i : (uint64_t)lds_peek(lds_word(gkeys, i)));
        if (g < 0) overflow = 1;
        else atomicAdd(&D.counts[g], (unsigned long long)c);
    }
    if (overflow) atomicOr(D.overflow, 1u);
}

// fallback when a bin overflowed its LDS set: one open-addressed set of words in global memory for the whole log
__global__ void distinct_words_global_kernel(const GlobalTable G, const uint64_t* words, uint64_t n, uint64_t* table,
                                             uint64_t mask, uint32_t key_shift, unsigned long long* counts,
                                             uint32_t* err_flags) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t w = words[i];
        if (w == kEmptyKey) continue;  // "no word"
        uint64_t h = mix64(w) & mask;
        int state = 0;
        for (uint64_t probe = 0; probe <= mask && !state; probe++) {
            unsigned long long cur = __hip_atomic_load((unsigned long long*)&table[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == w) state = 2;
            else if (cur == kEmptyKey) {
                unsigned long long old = atomicCAS((unsigned long long*)&table[h], (unsigned long long)kEmptyKey, (unsigned long long)w);
                if (old == kEmptyKey) state = 1;
                else if (old == w) state = 2;
            }
            h = (h + 1) & mask;
        }
        if (state == 1) {
            long long g = global_find(G, w >> key_shift);
            if (g >= 0) atomicAdd(&counts[g], 1ull);
            else atomicOr(err_flags, (uint32_t)ERR_TABLE_FULL);
        } else if (!state)
            atomicOr(err_flags, (uint32_t)ERR_TABLE_FULL);
    }
}

// `veto` (or null): two flags of the optimistic path (an LDS set / a bin of fixed capacity overflowed) — when either is
// set the counts are worthless, nothing is added and the host, which reads the flags with the results, takes the exact path
__global__ void distinct_add_counts_kernel(const Program P, const GlobalTable G, const unsigned long long* counts, uint32_t glob_off,
                                           const uint32_t* veto) {
    uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= G.capacity) return;
    if (veto && (veto[0] | veto[1])) return;
    if (counts[s]) G.acc[s * P.glob_words + glob_off] += counts[s];
}

// ------------------------------------------------------------------ K1+K2+K3(+K4): scan -> filter -> group
//
// One persistent workgroup per CU slice walks tiles of BLOCK*R rows.  DIRECT = the group-key domain is small
// and dictionary coded, so the LDS slot is a perfect hash of the key fields (no probing, no key compare);
// otherwise the LDS table is open addressed on the packed key.  Either way the workgroup's partial groups are
// merged into the global open-addressed table at the end (K4).

template <int R, int BLOCK, bool DIRECT>
__global__ __launch_bounds__(BLOCK) void scan_group_kernel(const Program P, const ScanArgs A, const GlobalTable G,
                                                          unsigned long long* ngroups) {
    extern __shared__ uint64_t lds[];
    __shared__ uint32_t lds_fill;
    __shared__ uint32_t log_wave_cnt[BLOCK / 64];
    __shared__ unsigned long long log_tile_base;
    const uint32_t S = A.lds_slots;
    const uint32_t tid = threadIdx.x;

    lds_table_init<BLOCK>(P, lds, S, tid);
    if (tid == 0) lds_fill = 0;
    uint64_t* dcache = lds + (size_t)S * P.lds_words;  // "already logged" caches of the COUNT(DISTINCT) aggregates
    for (uint32_t i = tid; i < A.dcache_slots * A.dcache_aggs; i += BLOCK) *(volatile lds_u64*)lds_word(dcache, i) = kEmptyKey;
    __shared__ uint32_t s_whist[kMaxDistinct * 256];  // first radix digit of the words this workgroup logs
    if (A.word_hist)
        for (uint32_t i = tid; i < kMaxDistinct * 256; i += BLOCK) s_whist[i] = 0;
    __syncthreads();

    uint32_t unsupported = 0, unpackable = 0;
    unsigned long long selected = 0;
    const uint64_t tile_rows = (uint64_t)BLOCK * R;
    const uint64_t nrows = A.nrows_dev && *A.nrows_dev < A.nrows ? *A.nrows_dev : A.nrows;
    const uint64_t ntiles = (nrows + tile_rows - 1) / tile_rows;

    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        uint64_t row[R];
        bool valid[R], pass[R];
#pragma unroll
        for (int j = 0; j < R; j++) {
            row[j] = tile * tile_rows + (uint64_t)j * BLOCK + tid;
            valid[j] = row[j] < nrows;
        }
        eval_predicate<R>(P, row, valid, pass, unsupported);

        // group key -> packed word (execution/group_util.go:18-35) and, in DIRECT mode, the LDS slot itself
        uint64_t key[R];
        uint32_t dslot[R];
#pragma unroll
        for (int j = 0; j < R; j++) { key[j] = 0; dslot[j] = 0; }
        for (uint32_t k = 0; k < P.nkeys; k++) {
            const KeySpec& ks = P.keys[k];
            uint32_t kt[R];
            uint64_t kp[R];
            load_operand<R>(P, ks.src, row, pass, kt, kp);
#pragma unroll
            for (int j = 0; j < R; j++) {
                uint64_t f = 0, canon;
                bool ok = !pass[j] || pack_key_field(P, ks, kt[j], kp[j], f, canon);
                if (DIRECT) ok = ok && f < (uint64_t)A.direct_radix[k];
                if (!ok) {
                    unpackable = 1;
                    pass[j] = false;
                    f = 0;
                }
                key[j] |= f << ks.shift;
                if (DIRECT) dslot[j] += (uint32_t)f * A.direct_stride[k];
            }
        }

        // InitialGroup: find / seed the group (execution/group_initial.go:69-79)
        int slot[R];
        long long grow[R];
#pragma unroll
        for (int j = 0; j < R; j++) {
            slot[j] = -1;
            grow[j] = -1;
            if (pass[j]) {
                selected++;
                if (DIRECT) {
                    slot[j] = (int)dslot[j];
                    if (lds_peek(lds_word(lds, dslot[j])) == kEmptyKey) *(volatile lds_u64*)lds_word(lds, dslot[j]) = key[j];
                } else {
                    slot[j] = lds_find_or_insert(lds, S, key[j], &lds_fill, A.lds_max_fill);
                    if (slot[j] < 0) {
                        grow[j] = global_find_or_insert(G, key[j], A.err_flags, ngroups);
                        if (grow[j] < 0) pass[j] = false;
                    }
                }
                if (P.want_rep_row && pass[j]) {
                    unsigned long long ord = A.row_base + row[j];
                    if (slot[j] >= 0) lds_min_u64(lds_word(lds, P.rep_lds_word * S + (uint32_t)slot[j]), ord);
                    else atomicMin((unsigned long long*)&G.rep_row[grow[j]], ord);
                }
            }
        }

        // CumulateInitial of every aggregate (execution/group_initial.go:89-97)
        for (uint32_t a = 0; a < P.naggs; a++) {
            const AggSpec& ag = P.aggs[a];
            uint32_t vt[R];
            uint64_t vp[R];
            if (ag.has_operand) load_operand<R>(P, ag.src, row, pass, vt, vp);
            if (ag.distinct) {
                // setAdd (algebra/agg_util.go:30-47): here the (group, value) pair is logged and de-duplicated at
                // finish; the per-class operand counts size each group's value set.  Log space is reserved once
                // per workgroup tile (one atomic on the cursor per BLOCK*R rows, not per wave).
                uint32_t cls[R];
                uint64_t val[R];
                bool q[R], nar[R];
                uint32_t mine = 0, mine_w = 0;
                const bool words = ag.kind == AGG_COUNT && A.log_word[ag.log_index] != nullptr;
#pragma unroll
                for (int j = 0; j < R; j++) {
                    cls[j] = 0;
                    val[j] = 0;
                    nar[j] = false;
                    if (ag.kind == AGG_ARRAY) {
                        // ArrayAgg.CumulateInitial (algebra/agg_array.go:86-97): every operand but MISSING joins the
                        // group's array; here it is logged (key, payload, TAG) and the arrays are put together at finish
                        q[j] = pass[j] && vt[j] != T_MISSING;
                        cls[j] = vt[j];
                        val[j] = vp[j];
                        if (q[j]) mine++;
                        continue;
                    }
                    q[j] = pass[j] && distinct_classify(ag.kind, vt[j], vp[j], cls[j], val[j]);
                    if (!q[j]) continue;
                    uint64_t word;
                    if (words && member_word(A, key[j], cls[j], val[j], word)) {
                        // one-word member: skip it when this workgroup logged the very same word already (a
                        // direct-mapped LDS cache; a racing duplicate only costs a redundant log entry)
                        nar[j] = true;
                        val[j] = word;
                        if (A.dcache_slots) {
                            lds_u64* c = lds_word(dcache, ag.log_index * A.dcache_slots + ((uint32_t)mix64(word) & (A.dcache_slots - 1)));
                            if (lds_peek(c) == word) q[j] = false;
                            else *(volatile lds_u64*)c = word;
                        }
                        if (q[j]) mine_w++;
                    } else {
                        mine++;
                        if (slot[j] >= 0) lds_add_u64(lds_word(lds, (ag.lds_off + cls[j]) * S + (uint32_t)slot[j]), 1ull);
                        else atomicAdd((unsigned long long*)&G.acc[(size_t)grow[j] * P.glob_words + ag.glob_off + 1 + cls[j]], 1ull);
                    }
                }
                // log space is reserved once per workgroup tile and log (one atomic on the cursor per BLOCK*R rows)
                unsigned long long pos = tile_reserve<BLOCK>(mine, &A.log_cursor[ag.log_index], log_wave_cnt, &log_tile_base, tid);
#pragma unroll
                for (int j = 0; j < R; j++) {
                    if (!q[j] || nar[j]) continue;
                    if (pos < A.log_capacity) {
                        A.log_key[ag.log_index][pos] = key[j];
                        A.log_val[ag.log_index][pos] = val[j];
                        A.log_cls[ag.log_index][pos] = (uint8_t)cls[j];
                    } else {
                        atomicOr(A.err_flags, (uint32_t)ERR_TABLE_FULL);
                    }
                    pos++;
                }
                __syncthreads();
                if (words) {
                    pos = tile_reserve<BLOCK>(mine_w, &A.word_cursor[ag.log_index], log_wave_cnt, &log_tile_base, tid);
#pragma unroll
                    for (int j = 0; j < R; j++) {
                        if (!q[j] || !nar[j]) continue;
                        if (pos < A.log_capacity) {
                            A.log_word[ag.log_index][pos] = val[j];
                            if (A.word_hist) atomicAdd(&s_whist[ag.log_index * 256 + radix_bin(val[j], 56)], 1u);
                        } else
                            atomicOr(A.err_flags, (uint32_t)ERR_TABLE_FULL);
                        pos++;
                    }
                    __syncthreads();
                }
                continue;
            }
#pragma unroll
            for (int j = 0; j < R; j++) {
                if (!pass[j]) continue;
                uint32_t t = ag.has_operand ? vt[j] : (uint32_t)T_NULL;
                uint64_t p = ag.has_operand ? vp[j] : 0ull;
                if (slot[j] >= 0) {
                    if (!acc_lds(P, ag, lds, S, (uint32_t)slot[j], t, p)) {
                        if (grow[j] < 0) grow[j] = global_find_or_insert(G, key[j], A.err_flags, ngroups);
                        if (grow[j] >= 0) acc_global(P, ag, &G.acc[(size_t)grow[j] * P.glob_words], t, p);
                    }
                } else {
                    acc_global(P, ag, &G.acc[(size_t)grow[j] * P.glob_words], t, p);
                }
            }
        }
    }

    if (unsupported) atomicOr(A.err_flags, (uint32_t)ERR_UNSUPPORTED_VALUE);
    if (unpackable) atomicOr(A.err_flags, (uint32_t)ERR_UNPACKABLE_KEY);
    if (A.word_hist) {
        __syncthreads();
        for (uint32_t i = tid; i < kMaxDistinct * 256; i += BLOCK)
            if (s_whist[i]) atomicAdd(&A.word_hist[i], (unsigned long long)s_whist[i]);
    }
    // rows that passed the Filter (≙ Filter #itemsOut): one global atomic per workgroup
    __shared__ unsigned long long block_selected;
    if (tid == 0) block_selected = 0;
    __syncthreads();
    for (int off = 32; off > 0; off >>= 1) selected += __shfl_down(selected, off, 64);
    if ((tid & 63) == 0 && selected) atomicAdd(&block_selected, selected);
    __syncthreads();
    if (tid == 0 && block_selected) atomicAdd(A.rows_selected, block_selected);
    // K4: merge this workgroup's partial groups into the global table (≙ IntermediateGroup)
    for (uint32_t s = tid; s < S; s += BLOCK) {
        uint64_t key = lds[s];
        if (key == kEmptyKey) continue;
        long long g = global_find_or_insert(G, key, A.err_flags, ngroups);
        if (g < 0) continue;
        merge_slot(P, lds, S, s, &G.acc[(size_t)g * P.glob_words]);
        if (P.want_rep_row) {
            uint64_t rep = lds[(size_t)P.rep_lds_word * S + s];
            if (rep != ~0ull) atomicMin((unsigned long long*)&G.rep_row[g], (unsigned long long)rep);
        }
    }
}

// ------------------------------------------------------------------ fast scan kernel (bounded plan shapes)
//
// Same algorithm and the same LDS/global tables as scan_group_kernel<.., DIRECT = true>, but every plan
// descriptor sits at a compile-time index (loops are fully unrolled to the kFast* maxima and guarded by
// wave-uniform counts), so the compiler hoists all kernarg reads out of the tile loop and nothing is
// interpreted per row.  Each column is loaded exactly once per row.

// one cheap predicate term on a register-resident column value -> "is TRUE" (the only thing Filter needs)
N1K_DEV bool fast_term_true(const FastTerm& t, uint32_t tg, uint64_t p) {
    switch (t.op) {
        case TERM_IS_NULL: return tg == T_NULL;
        case TERM_IS_NOT_NULL: return tg > T_NULL;
        case TERM_IS_MISSING: return tg == T_MISSING;
        case TERM_IS_NOT_MISSING: return tg != T_MISSING;
        case TERM_IS_VALUED: return tg > T_NULL;
        case TERM_IS_NOT_VALUED: return tg <= T_NULL;
        case TERM_STR_EQ: return tg == T_STRING && p == t.cpayload;
        default: {
            if (tg <= T_NULL) return false;  // MISSING / NULL are never TRUE
            int c;
            if (tg == T_INT && t.ctag == T_INT) {
                int64_t x = (int64_t)p, y = (int64_t)t.cpayload;
                c = x < y ? -1 : (x > y ? 1 : 0);
            } else if (tg == T_INT || tg == T_FLOAT) {
                c = collate_f64(num_actual(tg, p), num_actual(t.ctag, t.cpayload));
            } else {
                c = tg < T_INT ? -1 : 1;
            }
            return t.op == TERM_NUM_LT ? c < 0 : t.op == TERM_NUM_LE ? c <= 0 : t.op == TERM_NUM_GT ? c > 0
                   : t.op == TERM_NUM_GE ? c >= 0 : (c == 0 && (tg == T_INT || tg == T_FLOAT));
        }
    }
}

// pick one of the register-resident columns by a wave-uniform index
template <int R>
N1K_DEV void pick_col(uint32_t c, const uint32_t (&ctag)[kFastCols][R], const uint64_t (&cpay)[kFastCols][R],
                      uint32_t (&t)[R], uint64_t (&p)[R]) {
#pragma unroll
    for (int j = 0; j < R; j++) { t[j] = ctag[0][j]; p[j] = cpay[0][j]; }
#pragma unroll
    for (int k = 1; k < kFastCols; k++) {
        if (c == (uint32_t)k) {
#pragma unroll
            for (int j = 0; j < R; j++) { t[j] = ctag[k][j]; p[j] = cpay[k][j]; }
        }
    }
}

template <int R, int BLOCK>
__global__ __launch_bounds__(BLOCK) void scan_fast_kernel(const Program P, const FastArgs F, const GlobalTable G,
                                                         unsigned long long* ngroups) {
    extern __shared__ uint64_t lds[];
    const uint32_t S = F.lds_slots;
    const uint32_t tid = threadIdx.x;
    lds_table_init<BLOCK>(P, lds, S, tid);
    __syncthreads();

    uint32_t unpackable = 0;
    uint32_t selected = 0;
    const uint32_t tile_rows = BLOCK * R;
    const uint32_t n = F.nrows;

    for (uint32_t base = blockIdx.x * tile_rows; base < n; base += gridDim.x * tile_rows) {
        uint32_t idx[R];
        bool pass[R];
#pragma unroll
        for (int j = 0; j < R; j++) {
            idx[j] = base + (uint32_t)j * BLOCK + tid;
            pass[j] = idx[j] < n;
        }
        // every referenced column, once
        uint32_t ctag[kFastCols][R];
        uint64_t cpay[kFastCols][R];
#pragma unroll
        for (int c = 0; c < kFastCols; c++) {
            if (c < (int)F.ncols) {
                if (F.cols[c].kind == COLK_DICT32) {
                    const uint32_t* __restrict__ codes = F.cols[c].codes;
#pragma unroll
                    for (int j = 0; j < R; j++) {
                        uint32_t code = pass[j] ? codes[idx[j]] : 0xFFFFFFFFu;
                        ctag[c][j] = code == 0xFFFFFFFFu ? (uint32_t)T_MISSING : (code == 0xFFFFFFFEu ? (uint32_t)T_NULL : (uint32_t)T_STRING);
                        cpay[c][j] = code;
                    }
                } else {
                    const uint8_t* __restrict__ tags = F.cols[c].tags;
                    const uint64_t* __restrict__ payload = F.cols[c].payload;
#pragma unroll
                    for (int j = 0; j < R; j++) {
                        ctag[c][j] = pass[j] ? (uint32_t)tags[idx[j]] : (uint32_t)T_MISSING;
                        cpay[c][j] = pass[j] ? payload[idx[j]] : 0ull;
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < R; j++) { ctag[c][j] = T_MISSING; cpay[c][j] = 0; }
            }
        }
        // Filter: a conjunction passes iff every term is TRUE (expression/logic_and.go:64-89)
#pragma unroll
        for (int t = 0; t < kFastTerms; t++) {
            if (t < (int)F.nterms) {
                uint32_t vt[R];
                uint64_t vp[R];
                pick_col<R>(F.terms[t].col, ctag, cpay, vt, vp);
#pragma unroll
                for (int j = 0; j < R; j++) pass[j] = pass[j] && fast_term_true(F.terms[t], vt[j], vp[j]);
            }
        }
        // group key: perfect-hash slot (execution/group_util.go:18-35); the packed key is rebuilt from the slot
        // index when the workgroup's table is merged
        uint32_t slot[R];
#pragma unroll
        for (int j = 0; j < R; j++) slot[j] = 0;
#pragma unroll
        for (int k = 0; k < kFastKeys; k++) {
            if (k < (int)F.nkeys) {
                uint32_t vt[R];
                uint64_t vp[R];
                pick_col<R>(F.keys[k].col, ctag, cpay, vt, vp);
#pragma unroll
                for (int j = 0; j < R; j++) {
                    uint32_t tg = vt[j];
                    uint32_t f = tg == T_MISSING ? 0u : (tg == T_NULL ? 1u : (uint32_t)vp[j] + 2u);
                    if (pass[j] && ((tg > T_NULL && tg != T_STRING) || f >= F.keys[k].radix)) {
                        unpackable = 1;
                        pass[j] = false;
                    }
                    slot[j] += (pass[j] ? f : 0u) * F.keys[k].stride;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < R; j++) {
            if (pass[j]) {
                selected++;
                if (lds_peek(lds_word(lds, slot[j])) == kEmptyKey) *(volatile lds_u64*)lds_word(lds, slot[j]) = 1ull;
            }
        }
        // CumulateInitial of every aggregate (execution/group_initial.go:89-97)
#pragma unroll
        for (int a = 0; a < kFastAggs; a++) {
            if (a < (int)F.naggs) {
                const AggSpec& ag = P.aggs[a];
                if (!ag.has_operand) {
#pragma unroll
                    for (int j = 0; j < R; j++)
                        if (pass[j]) acc_lds(P, ag, lds, S, slot[j], T_NULL, 0ull);
                } else {
                    uint32_t vt[R];
                    uint64_t vp[R];
                    pick_col<R>(F.agg_col[a], ctag, cpay, vt, vp);
#pragma unroll
                    for (int j = 0; j < R; j++) {
                        if (pass[j] && !acc_lds(P, ag, lds, S, slot[j], vt[j], vp[j])) {
                            long long g = global_find_or_insert(G, fast_slot_key(F, slot[j]), F.err_flags, ngroups);
                            if (g >= 0) acc_global(P, ag, &G.acc[(size_t)g * P.glob_words], vt[j], vp[j]);
                        }
                    }
                }
            }
        }
    }

    if (unpackable) atomicOr(F.err_flags, (uint32_t)ERR_UNPACKABLE_KEY);
    __shared__ unsigned int block_selected;
    if (tid == 0) block_selected = 0;
    __syncthreads();
    for (int off = 32; off > 0; off >>= 1) selected += __shfl_down(selected, off, 64);
    if ((tid & 63) == 0 && selected) atomicAdd(&block_selected, selected);
    __syncthreads();
    if (F.slabs) {
        uint64_t* slab = F.slabs + (size_t)blockIdx.x * P.lds_words * S;
        for (uint32_t i = tid; i < P.lds_words * S; i += BLOCK) slab[i] = lds[i];
        if (tid == 0) F.block_selected[blockIdx.x] = block_selected;
        return;
    }
    if (tid == 0 && block_selected) atomicAdd(F.rows_selected, (unsigned long long)block_selected);
    // K4: merge this workgroup's partial groups into the global table (≙ IntermediateGroup)
    for (uint32_t s = tid; s < S; s += BLOCK) {
        if (lds[s] == kEmptyKey) continue;
        long long g = global_find_or_insert(G, fast_slot_key(F, s), F.err_flags, ngroups);
        if (g < 0) continue;
        merge_slot(P, lds, S, s, &G.acc[(size_t)g * P.glob_words]);
    }
}

}  // namespace n1k
#include "n1k_spec.h"
namespace n1k {

// ---- ahead-of-time instantiated plan shapes ------------------------------------------------------------------
// (column order = order of first use in the plan: condition, keys, aggregates; aggregates sorted by text as the
//  planner emits them, planner/build_select_sub.go:551-558)
#define T64 COLK_TAGGED64
#define D32 COLK_DICT32
#define ST(op, col, ci) SpecTerm{op, col, ci}
#define SA(kind, has, col) SpecAgg{kind, has, col, 0}
#define SAD(col) SpecAgg{AGG_COUNT, 1, col, 1}  // COUNT(DISTINCT col)
#define NOTERM ST(0, 0, 0)
#define NOAGG SA(0, 0, 0)
// SELECT k, SUM(x) WHERE x <cmp> int GROUP BY k          (BASELINE config 2)
N1K_DEFINE_SPEC(Spec_gt_sum, 2, T64, D32, 0, 1, ST(TERM_NUM_GT, 0, 1), NOTERM, 1, 1, 0, 1, SA(AGG_SUM, 1, 0), NOAGG, NOAGG, NOAGG, NOAGG);
N1K_DEFINE_SPEC(Spec_lt_sum, 2, T64, D32, 0, 1, ST(TERM_NUM_LT, 0, 1), NOTERM, 1, 1, 0, 1, SA(AGG_SUM, 1, 0), NOAGG, NOAGG, NOAGG, NOAGG);
N1K_DEFINE_SPEC(Spec_gtf_sum, 2, T64, D32, 0, 1, ST(TERM_NUM_GT, 0, 0), NOTERM, 1, 1, 0, 1, SA(AGG_SUM, 1, 0), NOAGG, NOAGG, NOAGG, NOAGG);
// ... with AVG, COUNT(*), MAX, MIN, SUM of the filtered column
N1K_DEFINE_SPEC(Spec_gt_all, 2, T64, D32, 0, 1, ST(TERM_NUM_GT, 0, 1), NOTERM, 1, 1, 0, 5, SA(AGG_AVG, 1, 0), SA(AGG_COUNT, 0, 0),
                SA(AGG_MAX, 1, 0), SA(AGG_MIN, 1, 0), SA(AGG_SUM, 1, 0));
// SELECT k, COUNT(*) WHERE x > int GROUP BY k
N1K_DEFINE_SPEC(Spec_gt_count, 2, T64, D32, 0, 1, ST(TERM_NUM_GT, 0, 1), NOTERM, 1, 1, 0, 1, SA(AGG_COUNT, 0, 0), NOAGG, NOAGG, NOAGG, NOAGG);
// SELECT k, SUM(x) GROUP BY k ; SELECT k, AVG(x) GROUP BY k ; SELECT k, COUNT(*) GROUP BY k
N1K_DEFINE_SPEC(Spec_sum, 2, D32, T64, 0, 0, NOTERM, NOTERM, 1, 0, 0, 1, SA(AGG_SUM, 1, 1), NOAGG, NOAGG, NOAGG, NOAGG);
N1K_DEFINE_SPEC(Spec_avg, 2, D32, T64, 0, 0, NOTERM, NOTERM, 1, 0, 0, 1, SA(AGG_AVG, 1, 1), NOAGG, NOAGG, NOAGG, NOAGG);
N1K_DEFINE_SPEC(Spec_count, 1, D32, 0, 0, 0, NOTERM, NOTERM, 1, 0, 0, 1, SA(AGG_COUNT, 0, 0), NOAGG, NOAGG, NOAGG, NOAGG);
// SELECT COUNT(*) / SUM(x) WHERE x > int
N1K_DEFINE_SPEC(Spec_gt_nokey_count, 1, T64, 0, 0, 1, ST(TERM_NUM_GT, 0, 1), NOTERM, 0, 0, 0, 1, SA(AGG_COUNT, 0, 0), NOAGG, NOAGG, NOAGG, NOAGG);
// integer (TAGGED64) keys: open-addressed LDS table
N1K_DEFINE_SPEC(Spec_ik_count, 1, T64, 0, 0, 0, NOTERM, NOTERM, 1, 0, 0, 1, SA(AGG_COUNT, 0, 0), NOAGG, NOAGG, NOAGG, NOAGG);
N1K_DEFINE_SPEC(Spec_ik_sum, 2, T64, T64, 0, 0, NOTERM, NOTERM, 1, 0, 0, 1, SA(AGG_SUM, 1, 1), NOAGG, NOAGG, NOAGG, NOAGG);
N1K_DEFINE_SPEC(Spec_dik_sum, 3, D32, T64, T64, 0, NOTERM, NOTERM, 2, 0, 1, 1, SA(AGG_SUM, 1, 2), NOAGG, NOAGG, NOAGG, NOAGG);
// SELECT k, COUNT(DISTINCT u), AVG(x) GROUP BY k        (BASELINE config 3; aggregates sorted by text: avg, count)
N1K_DEFINE_SPEC(Spec_cd_avg, 3, D32, T64, T64, 0, NOTERM, NOTERM, 1, 0, 0, 2, SA(AGG_AVG, 1, 1), SAD(2), NOAGG, NOAGG, NOAGG);
// SELECT k, COUNT(DISTINCT u) GROUP BY k ; ... , COUNT(u)
N1K_DEFINE_SPEC(Spec_cd, 2, D32, T64, 0, 0, NOTERM, NOTERM, 1, 0, 0, 1, SAD(1), NOAGG, NOAGG, NOAGG, NOAGG);
// SELECT k1, k2, SUM(x) GROUP BY k1, k2 (two dictionary keys)
N1K_DEFINE_SPEC(Spec_2k_sum, 3, D32, D32, T64, 0, NOTERM, NOTERM, 2, 0, 1, 1, SA(AGG_SUM, 1, 2), NOAGG, NOAGG, NOAGG, NOAGG);
#undef T64
#undef D32
#undef ST
#undef SA
#undef SAD
#undef NOTERM
#undef NOAGG

template <class Spec>
static SpecSig make_sig() {
    SpecSig g{};
    g.ncols = Spec::ncols; g.nterms = Spec::nterms; g.nkeys = Spec::nkeys; g.naggs = Spec::naggs;
    g.hashed = spec_hashed<Spec>() ? 1 : 0;
    for (int c = 0; c < kFastCols; c++) g.col_kind[c] = c < Spec::ncols ? Spec::col_kind[c] : 0;
    for (int t = 0; t < kFastTerms; t++) if (t < Spec::nterms) g.terms[t] = Spec::terms[t];
    for (int k = 0; k < kFastKeys; k++) g.key_col[k] = k < Spec::nkeys ? Spec::key_col[k] : 0;
    for (int a = 0; a < kFastAggs; a++) if (a < Spec::naggs) g.aggs[a] = Spec::aggs[a];
    return g;
}

template <class Spec>
static hipError_t launch_spec(const Program& P, const FastArgs& F, const GlobalTable& G, unsigned long long* ngroups,
                              uint32_t grid, uint32_t block, bool wide, const WordLogArgs& L, hipStream_t st) {
    size_t shmem = (size_t)F.lds_slots * P.lds_words * 8 + (size_t)L.dcache_slots * spec_ndistinct<Spec>() * 8;
#define N1K_LAUNCH(R, B, W)                                                                                       \
    do {                                                                                                          \
        auto k = scan_spec_kernel<Spec, R, B, W>;                                                                 \
        if (shmem > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); \
        hipLaunchKernelGGL(k, dim3(grid), dim3(B), shmem, st, P, F, G, ngroups, L);                               \
    } while (0)
    if (block == 512) { if (wide) N1K_LAUNCH(2, 512, true); else N1K_LAUNCH(4, 512, false); }
    else { if (wide) N1K_LAUNCH(2, 1024, true); else N1K_LAUNCH(4, 1024, false); }
#undef N1K_LAUNCH
    return hipGetLastError();
}

// records mode of a shape (n1k_spec.h): Filter + packed key, 16-byte records into the hash regions; tiles of 2048 rows,
// two in flight
template <class Spec>
static hipError_t launch_spec_records(const Program& P, const FastArgs& F, uint32_t grid, bool wide, const WordLogArgs& L, hipStream_t st) {
    const size_t shmem = sizeof(ScatterLds<Rec16, 512, 4>);
    if (wide) {
        auto k = scan_spec_records_kernel<Spec, 2, 512, true>;
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        hipLaunchKernelGGL(k, dim3(grid), dim3(512), shmem, st, P, F, L);
    } else {
        auto k = scan_spec_records_kernel<Spec, 4, 512, false>;
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        hipLaunchKernelGGL(k, dim3(grid), dim3(512), shmem, st, P, F, L);
    }
    return hipGetLastError();
}
size_t spec_records_lds_bytes() { return sizeof(ScatterLds<Rec16, 512, 4>); }

const std::vector<SpecEntry>& spec_registry() {
    static const std::vector<SpecEntry> reg = {
#define N1K_REG(S) SpecEntry{#S, make_sig<S>(), &launch_spec<S>, &launch_spec_records<S>}
        N1K_REG(Spec_gt_sum), N1K_REG(Spec_lt_sum), N1K_REG(Spec_gtf_sum), N1K_REG(Spec_gt_all), N1K_REG(Spec_gt_count),
        N1K_REG(Spec_sum), N1K_REG(Spec_avg), N1K_REG(Spec_count), N1K_REG(Spec_gt_nokey_count), N1K_REG(Spec_2k_sum),
        N1K_REG(Spec_ik_count), N1K_REG(Spec_ik_sum), N1K_REG(Spec_dik_sum), N1K_REG(Spec_cd_avg), N1K_REG(Spec_cd),
#undef N1K_REG
    };
    return reg;
}

__global__ void init_table_kernel(const Program P, const GlobalTable G, uint64_t first, uint64_t count,
                                  unsigned long long* counters) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (counters && i < kCounters) counters[i] = 0;  // reopen(): counters and error flags go back to zero in the same launch
    if (i >= count) return;
    uint64_t s = first + i;
    G.keys[s] = kEmptyKey;
    if (G.rep_row) G.rep_row[s] = ~0ull;
    glob_row_init(P, &G.acc[(size_t)s * P.glob_words]);
}

// ------------------------------------------------------------------ multi-GPU: hash partition of the filter's survivors
//
// No reference analogue (the reference fans in through one in-memory queue, execution/exchange.go:161-251): every
// surviving row goes to the GPU that owns hash(group key) % nparts, so that each group lives on exactly one GPU
// and COUNT(DISTINCT) needs no cross-GPU merge.  Region d of every output column receives the rows for rank d;
// positions come from a workgroup-level reservation (LDS counters, one global atomic per tile and destination).
template <int R, int BLOCK>
__global__ __launch_bounds__(BLOCK) void partition_kernel(const Program P, const PartArgs A) {
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    uint32_t unsupported = 0, unpackable = 0;
    __shared__ uint32_t s_cnt[kMaxParts];
    __shared__ unsigned long long s_base[kMaxParts];
    for (uint32_t d = tid; d < kMaxParts; d += BLOCK) s_cnt[d] = 0;
    __syncthreads();
    const uint64_t tile_rows = (uint64_t)BLOCK * R;
    const uint64_t ntiles = (A.nrows + tile_rows - 1) / tile_rows;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        uint64_t row[R];
        bool valid[R], pass[R];
#pragma unroll
        for (int j = 0; j < R; j++) {
            row[j] = tile * tile_rows + (uint64_t)j * BLOCK + tid;
            valid[j] = row[j] < A.nrows;
        }
        eval_predicate<R>(P, row, valid, pass, unsupported);
        // the destination is a function of the key VALUES (wide-value codes are local to a handle)
        uint64_t key[R];
#pragma unroll
        for (int j = 0; j < R; j++) key[j] = 0;
        for (uint32_t k = 0; k < P.nkeys; k++) {
            const KeySpec& ks = P.keys[k];
            uint32_t kt[R];
            uint64_t kp[R];
            load_operand<R>(P, ks.src, row, pass, kt, kp);
#pragma unroll
            for (int j = 0; j < R; j++) {
                uint64_t f = 0, canon = 0;
                if (pass[j] && !pack_key_field(P, ks, kt[j], kp[j], f, canon)) {
                    unpackable = 1;
                    pass[j] = false;
                }
                key[j] = mix64(key[j] ^ canon) + k;
            }
        }
        // Reservation in two levels: every wave adds its per-destination survivor counts to the workgroup's LDS
        // counters (one ds_add per wave and destination, rank inside the wave from the ballot), then ONE global
        // atomic per destination and tile reserves the block's run in that destination's region.  (One global atomic
        // per wave serialises on nparts addresses: 1.5 M same-address atomics per 100 M rows, ~19 ms measured.)
        uint64_t pos[R];
        uint32_t dest[R];
#pragma unroll
        for (int j = 0; j < R; j++) {
            dest[j] = (uint32_t)(((key[j] >> 32) * (uint64_t)A.nparts) >> 32);
            pos[j] = 0;
            for (uint32_t d = 0; d < A.nparts; d++) {
                unsigned long long m = __ballot(pass[j] && dest[j] == d);
                if (m == 0ull) continue;
                int leader = __ffsll((long long)m) - 1;
                uint32_t base = 0;
                if ((int)lane == leader) base = atomicAdd(&s_cnt[d], (uint32_t)__popcll(m));
                base = __shfl(base, leader, 64);
                if (pass[j] && dest[j] == d) pos[j] = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            }
        }
        __syncthreads();
        for (uint32_t d = tid; d < A.nparts; d += BLOCK) {
            uint32_t n = s_cnt[d];
            s_base[d] = n ? atomicAdd(&A.counts[(size_t)d * (A.count_stride ? A.count_stride : 1u)], (unsigned long long)n) : 0ull;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < R; j++) {
            if (!pass[j]) continue;
            uint64_t r = s_base[dest[j]] + pos[j];
            if (r >= (A.per_dest ? (uint64_t)A.dest_cap[dest[j]] : A.capacity)) {
                if (!(atomicOr(A.err_flags, (uint32_t)ERR_TABLE_FULL) & ERR_TABLE_FULL) && A.region_bytes)
                    for (uint32_t q = 0; q < A.nparts; q++)  // every receiver reads the verdict in the header it gets
                        atomicOr(&A.counts[(size_t)q * A.count_stride + 1], 1ull);
                pass[j] = false;
            }
            pos[j] = A.region_bytes ? r : (uint64_t)dest[j] * A.capacity + r;
        }
        __syncthreads();
        for (uint32_t d = tid; d < A.nparts; d += BLOCK) s_cnt[d] = 0;
        // (the next tile's first barrier orders this reset before its counters are read)
        uint64_t roff[R];  // per_dest: where the next array of the row's region starts
#pragma unroll
        for (int j = 0; j < R; j++) roff[j] = A.hdr_bytes;
        for (uint32_t c = 0; c < A.ncopy; c++) {
            Operand o{};
            o.is_const = 0;
            o.col = c;
            uint32_t vt[R];
            uint64_t vp[R];
            load_operand<R>(P, o, row, pass, vt, vp);
#pragma unroll
            for (int j = 0; j < R; j++) {
                if (!pass[j]) continue;
                if (A.per_dest) {  // the region is laid out for its own destination's capacity
                    char* const reg = (char*)A.counts + (size_t)dest[j] * A.region_bytes;
                    const uint64_t capd = A.dest_cap[dest[j]];
                    if (P.cols[c].kind == COLK_DICT32) {
                        ((uint32_t*)(reg + roff[j]))[pos[j]] = (uint32_t)vp[j];
                        roff[j] = part_region_next(roff[j], capd, 4);
                    } else {
                        ((uint64_t*)(reg + roff[j]))[pos[j]] = vp[j];
                        roff[j] = part_region_next(roff[j], capd, 8);
                        ((uint8_t*)(reg + roff[j]))[pos[j]] = (uint8_t)vt[j];
                        roff[j] = part_region_next(roff[j], capd, 1);
                    }
                    continue;
                }
                const size_t shift = (size_t)dest[j] * A.region_bytes;  // (0 without packed regions)
                if (P.cols[c].kind == COLK_DICT32) {
                    ((uint32_t*)((char*)A.out_codes[c] + shift))[pos[j]] = (uint32_t)vp[j];
                } else {
                    ((uint8_t*)((char*)A.out_tags[c] + shift))[pos[j]] = (uint8_t)vt[j];
                    ((uint64_t*)((char*)A.out_payload[c] + shift))[pos[j]] = vp[j];
                }
            }
        }
    }
    if (unsupported) atomicOr(A.err_flags, (uint32_t)ERR_UNSUPPORTED_VALUE);
    if (unpackable) atomicOr(A.err_flags, (uint32_t)ERR_UNPACKABLE_KEY);
}

// ------------------------------------------------------------------ K4: IntermediateGroup over workgroup slabs
//
// execution/group_intermediate.go:56-104: the first partial met for a key is kept, later ones are merged with
// CumulateIntermediate.  Every workgroup of a DIRECT-mode scan left its LDS table (same slot for the same key in
// every workgroup) in a slab; one thread per slot folds the slabs in workgroup order — no atomics between
// workgroups, and float sums are reproducible for a given grid.
enum { RED_ADD_U64, RED_ADD_F64, RED_OR, RED_MIN_I64, RED_MAX_I64, RED_MIN_U64, RED_MAX_U64 };

// reduce one 64-bit value over threadIdx.y (16 rows) for every threadIdx.x column; all threads of the block call it
template <int OP>
N1K_DEV uint64_t reduce_over_y(uint64_t (*red)[64], uint64_t v) {
    const uint32_t tx = threadIdx.x, ty = threadIdx.y;
    red[ty][tx] = v;
    __syncthreads();
    for (uint32_t off = 8; off > 0; off >>= 1) {
        if (ty < off) {
            uint64_t a = red[ty][tx], b = red[ty + off][tx], r;
            if (OP == RED_ADD_U64) r = a + b;
            else if (OP == RED_ADD_F64) r = f64_bits(as_f64(a) + as_f64(b));
            else if (OP == RED_OR) r = a | b;
            else if (OP == RED_MIN_I64) r = (int64_t)a < (int64_t)b ? a : b;
            else if (OP == RED_MAX_I64) r = (int64_t)a > (int64_t)b ? a : b;
            else if (OP == RED_MIN_U64) r = a < b ? a : b;
            else r = a > b ? a : b;
            red[ty][tx] = r;
        }
        __syncthreads();
    }
    uint64_t out = red[0][tx];
    __syncthreads();
    return out;
}

// (defined with finalize_small_kernel below: FinalGroup of a small table by ONE workgroup of 1024 threads)
N1K_DEV void finalize_small_body(const Program& P, const GlobalTable& G, OutValue* out_keys, OutValue* out_aggs, OutPartial* out_parts,
                                 uint64_t* out_rep, unsigned long long* counters, unsigned long long* host_counters, uint64_t max_out,
                                 uint32_t* err_flags, uint32_t clear, uint32_t tid);

// T.enabled: the workgroup that finishes LAST goes on to run the query's tail (finalize_small_body) — the merge and FinalGroup of a
// small table are then one launch, without the ~ 6 us between two dependent kernels.
__global__ __launch_bounds__(1024) void merge_slabs_kernel(const Program P, const FastArgs F, const GlobalTable G,
                                                          uint32_t nblocks_total, unsigned long long* ngroups, const TailArgs T) {
    // block = 64 slots (x) x 16 chunks (y); thread (x, y) folds workgroup slabs y', y' + Y, ... (y' = its global chunk,
    // Y = 16 * gridDim.y chunks) with unconditional, unrolled loads; the 16 chunks of a block meet through an LDS tree
    // and thread y == 0 applies the result to the global row (a handful of atomics per slot and block row).
    __shared__ uint64_t red[16][64];
    const uint32_t S = F.lds_slots;
    const uint32_t s = blockIdx.x * 64 + threadIdx.x;
    const bool in_range = s < S;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.y == 0) {
        // survivor counts of the scan's workgroups -> Filter #itemsOut (wave 0 of the first block)
        unsigned long long c = 0;
        for (uint32_t b = threadIdx.x; b < nblocks_total; b += 64) c += F.block_selected[b];
        for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
        if (threadIdx.x == 0 && c) atomicAdd(F.rows_selected, c);
    }
    const uint32_t Y = 16 * gridDim.y, y = blockIdx.y * 16 + threadIdx.y;
    const size_t slab_words = (size_t)P.lds_words * S * Y;  // stride between the slabs this thread reads
    const uint32_t nblocks = (in_range && y < nblocks_total) ? (nblocks_total - y + Y - 1) / Y : 0;
    const uint64_t* base = F.slabs + (size_t)y * P.lds_words * S + (in_range ? s : 0);
    uint64_t touched = 0;
#pragma unroll 8
    for (uint32_t b = 0; b < nblocks; b++) touched |= base[b * slab_words] ^ kEmptyKey;
    touched = reduce_over_y<RED_OR>(red, touched);
    const bool leader = threadIdx.y == 0 && in_range && touched != 0;
    long long g = -1;
    if (leader) g = global_find_or_insert(G, fast_slot_key(F, s), F.err_flags, ngroups);
    uint64_t* grow = g >= 0 ? &G.acc[(size_t)g * P.glob_words] : nullptr;
    for (uint32_t a = 0; a < P.naggs; a++) {
        const AggSpec& ag = P.aggs[a];
        if (ag.distinct) continue;
        const uint64_t* l = base + (size_t)ag.lds_off * S;  // word i of workgroup b at l[b * slab_words + i * S]
        unsigned long long* w = grow ? (unsigned long long*)(grow + ag.glob_off) : nullptr;
        switch (ag.kind) {
            case AGG_COUNT:
            case AGG_COUNTN: {
                unsigned long long c = 0;
#pragma unroll 8
                for (uint32_t b = 0; b < nblocks; b++) c += l[b * slab_words];
                c = reduce_over_y<RED_ADD_U64>(red, c);
                if (w && c) atomicAdd(&w[0], c);
                break;
            }
            case AGG_SUM:
            case AGG_AVG: {
                unsigned long long lo = 0, hi = 0, fl = 0, n = 0;
                double fs = 0.0;
                const bool avg = ag.kind == AGG_AVG;
#pragma unroll 4
                for (uint32_t b = 0; b < nblocks; b++) {
                    const uint64_t* lb = l + b * slab_words;
                    int64_t x = (int64_t)lb[0];        // 0 when no int was added
                    double f = as_f64(lb[(size_t)S]);  // +0.0 when no float was added
                    fl |= lb[2 * (size_t)S];
                    lo += (unsigned long long)(uint32_t)x;
                    hi += (unsigned long long)(x >> 32);
                    fs += f;
                    if (avg) n += lb[3 * (size_t)S];
                }
                lo = reduce_over_y<RED_ADD_U64>(red, lo);
                hi = reduce_over_y<RED_ADD_U64>(red, hi);
                fs = as_f64(reduce_over_y<RED_ADD_F64>(red, f64_bits(fs)));
                fl = reduce_over_y<RED_OR>(red, fl);
                if (avg) n = reduce_over_y<RED_ADD_U64>(red, n);
                if (!w || !fl) break;
                if (fl & (SF_NONNEG_INT | SF_NEG_INT)) { atomicAdd(&w[0], lo); atomicAdd(&w[1], hi); }
                if (fl & SF_FLOAT) atomicAdd((double*)&w[2], fs);
                atomicOr(&w[3], fl);
                if (avg) atomicAdd(&w[4], n);
                break;
            }
            default: {
                const bool mn = ag.kind == AGG_MIN;
                unsigned long long fl = 0;
                long long iv = mn ? INT64_MAX : INT64_MIN;
                unsigned long long fv = mn ? ~0ull : 0ull, sv = mn ? ~0ull : 0ull;
#pragma unroll 4
                for (uint32_t b = 0; b < nblocks; b++) {
                    const uint64_t* lb = l + b * slab_words;
                    fl |= lb[0];
                    long long x = (long long)lb[(size_t)S];  // identities of untouched slabs never win
                    unsigned long long yv = lb[2 * (size_t)S], z = lb[3 * (size_t)S];
                    iv = mn ? (x < iv ? x : iv) : (x > iv ? x : iv);
                    fv = mn ? (yv < fv ? yv : fv) : (yv > fv ? yv : fv);
                    sv = mn ? (z < sv ? z : sv) : (z > sv ? z : sv);
                }
                fl = reduce_over_y<RED_OR>(red, fl);
                if (mn) {
                    iv = (long long)reduce_over_y<RED_MIN_I64>(red, (uint64_t)iv);
                    fv = reduce_over_y<RED_MIN_U64>(red, fv);
                    sv = reduce_over_y<RED_MIN_U64>(red, sv);
                } else {
                    iv = (long long)reduce_over_y<RED_MAX_I64>(red, (uint64_t)iv);
                    fv = reduce_over_y<RED_MAX_U64>(red, fv);
                    sv = reduce_over_y<RED_MAX_U64>(red, sv);
                }
                if (!w || !fl) break;
                atomicOr(&w[0], fl);
                if (fl & MM_INT) { if (mn) atomicMin((long long*)&w[1], iv); else atomicMax((long long*)&w[1], iv); }
                if (fl & MM_FLOAT) { if (mn) atomicMin(&w[2], fv); else atomicMax(&w[2], fv); }
                if (fl & MM_STRING) { if (mn) atomicMin(&w[3], sv); else atomicMax(&w[3], sv); }
                break;
            }
        }
    }
    if (T.enabled) {
        // every workgroup: its atomics are out (fence), then one ticket; the last one sees everybody's (fence) and runs the tail
        __shared__ uint32_t s_last;
        const uint32_t lt = threadIdx.y * 64 + threadIdx.x;
        // (one release fence per workgroup, behind the barrier that has every wave's atomics out; a fence by every thread —
        //  4096 L2 write-backs and invalidations per CU's worth of workgroups — made the kernel 50 us longer)
        __syncthreads();
        if (lt == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            s_last = atomicAdd(T.done, 1u) == gridDim.x * gridDim.y - 1u ? 1u : 0u;
        }
        __syncthreads();
        if (!s_last) return;
        if (lt == 0) *T.done = 0;  // (the next launch counts from zero)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        finalize_small_body(P, G, T.out_keys, T.out_aggs, T.out_parts, T.out_rep, T.counters, T.host_counters, T.max_out, F.err_flags,
                            T.clear, lt);
    }
}

// ------------------------------------------------------------------ multi-GPU: exchange of PARTIAL GROUPS
//
// When groups are few next to rows, every GPU first aggregates its own shard (scan kernels above) and only the
// partial groups travel: hash-partitioned on the group key so that each key is finished on exactly one GPU
// (≙ CumulateIntermediate on the owner, execution/group_intermediate.go:91-101).  Region d of `out` is
//   [count u64][reserved u64][keys: cap x u64][acc: cap x glob_words x u64]
__global__ void export_partials_kernel(const Program P, const GlobalTable G, uint32_t nparts, uint64_t cap,
                                       uint64_t* out, uint64_t region_words, uint32_t* err_flags) {
    uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // header word 1 of EVERY region carries this sender's verdict to all receivers (so that the ranks agree on a
    // retry without a second collective): bit 0 = some region overflowed, bit 1 = keys hold device-local wide codes
    if (s == 0 && P.wide_count && *P.wide_count)
        for (uint32_t r = 0; r < nparts; r++) atomicOr((unsigned long long*)&out[(size_t)r * region_words + 1], 2ull);
    const uint64_t key = s < G.capacity ? G.keys[s] : kEmptyKey;
    const bool used = key != kEmptyKey;
    const uint32_t d = used ? (uint32_t)(((mix64(key) >> 32) * (uint64_t)nparts) >> 32) : 0xFFFFFFFFu;
    // positions: one atomic per wave and destination (a thousand same-address atomics cost ~10 us on their own)
    unsigned long long pos = 0;
    unsigned long long todo = __ballot(used);
    const uint32_t lane = threadIdx.x & 63;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t dl = __shfl(d, leader, 64);
        const unsigned long long same = __ballot(used && d == dl);
        unsigned long long base = 0;
        if ((int)lane == leader) base = atomicAdd((unsigned long long*)&out[(size_t)dl * region_words], (unsigned long long)__popcll(same));
        base = __shfl(base, leader, 64);
        if (used && d == dl) pos = base + (unsigned long long)__popcll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
    }
    if (!used) return;
    uint64_t* region = out + (size_t)d * region_words;
    if (pos >= cap) {
        if (!(atomicOr(err_flags, (uint32_t)ERR_TABLE_FULL) & ERR_TABLE_FULL))
            for (uint32_t r = 0; r < nparts; r++) atomicOr((unsigned long long*)&out[(size_t)r * region_words + 1], 1ull);
        return;
    }
    region[2 + pos] = key;
    uint64_t* acc = region + 2 + cap + pos * P.glob_words;
    for (uint32_t w = 0; w < P.glob_words; w++) acc[w] = G.acc[(size_t)s * P.glob_words + w];
}

// CumulateIntermediate of one partial group (packed key + raw global accumulators) into the table
N1K_DEV void merge_partial_record(const Program& P, const GlobalTable& G, uint64_t key, const uint64_t* l, uint32_t* err_flags,
                                  bool& fresh, bool unique_keys) {
    long long g = global_find_or_insert_quiet(G, key, err_flags, fresh);
    if (g < 0) return;
    if (fresh && unique_keys) {
        // no other record of this launch has the key: the new row is this record (plain stores, no atomics)
        uint64_t* w = &G.acc[(size_t)g * P.glob_words];
        for (uint32_t i = 0; i < P.glob_words; i++) w[i] = l[i];
        return;
    }
    for (uint32_t a = 0; a < P.naggs; a++) {
        const AggSpec& ag = P.aggs[a];
        const uint64_t* p = l + ag.glob_off;
        unsigned long long* w = (unsigned long long*)&G.acc[(size_t)g * P.glob_words + ag.glob_off];
        if (ag.distinct) {  // set members do not travel with partial groups: the caller uses the row exchange
            atomicOr(err_flags, (uint32_t)ERR_UNSUPPORTED_VALUE);
            continue;
        }
        switch (ag.kind) {
            case AGG_COUNT:
            case AGG_COUNTN:
                if (p[0]) atomicAdd(&w[0], (unsigned long long)p[0]);
                break;
            case AGG_SUM:
            case AGG_AVG:
                if (!p[3]) break;
                atomicAdd(&w[0], (unsigned long long)p[0]);
                atomicAdd(&w[1], (unsigned long long)p[1]);
                if (p[3] & SF_FLOAT) atomicAdd((double*)&w[2], as_f64(p[2]));
                atomicOr(&w[3], (unsigned long long)p[3]);
                if (ag.kind == AGG_AVG) atomicAdd(&w[4], (unsigned long long)p[4]);
                break;
            default: {
                if (!p[0]) break;
                bool mn = ag.kind == AGG_MIN;
                atomicOr(&w[0], (unsigned long long)p[0]);
                if (p[0] & MM_INT) { if (mn) atomicMin((long long*)&w[1], (long long)p[1]); else atomicMax((long long*)&w[1], (long long)p[1]); }
                if (p[0] & MM_FLOAT) { if (mn) atomicMin(&w[2], (unsigned long long)p[2]); else atomicMax(&w[2], (unsigned long long)p[2]); }
                if (p[0] & MM_STRING) { if (mn) atomicMin(&w[3], (unsigned long long)p[3]); else atomicMax(&w[3], (unsigned long long)p[3]); }
                break;
            }
        }
    }
}

// what a receiver makes of the verdict words it was sent (n1k_types.h VD_*): error flags that void the step on every rank
// alike, and the largest host status any sender reported
N1K_DEV void raise_verdict(uint64_t verdict, uint32_t* err_flags) {
    uint32_t f = 0;
    if (verdict & VD_OVERFLOW) f |= ERR_EXCHANGE_OVERFLOW;
    if (verdict & VD_WIDE) f |= ERR_EXCHANGE_WIDE;
    if (verdict & VD_UNPACKABLE) f |= ERR_PEER_UNPACKABLE;
    if (verdict & VD_UNSUPPORTED) f |= ERR_PEER_UNSUPPORTED;
    if (verdict >> VD_STATUS_SHIFT) f |= ERR_PEER_FAILED;
    atomicOr(err_flags, f);
}

// merge received partial groups into this GPU's table (one thread per record)
__global__ void merge_partials_kernel(const Program P, const GlobalTable G, uint32_t nregions, uint64_t cap,
                                      const uint64_t* in, uint64_t region_words, uint32_t* err_flags,
                                      unsigned long long* ngroups, uint32_t unique_keys) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t r = (uint32_t)(i / cap);
    uint64_t pos = i % cap;
    if (r >= nregions) return;
    // a sender that could not export (see export_partials_kernel) voids the whole exchange: nothing is merged and
    // n1k_finish reports it, on every rank alike
    uint64_t verdict = 0, status = 0;
    for (uint32_t q = 0; q < nregions; q++) {
        const uint64_t v = in[(size_t)q * region_words + 1];
        verdict |= v;
        status = (v >> VD_STATUS_SHIFT) > status ? (v >> VD_STATUS_SHIFT) : status;
    }
    if (verdict) {
        if (i == 0) {
            raise_verdict(verdict, err_flags);
            if (status) atomicMax((unsigned long long*)err_flags + kPeerStatusFromErr, (unsigned long long)(status & 0xFF));
        }
        return;
    }
    const uint64_t* region = in + (size_t)r * region_words;
    uint64_t count = region[0] < cap ? region[0] : cap;
    bool fresh = false;
    if (pos < count) merge_partial_record(P, G, region[2 + pos], region + 2 + cap + pos * P.glob_words, err_flags, fresh, unique_keys != 0);
    // new groups: one atomic on the counter per wave (per new group they serialise on one address: 4 ns each)
    const unsigned long long m = __ballot(fresh);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(ngroups, (unsigned long long)__popcll(m));
}

// MIN / MAX over strings keep (rank << 32 | code) so that atomicMin / atomicMax order them bytewise; the ranks belong to
// the dictionary as it was when the row was seen.  When the dictionary has grown since (new strings between batches)
// the winners kept so far are re-stamped with their ranks in the new order before any new row is compared with them.
__global__ void restamp_ranks_kernel(const Program P, const GlobalTable G) {
    uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= G.capacity || G.keys[s] == kEmptyKey) return;
    for (uint32_t a = 0; a < P.naggs; a++) {
        const AggSpec& ag = P.aggs[a];
        if (ag.distinct || (ag.kind != AGG_MIN && ag.kind != AGG_MAX)) continue;
        uint64_t* w = &G.acc[s * P.glob_words + ag.glob_off];
        if (w[0] & MM_STRING) {
            const uint32_t code = (uint32_t)w[3];
            w[3] = ((uint64_t)P.str_rank[code] << 32) | code;
        }
    }
}

hipError_t launch_restamp_ranks(const Program& P, const GlobalTable& G, hipStream_t st) {
    if (!G.capacity) return hipSuccess;
    hipLaunchKernelGGL(restamp_ranks_kernel, dim3((uint32_t)((G.capacity + 255) / 256)), dim3(256), 0, st, P, G);
    return hipGetLastError();
}

// grow the global table: re-insert every occupied slot (keys keep their packed form)
__global__ void rehash_kernel(const Program P, const GlobalTable oldt, const GlobalTable newt, uint32_t* err_flags,
                              unsigned long long* scratch) {
    uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= oldt.capacity) return;
    uint64_t key = oldt.keys[s];
    if (key == kEmptyKey) return;
    bool fresh = false;  // (the group count does not change: no counter)
    long long g = global_find_or_insert_quiet(newt, key, err_flags, fresh);
    if (g < 0) return;
    for (uint32_t w = 0; w < P.glob_words; w++) newt.acc[(size_t)g * P.glob_words + w] = oldt.acc[(size_t)s * P.glob_words + w];
    if (newt.rep_row && oldt.rep_row) newt.rep_row[g] = oldt.rep_row[s];
}

// ------------------------------------------------------------------ K5: FinalGroup

N1K_DEV void put_value(OutValue* o, uint32_t tag, uint64_t p) {
    o->tag = tag;
    o->payload = p;
}

// ComputeFinal per aggregate (algebra/agg_sum.go:109-111, agg_count.go:128-130, agg_avg.go:111-129,
// agg_min.go:107-109, agg_count_distinct.go:118-126) + the mergeable partial
N1K_DEV void finalize_agg(const Program& P, const AggSpec& ag, const uint64_t* g, OutValue* fin, OutPartial* part,
                          uint32_t* err_flags) {
    const uint64_t* w = g + ag.glob_off;
    OutPartial pt;
    pt.count = 0; pt.isum = 0; pt.fsum = 0.0; pt.flags = 0; pt.ext_tag = T_NULL; pt.ext_payload = 0; pt.distinct = 0;
    if (ag.kind == AGG_ARRAY) {  // the host builds the array from the logged operands (n1k_finish); NULL when there are none
        put_value(fin, T_NULL, 0);
        *part = pt;
        return;
    }
    if (ag.distinct) {
        pt.distinct = (int64_t)w[0];
        if (ag.kind == AGG_COUNT || ag.kind == AGG_COUNTN) {
            put_value(fin, T_INT, w[0]);  // Set.Len() (agg_count_distinct.go:118-126)
        } else if (w[0] == 0) {
            put_value(fin, T_NULL, 0);  // empty set -> NULL (agg_sum_distinct.go:119-121)
        } else {
            // sum := ZERO_NUMBER; for each member sum = sum.Add(v): the fold starts from int 0, so one negative
            // member already makes it a float (value/integer.go:266-277)
            uint64_t fl = w[8];
            __int128 tot = ((__int128)(int64_t)w[6] << 32) + (__int128)(unsigned __int128)w[5];
            bool fits = tot >= (__int128)INT64_MIN && tot <= (__int128)INT64_MAX;
            bool int_exact = !(fl & SF_FLOAT) && !(fl & SF_NEG_INT) && fits;
            double itot = fits ? (double)(int64_t)tot : (double)(int64_t)w[6] * 4294967296.0 + (double)w[5];
            double fsum = as_f64(w[7]);
            pt.isum = int_exact ? (int64_t)tot : 0;
            pt.fsum = int_exact ? 0.0 : fsum + itot;
            pt.flags = (int_exact ? 1u : 0u) | ((fl & SF_FLOAT) ? 2u : 0u);
            if (ag.kind == AGG_SUM) {
                if (int_exact) put_value(fin, T_INT, (uint64_t)(int64_t)tot);
                else put_value(fin, T_FLOAT, f64_bits(fsum + itot));
            } else {
                double avg = (int_exact ? (double)(int64_t)tot : fsum + itot) / (double)w[0];  // agg_avg_distinct.go:133
                if (is_int_f64(avg)) put_value(fin, T_INT, (uint64_t)go_f2i(avg));
                else put_value(fin, T_FLOAT, f64_bits(avg));
            }
        }
        *part = pt;
        return;
    }
    switch (ag.kind) {
        case AGG_COUNT:
        case AGG_COUNTN:
            pt.count = (int64_t)w[0];
            put_value(fin, T_INT, w[0]);
            break;
        case AGG_SUM:
        case AGG_AVG: {
            uint64_t fl = w[3];
            bool has_nn = fl & SF_NONNEG_INT, has_ng = fl & SF_NEG_INT, has_f = fl & SF_FLOAT;
            // exact 128-bit integer total = hi * 2^32 + lo
            __int128 tot = ((__int128)(int64_t)w[1] << 32) + (__int128)(unsigned __int128)w[0];
            bool fits = tot >= (__int128)INT64_MIN && tot <= (__int128)INT64_MAX;
            // intValue.Add keeps an int only for same-sign operands without overflow (value/integer.go:266-277)
            bool int_exact = !has_f && !(has_nn && has_ng) && fits;
            double itot = (double)(int64_t)w[1] * 4294967296.0 + (double)w[0];
            if (fits) itot = (double)(int64_t)tot;
            double fsum = as_f64(w[2]);
            uint64_t n = ag.kind == AGG_AVG ? w[4] : 0;
            pt.count = (int64_t)n;
            pt.isum = int_exact ? (int64_t)tot : 0;
            pt.fsum = int_exact ? fsum : fsum + itot;
            pt.flags = (int_exact ? 1u : 0u) | (has_f ? 2u : 0u);
            if (fl == 0) {
                put_value(fin, T_NULL, 0);  // Default(): NULL (agg_sum.go:77, agg_avg.go:77)
            } else if (ag.kind == AGG_SUM) {
                if (int_exact) put_value(fin, T_INT, (uint64_t)(int64_t)tot);
                else put_value(fin, T_FLOAT, f64_bits(fsum + itot));
            } else {
                double sm = int_exact ? (double)(int64_t)tot : fsum + itot;
                double avg = sm / (double)n;  // agg_avg.go:124-125 -> value.NewValue folds integral results
                if (is_int_f64(avg)) put_value(fin, T_INT, (uint64_t)go_f2i(avg));
                else put_value(fin, T_FLOAT, f64_bits(avg));
            }
            break;
        }
        default: {  // MIN / MAX over the full collation (agg_min.go:117-127)
            uint64_t fl = w[0];
            bool mn = ag.kind == AGG_MIN;
            uint32_t tag = T_NULL;
            uint64_t p = 0;
            bool has_num = fl & (MM_INT | MM_FLOAT);
            bool has_bool = fl & (MM_FALSE | MM_TRUE);
            int which = 0;  // 1 bool, 2 number, 3 string, 4 other
            if (mn) which = has_bool ? 1 : (has_num ? 2 : ((fl & MM_STRING) ? 3 : ((fl & MM_OTHER) ? 4 : 0)));
            else which = (fl & MM_OTHER) ? 4 : ((fl & MM_STRING) ? 3 : (has_num ? 2 : (has_bool ? 1 : 0)));
            if (which == 1) {
                tag = mn ? ((fl & MM_FALSE) ? T_FALSE : T_TRUE) : ((fl & MM_TRUE) ? T_TRUE : T_FALSE);
            } else if (which == 2) {
                bool hi = fl & MM_INT, hf = fl & MM_FLOAT;
                double f = f64_unsortable(w[2]);
                int64_t iv = (int64_t)w[1];
                bool pick_int = hi;
                if (hi && hf) {
                    int c = collate_f64((double)iv, f);
                    pick_int = mn ? c <= 0 : c >= 0;
                }
                if (pick_int) { tag = T_INT; p = (uint64_t)iv; }
                else { tag = T_FLOAT; p = f64_bits(f); }
            } else if (which == 3) {
                tag = T_STRING;
                p = w[3] & 0xFFFFFFFFull;
            } else if (which == 4) {
                atomicOr(err_flags, (uint32_t)ERR_UNSUPPORTED_VALUE);
            }
            pt.ext_tag = tag;
            pt.ext_payload = p;
            put_value(fin, tag, p);
            break;
        }
    }
    *part = pt;
}

// Output positions are handed out per workgroup CHUNK (count the chunk's groups, one atomic, then ranks inside the
// chunk): one same-address atomic per group cost 24 ms for 6.4 M groups.
constexpr uint32_t kFinalizeChunkMax = 16384;  // table slots per workgroup: capacity / 4096, within [256, 16384]
__global__ __launch_bounds__(256) void finalize_kernel(const Program P, const GlobalTable G, OutValue* out_keys, OutValue* out_aggs,
                                                       OutPartial* out_parts, uint64_t* out_rep, unsigned long long* out_count,
                                                       uint64_t max_out, uint32_t* err_flags, uint32_t chunk) {
    __shared__ uint32_t wave_cnt[4];
    __shared__ unsigned long long chunk_base;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t lo = (uint64_t)blockIdx.x * chunk;
    const uint64_t hi = lo + chunk < G.capacity ? lo + chunk : G.capacity;
    // pass 1: groups in the chunk
    uint32_t mine = 0;
    for (uint64_t s = lo + tid; s < hi; s += 256) mine += G.keys[s] != kEmptyKey ? 1u : 0u;
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off, 64);
    if (lane == 0) wave_cnt[wave] = mine;
    __syncthreads();
    if (tid == 0) {
        const uint32_t total = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        chunk_base = total ? atomicAdd(out_count, (unsigned long long)total) : 0ull;
    }
    __syncthreads();
    unsigned long long next = chunk_base;
    // pass 2: the same slots in the same order; positions from ballots over each step of 256 slots
    for (uint64_t s0 = lo; s0 < hi; s0 += 256) {
        const uint64_t s = s0 + tid;
        const uint64_t key = s < hi ? G.keys[s] : kEmptyKey;
        const bool used = key != kEmptyKey;
        const unsigned long long m = __ballot(used);
        __syncthreads();
        if (lane == 0) wave_cnt[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = 0;
        for (uint32_t w = 0; w < wave; w++) before += wave_cnt[w];
        const uint32_t step_total = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        const unsigned long long idx = next + before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        next += step_total;
        if (!used || idx >= max_out) continue;
        for (uint32_t k = 0; k < P.nkeys; k++) {
            const KeySpec& ks = P.keys[k];
            uint64_t field = ks.bits >= 64 ? key : ((key >> ks.shift) & ((1ull << ks.bits) - 1ull));
            uint32_t tag;
            uint64_t p;
            unpack_key_field(P, ks.mode, field, tag, p);
            put_value(&out_keys[idx * P.nkeys + k], tag, p);
        }
        const uint64_t* g = &G.acc[(size_t)s * P.glob_words];
        for (uint32_t a = 0; a < P.naggs; a++)
            finalize_agg(P, P.aggs[a], g, &out_aggs[idx * P.naggs + a], &out_parts[idx * P.naggs + a], err_flags);
        if (out_rep) out_rep[idx] = P.emit_packed_key ? key : (G.rep_row ? G.rep_row[s] : ~0ull);
    }
}

// FinalGroup of a SMALL table (capacity <= kFinalizeSmallMax slots) as the one last kernel of a query: one workgroup finalises
// every group in table order, publishes the counters (with the number of output rows in [2]) into pinned host memory
// behind the rows, and — `clear` — leaves the table and the counters as n1k_reset would: the next execution of the operator
// over resident columns (n1k_run_device_batch) then starts with its scan, no reopen kernel in front of it.  Replaces
// finalize_kernel + publish_counters_kernel (+ the next init_table_kernel): three dependent launches of a few microseconds
// each, which is what a query over 10 M cached rows is made of.
constexpr uint32_t kFinalizeSmallMax = 8192;
N1K_DEV void finalize_small_body(const Program& P, const GlobalTable& G, OutValue* out_keys, OutValue* out_aggs, OutPartial* out_parts,
                                 uint64_t* out_rep, unsigned long long* counters, unsigned long long* host_counters, uint64_t max_out,
                                 uint32_t* err_flags, uint32_t clear, uint32_t tid) {
    __shared__ uint32_t wave_cnt[16];
    const uint32_t lane = tid & 63, wave = tid >> 6;
    unsigned long long next = 0;
    for (uint64_t s0 = 0; s0 < G.capacity; s0 += 1024) {
        const uint64_t s = s0 + tid;
        const uint64_t key = s < G.capacity ? G.keys[s] : kEmptyKey;
        const bool used = key != kEmptyKey;
        const unsigned long long m = __ballot(used);
        __syncthreads();
        if (lane == 0) wave_cnt[wave] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = 0, step_total = 0;
        for (uint32_t w = 0; w < 16; w++) {
            before += w < wave ? wave_cnt[w] : 0u;
            step_total += wave_cnt[w];
        }
        const unsigned long long idx = next + before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        next += step_total;
        if (!used) continue;
        uint64_t* g = &G.acc[(size_t)s * P.glob_words];
        if (idx < max_out) {
            for (uint32_t k = 0; k < P.nkeys; k++) {
                const KeySpec& ks = P.keys[k];
                uint64_t field = ks.bits >= 64 ? key : ((key >> ks.shift) & ((1ull << ks.bits) - 1ull));
                uint32_t tag;
                uint64_t p;
                unpack_key_field(P, ks.mode, field, tag, p);
                put_value(&out_keys[idx * P.nkeys + k], tag, p);
            }
            for (uint32_t a = 0; a < P.naggs; a++)
                finalize_agg(P, P.aggs[a], g, &out_aggs[idx * P.naggs + a], &out_parts[idx * P.naggs + a], err_flags);
            if (out_rep) out_rep[idx] = P.emit_packed_key ? key : (G.rep_row ? G.rep_row[s] : ~0ull);
        }
        if (clear) {
            G.keys[s] = kEmptyKey;
            if (G.rep_row) G.rep_row[s] = ~0ull;
            glob_row_init(P, g);
        }
    }
    __syncthreads();  // (every error flag of this launch is raised)
    if (tid == 0) counters[2] = next;
    __syncthreads();
    if (tid < kCounters) {
        host_counters[tid] = __hip_atomic_load(&counters[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (clear) counters[tid] = 0;
    }
}

__global__ __launch_bounds__(1024) void finalize_small_kernel(const Program P, const GlobalTable G, OutValue* out_keys, OutValue* out_aggs,
                                                             OutPartial* out_parts, uint64_t* out_rep, unsigned long long* counters,
                                                             unsigned long long* host_counters, uint64_t max_out, uint32_t* err_flags,
                                                             uint32_t clear) {
    finalize_small_body(P, G, out_keys, out_aggs, out_parts, out_rep, counters, host_counters, max_out, err_flags, clear, threadIdx.x);
}

hipError_t launch_finalize_small(const Program& P, const GlobalTable& G, OutValue* out_keys, OutValue* out_aggs, OutPartial* out_parts,
                                 uint64_t* out_rep, unsigned long long* counters, unsigned long long* host_counters, uint64_t max_out,
                                 uint32_t* err_flags, bool clear, hipStream_t st) {
    if (G.capacity > kFinalizeSmallMax) return hipErrorInvalidValue;
    hipLaunchKernelGGL(finalize_small_kernel, dim3(1), dim3(1024), 0, st, P, G, out_keys, out_aggs, out_parts, out_rep, counters,
                       host_counters, max_out, err_flags, clear ? 1u : 0u);
    return hipGetLastError();
}

// FinalGroup straight from a region of partial groups with unique keys ([count][0][keys: cap][accumulators]): record i
// is output row i (no table, no position counter).  Two lean forms for ORDER BY ... LIMIT over millions of groups (only
// the top-k candidates need rows; writing every group's row was 0.72 GB per 6.4 M groups):
//   ord != nullptr   every group is finalised (errors are raised as ever) but only the value of the first ORDER BY term
//                    is stored, ord[i] — what the top-k selection reads;
//                    is stored, ord[i] — what the top-k selection reads; with `images` its 64-bit order image instead (8 B per
//                    group, and no second kernel that turns 16-byte values into images);
//   cand != nullptr  the groups cand[0 .. count) only, group cand[j] as output row j.
N1K_DEV uint64_t order_image(const Program& P, uint64_t tag, uint64_t p, bool desc);
__global__ void finalize_region_kernel(const Program P, const uint64_t* region, uint64_t cap, uint64_t count, OutValue* out_keys,
                                       OutValue* out_aggs, OutPartial* out_parts, uint64_t* out_rep, uint32_t* err_flags,
                                       const uint32_t* cand, OutValue* ord, uint32_t ord_is_key, uint32_t ord_index, uint64_t* images,
                                       uint32_t desc) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    const uint64_t i = cand ? (uint64_t)cand[j] : j;
    const uint64_t key = region[2 + i];
    const uint64_t* g = region + 2 + cap + i * P.glob_words;
    if (ord || images) {
        OutValue v, term;
        OutPartial part;
        term.tag = T_MISSING;
        term.payload = 0;
        for (uint32_t a = 0; a < P.naggs; a++) {
            finalize_agg(P, P.aggs[a], g, &v, &part, err_flags);
            if (!ord_is_key && a == ord_index) term = v;
        }
        if (ord_is_key) {
            const KeySpec& ks = P.keys[ord_index];
            uint64_t field = ks.bits >= 64 ? key : ((key >> ks.shift) & ((1ull << ks.bits) - 1ull));
            uint32_t tag;
            uint64_t p;
            unpack_key_field(P, ks.mode, field, tag, p);
            put_value(&term, tag, p);
        }
        if (images) images[i] = order_image(P, term.tag, term.payload, desc != 0);
        else ord[i] = term;
        return;
    }
    for (uint32_t k = 0; k < P.nkeys; k++) {
        const KeySpec& ks = P.keys[k];
        uint64_t field = ks.bits >= 64 ? key : ((key >> ks.shift) & ((1ull << ks.bits) - 1ull));
        uint32_t tag;
        uint64_t p;
        unpack_key_field(P, ks.mode, field, tag, p);
        put_value(&out_keys[j * P.nkeys + k], tag, p);
    }
    for (uint32_t a = 0; a < P.naggs; a++)
        finalize_agg(P, P.aggs[a], g, &out_aggs[j * P.naggs + a], &out_parts[j * P.naggs + a], err_flags);
    if (out_rep) out_rep[j] = ~0ull;
}

// ------------------------------------------------------------------ high-cardinality GROUP BY: partition, then LDS
//
// With millions of groups the workgroup tables stop absorbing anything and every row costs 3-5 atomics on a table in
// HBM (≈ 20 G/s: config 5's keys ran at 28 ms per 100 M rows).  Instead: (1) Filter + group key + the aggregates'
// operands are projected to records; (2) the records are radix partitioned on bits of mix64(key) (radix_*_kernel,
// the record arrays follow the key) until a bin holds about a thousand groups; (3) one workgroup per bin runs
// InitialGroup in an LDS table and hands the bin's groups to the global table once (≙ IntermediateGroup).  All rows
// of a key meet in one bin, so a group costs a handful of global atomics instead of its rows.
template <int R, int BLOCK>
__global__ __launch_bounds__(BLOCK) void project_records_kernel(const Program P, const ProjectArgs A) {
    __shared__ uint32_t wave_cnt[BLOCK / 64];
    __shared__ unsigned long long tile_base;
    __shared__ uint32_t s_hist[256];  // first radix digit of the records' keys: the first partition pass needs no histogram pass
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < 256; i += BLOCK) s_hist[i] = 0;
    __syncthreads();
    uint32_t unsupported = 0, unpackable = 0;
    const uint64_t tile_rows = (uint64_t)BLOCK * R;
    const uint64_t ntiles = (A.nrows + tile_rows - 1) / tile_rows;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        uint64_t row[R];
        bool valid[R], pass[R];
#pragma unroll
        for (int j = 0; j < R; j++) {
            row[j] = tile * tile_rows + (uint64_t)j * BLOCK + tid;
            valid[j] = row[j] < A.nrows;
        }
        eval_predicate<R>(P, row, valid, pass, unsupported);
        uint64_t key[R];
#pragma unroll
        for (int j = 0; j < R; j++) key[j] = 0;
        for (uint32_t k = 0; k < P.nkeys; k++) {
            const KeySpec& ks = P.keys[k];
            uint32_t kt[R];
            uint64_t kp[R];
            load_operand<R>(P, ks.src, row, pass, kt, kp);
#pragma unroll
            for (int j = 0; j < R; j++) {
                uint64_t f = 0, canon = 0;
                if (pass[j] && !pack_key_field(P, ks, kt[j], kp[j], f, canon)) {
                    unpackable = 1;
                    pass[j] = false;
                }
                key[j] |= f << ks.shift;
            }
        }
        uint32_t mine = 0;
#pragma unroll
        for (int j = 0; j < R; j++) {
            mine += pass[j] ? 1u : 0u;
            if (pass[j] && A.hist) atomicAdd(&s_hist[radix_bin(key[j], 56)], 1u);
        }
        unsigned long long pos = tile_reserve<BLOCK>(mine, A.cursor, wave_cnt, &tile_base, tid);
        unsigned long long q = pos;
#pragma unroll
        for (int j = 0; j < R; j++)
            if (pass[j] && q < A.capacity) A.out.key[q++] = key[j];
        for (uint32_t e = 0; e < A.nsrc; e++) {
            uint32_t vt[R];
            uint64_t vp[R];
            load_operand<R>(P, A.src[e], row, pass, vt, vp);
            q = pos;
#pragma unroll
            for (int j = 0; j < R; j++)
                if (pass[j] && q < A.capacity) {
                    A.out.pay[e][q] = vp[j];
                    A.out.tag[e][q] = (uint8_t)vt[j];
                    q++;
                }
        }
        if (pos + mine > A.capacity) atomicOr(A.err_flags, (uint32_t)ERR_TABLE_FULL);
        __syncthreads();
    }
    if (unsupported) atomicOr(A.err_flags, (uint32_t)ERR_UNSUPPORTED_VALUE);
    if (unpackable) atomicOr(A.err_flags, (uint32_t)ERR_UNPACKABLE_KEY);
    __syncthreads();
    if (A.hist)
        for (uint32_t i = tid; i < 256; i += BLOCK)
            if (s_hist[i]) atomicAdd(&A.hist[i], (unsigned long long)s_hist[i]);
}

// How many groups do the first rows of a batch bring?  Filter + group key only: every surviving row's key is looked up /
// inserted in the global table (its accumulators stay at their identities; the rows themselves are aggregated later,
// by whichever path the answer selects).
template <int R, int BLOCK>
__global__ __launch_bounds__(BLOCK) void probe_keys_kernel(const Program P, uint64_t nrows, const GlobalTable G, uint32_t* err_flags,
                                                          unsigned long long* ngroups) {
    const uint32_t tid = threadIdx.x;
    uint32_t unsupported = 0, fresh = 0;
    const uint64_t tile_rows = (uint64_t)BLOCK * R;
    const uint64_t ntiles = (nrows + tile_rows - 1) / tile_rows;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        uint64_t row[R];
        bool valid[R], pass[R];
#pragma unroll
        for (int j = 0; j < R; j++) {
            row[j] = tile * tile_rows + (uint64_t)j * BLOCK + tid;
            valid[j] = row[j] < nrows;
        }
        eval_predicate<R>(P, row, valid, pass, unsupported);
        uint64_t key[R];
#pragma unroll
        for (int j = 0; j < R; j++) key[j] = 0;
        for (uint32_t k = 0; k < P.nkeys; k++) {
            const KeySpec& ks = P.keys[k];
            uint32_t kt[R];
            uint64_t kp[R];
            load_operand<R>(P, ks.src, row, pass, kt, kp);
#pragma unroll
            for (int j = 0; j < R; j++) {
                uint64_t f = 0, canon = 0;
                if (pass[j] && !pack_key_field(P, ks, kt[j], kp[j], f, canon)) pass[j] = false;  // reported by the real pass
                key[j] |= f << ks.shift;
            }
        }
#pragma unroll
        for (int j = 0; j < R; j++) {
            bool f = false;
            if (pass[j]) (void)global_find_or_insert_quiet(G, key[j], err_flags, f);
            fresh += f ? 1u : 0u;
        }
    }
    // new groups: one atomic per workgroup (same-address atomics cost ~10 ns each; a probe of all-new keys spent 0.4 ms
    // on one per wave and row)
    __shared__ unsigned int block_fresh;
    if (tid == 0) block_fresh = 0;
    __syncthreads();
    for (int off = 32; off > 0; off >>= 1) fresh += __shfl_down(fresh, off, 64);
    if ((tid & 63) == 0 && fresh) atomicAdd(&block_fresh, fresh);
    __syncthreads();
    if (tid == 0 && block_fresh) atomicAdd(ngroups, (unsigned long long)block_fresh);
}

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void agg_bins_kernel(const Program P, const BinAggArgs A, const GlobalTable G,
                                                        unsigned long long* ngroups) {
    extern __shared__ uint64_t lds[];
    __shared__ uint32_t lds_fill;
    __shared__ uint32_t emit_wave_cnt[BLOCK / 64];
    __shared__ unsigned long long emit_base;
    const uint32_t S = A.lds_slots, tid = threadIdx.x;
    uint32_t new_groups = 0;
    const Rec16* const rec16 = (const Rec16*)A.rec;
    for (uint32_t bin = blockIdx.x; bin < A.nbins; bin += gridDim.x) {
        uint64_t lo, hi;
        if (rec16) {
            const uint64_t c = A.bin_count[(size_t)bin * (A.bin_count_stride ? A.bin_count_stride : 1u)];
            lo = (uint64_t)bin * A.bin_stride;
            hi = lo + (c < A.bin_stride ? c : A.bin_stride);
        } else {
            lo = A.bin_start[bin];
            hi = A.bin_start[bin + 1];
        }
        if (lo == hi) continue;
        __syncthreads();
        lds_table_init<BLOCK>(P, lds, S, tid);
        if (tid == 0) lds_fill = 0;
        __syncthreads();
        // InitialGroup over the bin's records (execution/group_initial.go:56-100)
        for (uint64_t i = lo + tid; i < hi; i += BLOCK) {
            uint64_t key;
            uint32_t vt[kRecOperands];
            uint64_t vp[kRecOperands];
            if (rec16) {
                const Rec16 r = rec16[i];
                if (r.k == kEmptyKey) continue;  // padding of the hash regions
                rec16_decode(r, key, vt[0], vp[0]);
                vt[1] = T_NULL;
                vp[1] = 0;
            } else {
                key = A.in.key[i];
                for (uint32_t e = 0; e < A.nsrc; e++) {
                    vt[e] = A.in.tag[e][i];
                    vp[e] = A.in.pay[e][i];
                }
            }
            const int slot = lds_find_or_insert(lds, S, key, &lds_fill, A.lds_max_fill);
            long long grow = -1;
            if (slot < 0 && A.emit) {
                // more groups in the bin than the LDS table takes: the row leaves as a group of its own
                const unsigned long long q = atomicAdd((unsigned long long*)&A.emit[0], 1ull);
                atomicAdd(A.emit_singletons, 1ull);  // keys in the region are no longer unique
                if (q >= A.emit_cap) {
                    atomicOr(A.err_flags, (uint32_t)ERR_TABLE_FULL);
                    continue;
                }
                A.emit[2 + q] = key;
                uint64_t* row = A.emit + 2 + A.emit_cap + q * P.glob_words;
                glob_row_init(P, row);
                for (uint32_t a = 0; a < P.naggs; a++) {
                    const uint32_t e = A.agg_src[a];
                    acc_global(P, P.aggs[a], row, e < kRecOperands ? vt[e] : (uint32_t)T_NULL, e < kRecOperands ? vp[e] : 0ull);
                }
                continue;
            }
            if (slot < 0) {  // the same, straight to the global table
                grow = global_find_or_insert(G, key, A.err_flags, ngroups);
                if (grow < 0) continue;
            }
            for (uint32_t a = 0; a < P.naggs; a++) {
                const AggSpec& ag = P.aggs[a];
                const uint32_t e = A.agg_src[a];
                const uint32_t t = e < kRecOperands ? vt[e] : (uint32_t)T_NULL;
                const uint64_t p = e < kRecOperands ? vp[e] : 0ull;
                if (slot >= 0) {
                    if (!acc_lds(P, ag, lds, S, (uint32_t)slot, t, p)) {
                        // a value the narrow LDS accumulators do not take (|int| >= 2^40)
                        if (A.emit) {  // leaves as a partial group of its own that holds just this contribution
                            const unsigned long long q = atomicAdd((unsigned long long*)&A.emit[0], 1ull);
                            atomicAdd(A.emit_singletons, 1ull);
                            if (q < A.emit_cap) {
                                A.emit[2 + q] = key;
                                uint64_t* row = A.emit + 2 + A.emit_cap + q * P.glob_words;
                                glob_row_init(P, row);
                                acc_global(P, ag, row, t, p);
                            } else
                                atomicOr(A.err_flags, (uint32_t)ERR_TABLE_FULL);
                        } else {
                            if (grow < 0) grow = global_find_or_insert(G, key, A.err_flags, ngroups);
                            if (grow >= 0) acc_global(P, ag, &G.acc[(size_t)grow * P.glob_words], t, p);
                        }
                    }
                } else
                    acc_global(P, ag, &G.acc[(size_t)grow * P.glob_words], t, p);
            }
        }
        __syncthreads();
        if (A.emit) {
            // the bin's groups leave as partial groups [key][raw accumulators] in one compact region: their exact number
            // sizes the global table before they are merged into it (merge_partials_kernel)
            uint32_t mine = 0;
            for (uint32_t s = tid; s < S; s += BLOCK) mine += lds[s] != kEmptyKey ? 1u : 0u;
            unsigned long long q = tile_reserve<BLOCK>(mine, (unsigned long long*)&A.emit[0], emit_wave_cnt, &emit_base, tid);
            for (uint32_t s = tid; s < S; s += BLOCK) {
                const uint64_t key = lds[s];
                if (key == kEmptyKey) continue;
                if (q < A.emit_cap) {
                    A.emit[2 + q] = key;
                    uint64_t* row = A.emit + 2 + A.emit_cap + q * P.glob_words;
                    glob_row_init(P, row);
                    merge_slot(P, lds, S, s, row);
                } else
                    atomicOr(A.err_flags, (uint32_t)ERR_TABLE_FULL);
                q++;
            }
            continue;
        }
        // the bin's groups go to the global table once (≙ IntermediateGroup, execution/group_intermediate.go:56-104)
        for (uint32_t s = tid; s < S; s += BLOCK) {
            const uint64_t key = lds[s];
            if (key == kEmptyKey) continue;
            bool fresh = false;
            long long g = global_find_or_insert_quiet(G, key, A.err_flags, fresh);
            if (g < 0) continue;
            new_groups += fresh ? 1u : 0u;
            merge_slot(P, lds, S, s, &G.acc[(size_t)g * P.glob_words]);
        }
    }
    // new groups: one atomic on the counter per workgroup (millions of same-address atomics cost ~4 ns each)
    __shared__ uint32_t block_new;
    __syncthreads();
    if (tid == 0) block_new = 0;
    __syncthreads();
    for (int off = 32; off > 0; off >>= 1) new_groups += __shfl_down(new_groups, off, 64);
    if ((tid & 63) == 0 && new_groups) atomicAdd(&block_new, new_groups);
    __syncthreads();
    if (tid == 0 && block_new) atomicAdd(ngroups, (unsigned long long)block_new);
}

// ------------------------------------------------------------------ ORDER BY ... LIMIT k over the groups: top-k filter
//
// execution/order_limit.go keeps the best offset+limit rows in a heap.  Here every finalised group gets a 64-bit
// ORDER IMAGE of its first sort term — monotone in value.Collate (type class in the top 3 bits; numbers through the
// float64 order image, strings through the bytewise rank of their code), not injective — a radix select finds the
// image T of the (offset+limit)-th row, and only the groups with image <= T (the answer plus ties on the first
// term) are compacted and copied to the host, which orders them exactly.  G = 6.4 M groups ship ~k rows, not 640 MB.
N1K_DEV uint64_t order_image(const Program& P, uint64_t tag, uint64_t p, bool desc) {
    uint64_t cls, body = 0;
    switch ((uint32_t)(tag & 0xFF)) {
        case T_MISSING: cls = 0; break;
        case T_NULL: cls = 1; break;
        case T_FALSE: cls = 2; break;
        case T_TRUE: cls = 2; body = 1; break;
        case T_INT: cls = 3; body = f64_sortable((double)(int64_t)p) >> 3; break;
        case T_FLOAT: cls = 3; body = f64_sortable(as_f64(p)) >> 3; break;
        case T_STRING: cls = 4; body = P.str_rank ? (uint64_t)P.str_rank[(uint32_t)p] : 0ull; break;
        case T_ARRAY: cls = 5; break;
        default: cls = 6; break;
    }
    const uint64_t img = (cls << 61) | body;
    return desc ? ~img : img;
}

constexpr uint32_t kTopkSegs = 256;  // candidate lists of the sampled path (one counter each: thousands of candidates on ONE counter serialise)
struct TopkState {
    unsigned long long prefix;     // digits of T fixed so far (high to low)
    unsigned long long remaining;  // rank of T among the images that share the prefix (1-based)
    unsigned long long hist[256];
    unsigned long long ncand;
    unsigned long long seg_count[kTopkSegs * 16];  // sampled path: candidates per segment, 128 B apart
};

__global__ void topk_images_kernel(const Program P, const OutValue* vals, uint32_t stride, uint32_t index, uint64_t n, uint32_t desc,
                                   uint64_t* images, TopkState* st, uint64_t keep) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        st->prefix = 0;
        st->remaining = keep;
        st->ncand = 0;
    }
    if (i < n) images[i] = order_image(P, vals[i * stride + index].tag, vals[i * stride + index].payload, desc != 0);
}

__global__ void topk_state_kernel(TopkState* st, uint64_t keep) {
    st->prefix = 0;
    st->remaining = keep;
    st->ncand = 0;
}

// pass d (0 = top byte): histogram of byte d over the images that share the prefix of the bytes above it
__global__ __launch_bounds__(256) void topk_hist_kernel(const uint64_t* images, uint64_t n, uint32_t pass, TopkState* st) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t shift = 56 - 8 * pass;
    const unsigned long long prefix = st->prefix;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const uint64_t x = images[i];
        if (pass == 0 || (x >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&h[(x >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&st->hist[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

// one wave: lane l owns bins 4l .. 4l + 3 (a serial walk over 256 dependent global loads took 15 us per pass, eight passes
// per query)
__global__ void topk_pick_kernel(uint32_t pass, TopkState* st) {
    if (blockIdx.x != 0 || threadIdx.x >= 64) return;
    const uint32_t lane = threadIdx.x;
    const uint32_t shift = 56 - 8 * pass;
    const unsigned long long rem = st->remaining;
    unsigned long long c[4], mine = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        c[k] = st->hist[lane * 4 + k];
        mine += c[k];
    }
    unsigned long long incl = mine;  // bins before and including this lane's
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long t = __shfl_up(incl, off, 64);
        if ((int)lane >= off) incl += t;
    }
    unsigned long long run = incl - mine;  // images in the bins before this lane's
    // the first bin b with (images in bins <= b) >= rem; none (rem beyond the total): bin 255, as the serial walk did
    uint32_t d = 256;
    unsigned long long before = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (d == 256 && run + c[k] >= rem) {
            d = lane * 4 + k;
            before = run;
        }
        run += c[k];
    }
    const unsigned long long found = __ballot(d != 256);
    const int src = found ? __ffsll((long long)found) - 1 : 63;
    uint32_t dd = (uint32_t)__shfl((int)d, src, 64);
    unsigned long long bb = __shfl(before, src, 64);
    const unsigned long long total = __shfl(incl, 63, 64);
    if (!found) {  // (rem beyond the total: what the serial walk left behind)
        dd = 255;
        bb = total;
    }
    if (lane == 0) {
        st->prefix |= (unsigned long long)dd << shift;
        st->remaining = rem - bb;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) st->hist[lane * 4 + k] = 0;
}

// ORDER BY ... LIMIT over millions of groups, the cheap way to the threshold.  Three small kernels instead of eight
// (histogram + pick) passes over all images (0.2 ms per 6.4 M groups):
//   1. topk_sample_kernel: the images of a SAMPLE of the groups (one workgroup, kTopkSample images in LDS), the r-th smallest
//      of them by a radix select inside LDS = a first threshold T'; r is chosen so that about 2 x keep + 16 n / kTopkSample
//      groups are expected at or below T';
//   2. topk_gather_seg_kernel: the groups with image <= T' into kTopkSegs lists (segment = workgroup number mod kTopkSegs, its
//      own counter: thousands of candidates on ONE counter serialise at ~ 12 ns each);
//   3. topk_refine_kernel: the lists packed into cand[]; when at least `keep` and at most kTopkSample groups came through, the
//      EXACT threshold T = the keep-th smallest image among them (the same radix select, in LDS) and only the groups with
//      image <= T stay — the candidate set of the exact path: the first `keep` rows of the order and every tie on the first term.
// Fewer than `keep` candidates (the sample was unlucky: T' < T): the host sees it in ncand and runs the exact radix select.
constexpr uint32_t kTopkSample = 16384;

// the rank-th smallest (1-based) of the n <= kTopkSample images in LDS; all 1024 threads of the workgroup call it.  Lanes of a
// wave that count the same bin add once (early passes put every image in one bin: 16 k same-address LDS atomics were 27 us).
// `passes` < 8: only the high bytes are selected and the bytes below are all ones — an UPPER bound of the rank-th image (what the
// sample needs: a few more candidates, half the passes).
N1K_DEV uint64_t lds_radix_select(const uint64_t* smp, uint32_t n, unsigned long long rank, uint32_t* h, unsigned long long* s_prefix,
                                  unsigned long long* s_rem, uint32_t passes = 8) {
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    if (tid == 0) {
        *s_prefix = 0;
        *s_rem = rank;
    }
    __syncthreads();
    for (uint32_t pass = 0; pass < passes; pass++) {
        const uint32_t shift = 56 - 8 * pass;
        if (tid < 256) h[tid] = 0;
        __syncthreads();
        const unsigned long long prefix = *s_prefix;
        for (uint32_t i0 = 0; i0 < n; i0 += 1024) {
            const uint32_t i = i0 + tid;
            const uint64_t x = i < n ? smp[i] : 0;
            const bool in = i < n && (pass == 0 || (x >> (shift + 8)) == (prefix >> (shift + 8)));
            const uint32_t bin = (uint32_t)(x >> shift) & 255u;
            // lanes with one digit count themselves with ONE atomic (the high digits of images are all alike: thousands of
            // atomics on one counter); what is left after the leader's digit is spread out and goes in lane by lane (a wave whose
            // 64 digits all differ would otherwise spend 64 rounds here)
            unsigned long long todo = __ballot(in);
            for (int round = 0; todo && round < 1; round++) {
                const int leader = __ffsll((long long)todo) - 1;
                const uint32_t lb = (uint32_t)__shfl((int)bin, leader, 64);
                const unsigned long long same = __ballot(in && bin == lb) & todo;
                if ((int)lane == leader) atomicAdd(&h[lb], (uint32_t)__popcll(same));
                todo &= ~same;
            }
            if (todo >> lane & 1ull) atomicAdd(&h[bin], 1u);
        }
        __syncthreads();
        if (tid < 64) {  // one wave: lane l owns bins 4l .. 4l + 3 (as topk_pick_kernel)
            const unsigned long long rem = *s_rem;
            unsigned long long c[4], mine = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                c[k] = h[tid * 4 + k];
                mine += c[k];
            }
            unsigned long long incl = mine;
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned long long t = __shfl_up(incl, off, 64);
                if ((int)tid >= off) incl += t;
            }
            unsigned long long run = incl - mine, before = 0;
            uint32_t d = 256;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (d == 256 && run + c[k] >= rem) {
                    d = tid * 4 + k;
                    before = run;
                }
                run += c[k];
            }
            const unsigned long long found = __ballot(d != 256);
            const int src = found ? __ffsll((long long)found) - 1 : 63;
            uint32_t dd = (uint32_t)__shfl((int)d, src, 64);
            unsigned long long bb = __shfl(before, src, 64);
            if (!found) {
                dd = 255;
                bb = __shfl(incl, 63, 64);
            }
            if (tid == 0) {
                *s_prefix = prefix | ((unsigned long long)dd << shift);
                *s_rem = rem - bb;
            }
        }
        __syncthreads();
    }
    return passes < 8 ? (*s_prefix | ((1ull << (64 - 8 * passes)) - 1ull)) : *s_prefix;
}

__global__ __launch_bounds__(1024) void topk_sample_kernel(const uint64_t* images, uint64_t n, uint32_t rank, TopkState* st) {
    extern __shared__ uint64_t smp[];  // kTopkSample images
    __shared__ uint32_t h[256];
    __shared__ unsigned long long s_prefix, s_rem;
    const uint32_t tid = threadIdx.x;
    const uint32_t stride = (uint32_t)(n / kTopkSample);  // (the caller made sure that 4 * kTopkSample <= n < 2^32)
    uint64_t mine[kTopkSample / 1024];  // (all loads first: one memory latency, not one per image)
#pragma unroll
    for (uint32_t k = 0; k < kTopkSample / 1024; k++) {
        const uint32_t i = k * 1024 + tid;
        const uint32_t off = (uint32_t)mix64(0x5EEDull + i) % stride;  // one image out of every `stride`, at a scattered place
        mine[k] = images[(uint64_t)i * stride + off];
    }
#pragma unroll
    for (uint32_t k = 0; k < kTopkSample / 1024; k++) smp[k * 1024 + tid] = mine[k];
    __syncthreads();
    const uint64_t t = lds_radix_select(smp, kTopkSample, rank, h, &s_prefix, &s_rem, 4);  // (32 bits of the image: 20 of a number's mantissa)
    if (tid == 0) {
        st->prefix = t;
        st->ncand = 0;
    }
}

__global__ void topk_gather_seg_kernel(const uint64_t* images, uint64_t n, TopkState* st, uint32_t* seg_lists, uint64_t seg_cap) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (images[i] <= st->prefix) {
        const uint32_t sg = blockIdx.x % kTopkSegs;
        const unsigned long long at = atomicAdd(&st->seg_count[sg * 16], 1ull);
        if (at < seg_cap) seg_lists[(size_t)sg * seg_cap + at] = (uint32_t)i;
    }
}

__global__ __launch_bounds__(1024) void topk_refine_kernel(const uint64_t* images, TopkState* st, const uint32_t* seg_lists, uint64_t seg_cap,
                                                          uint32_t* cand, uint64_t keep) {
    extern __shared__ uint64_t smp[];  // up to kTopkSample candidate images
    __shared__ uint32_t h[256];
    __shared__ unsigned long long s_prefix, s_rem;
    __shared__ uint32_t off[kTopkSegs + 1], kept;
    const uint32_t tid = threadIdx.x;
    // the lists' lengths -> offsets (one wave, four segments per lane)
    if (tid < 64) {
        uint32_t c[4], mine = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const unsigned long long x = st->seg_count[(tid * 4 + k) * 16];
            c[k] = (uint32_t)(x < seg_cap ? x : seg_cap);
            mine += c[k];
        }
        uint32_t incl = mine;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(incl, o, 64);
            if ((int)tid >= o) incl += t;
        }
        uint32_t run = incl - mine;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            off[tid * 4 + k] = run;
            run += c[k];
        }
        if (tid == 63) off[kTopkSegs] = run;
    }
    if (tid == 0) kept = 0;
    __syncthreads();
    const uint32_t M = off[kTopkSegs];
    const bool refine = M >= keep && M <= kTopkSample;
    // pack: candidate i of all lies in the segment whose offsets bracket i (binary search in LDS) — every thread takes every
    // 1024-th candidate, so that the two dependent loads per candidate (its group, then the group's image) overlap across
    // candidates instead of queueing up behind each other in four threads per segment
    for (uint32_t i = tid; i < M; i += 1024) {
        uint32_t lo = 0, hi = kTopkSegs;  // off[lo] <= i < off[hi]
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (off[mid] <= i) lo = mid; else hi = mid;
        }
        const uint32_t g = seg_lists[(size_t)lo * seg_cap + (i - off[lo])];
        cand[i] = g;
        if (refine) smp[i] = images[g];
    }
    __syncthreads();
    if (!refine) {
        if (tid == 0) st->ncand = M;
        return;
    }
    __syncthreads();
    const uint64_t t = lds_radix_select(smp, M, keep, h, &s_prefix, &s_rem);
    // the groups at or below the exact threshold stay (cand is rewritten front to back: reads of a round precede its writes)
    for (uint32_t i0 = 0; i0 < M; i0 += 1024) {
        const uint32_t i = i0 + tid;
        const bool keepit = i < M && smp[i] <= t;
        const uint32_t g = i < M ? cand[i] : 0u;
        __syncthreads();
        if (keepit) cand[atomicAdd(&kept, 1u)] = g;
        __syncthreads();
    }
    if (tid == 0) {
        st->prefix = t;
        st->ncand = kept;
    }
}

__global__ void topk_gather_kernel(const uint64_t* images, uint64_t n, TopkState* st, uint32_t* cand) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (images[i] <= st->prefix) cand[atomicAdd(&st->ncand, 1ull)] = (uint32_t)i;
}

// compact the candidates' records: output arrays laid out for `ncand` groups (same order of arrays as finalize's)
__global__ void topk_compact_kernel(const uint32_t* cand, uint64_t ncand, uint32_t nk, uint32_t na, const OutValue* keys,
                                    const OutValue* aggs, const OutPartial* parts, const uint64_t* rep, OutValue* okeys,
                                    OutValue* oaggs, OutPartial* oparts, uint64_t* orep) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ncand) return;
    const uint64_t g = cand[j];
    for (uint32_t k = 0; k < nk; k++) okeys[j * nk + k] = keys[g * nk + k];
    for (uint32_t a = 0; a < na; a++) {
        oaggs[j * na + a] = aggs[g * na + a];
        oparts[j * na + a] = parts[g * na + a];
    }
    orep[j] = rep[g];
}

// ------------------------------------------------------------------ Filter alone: mask, scan, compaction
//
// Filter.processItem forwards the rows whose condition is TRUE, in input order (execution/filter.go:49-61).
// K1: every wave evaluates 64 x R consecutive rows and stores one ballot word per 64 rows (1 bit/row) plus a
//     survivor count per tile of kFilterTile rows.  K2 (after an exclusive scan of the tile counts): LDS-staged,
//     ordered stream compaction of the set bits into row ordinals.
template <int R, int BLOCK>
__global__ __launch_bounds__(BLOCK) void filter_mask_kernel(const Program P, uint64_t nrows, uint64_t* mask_words,
                                                           uint32_t* tile_counts, uint32_t* err_flags) {
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t unsupported = 0;
    const uint64_t chunk_rows = (uint64_t)BLOCK * R;
    const uint64_t nchunks = (nrows + chunk_rows - 1) / chunk_rows;
    for (uint64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const uint64_t wave_base = chunk * chunk_rows + (uint64_t)wave * 64 * R;
        uint64_t row[R];
        bool valid[R], pass[R];
#pragma unroll
        for (int j = 0; j < R; j++) {
            row[j] = wave_base + (uint64_t)j * 64 + lane;
            valid[j] = row[j] < nrows;
        }
        eval_predicate<R>(P, row, valid, pass, unsupported);
        uint32_t cnt = 0;
#pragma unroll
        for (int j = 0; j < R; j++) {
            unsigned long long b = __ballot(pass[j]);
            cnt += (uint32_t)__popcll(b);
            if (lane == 0 && wave_base + (uint64_t)j * 64 < nrows) mask_words[(wave_base >> 6) + j] = b;
        }
        if (lane == 0 && cnt) atomicAdd(&tile_counts[wave_base / kFilterTile], cnt);  // 64*R divides kFilterTile
    }
    if (unsupported) atomicOr(err_flags, (uint32_t)ERR_UNSUPPORTED_VALUE);
}

// K2 in ONE pass (execution/filter.go:49-61: the rows whose condition is TRUE, in input order): predicate, survivor count and
// ordered compaction of a tile by one workgroup, the tiles' output offsets by a chained scan with decoupled look-back — no
// mask array, no single-workgroup scan between two kernels, no host round trip before the ordinals are written.
//   * tiles (8192 rows) are handed out in order by one counter: a workgroup only ever waits for tiles that were handed out
//     before its own;
//   * tile t publishes ONE 64-bit word: [flag:2][count:62], flag 1 = the tile's own count (aggregate), 2 = the count of
//     tiles 0..t (inclusive prefix).  The word is the whole message — relaxed agent-scope store / load, nothing to order;
//   * wave 0 looks back 64 tiles at a time: it adds aggregates until it meets an inclusive prefix;
//   * survivors leave in row order: lane-private ranks from ballots, per-(j, wave) counts in LDS.
// Algorithmic traffic: the predicate's columns once + 8 B per survivor.
constexpr unsigned long long kTileAgg = 1ull << 62, kTilePrefix = 2ull << 62, kTileValue = (1ull << 62) - 1ull;

// Tried and measured (100 M rows, 0.38 ms as it stands): all of a tile's loads issued at once — 179 registers, two workgroups
// per CU: 0.74 ms; in two halves at 128 registers: 0.49 ms; the NEXT tile's ticket drawn while this one is worked on (to hide
// the returning atomic): twice as slow — a tile drawn early publishes late, and every higher tile's look-back waits for it.
// The kernel wants many small workgroups: the chain of look-backs, not the loads, is what a tile waits for.
// A tile = SUB passes of BLOCK x R rows: the predicate runs pass by pass (R rows per thread in registers at a time), what it
// leaves per pass is one bit per row and the per-(pass, j, wave) survivor counts in LDS; then ONE look-back for the tile and
// the survivors' ordinals pass by pass.  (Tiles of 1024 rows with one ticket each were 97 k same-address atomics per 100 M
// rows — at ~ 12 ns each they alone took 1.2 ms; several tiles per ticket serialise: a tile's aggregate must be out before
// its workgroup looks back.)
// FAST: the condition is ONE comparison of a TAGGED64 column with a NUMBER constant (config 2's `price > 50`): the column is
// read as the scan kernels read it — two adjacent rows per lane and load (16 B of payload, 2 B of tags, nontemporal: each byte
// is read once) — instead of through the interpreter's per-term loads (9 B per row in 1- and 8-byte pieces: 0.51 ms per
// 100 M rows, 32 % of the HBM peak).  Rows keep their order: item i holds rows 2 i and 2 i + 1.
struct FilterFast {
    const uint8_t* tags;      // the column (an even number of rows in front of it: 2-byte tag loads, 16-byte payload loads)
    const uint64_t* payload;
    uint32_t op, ctag;        // TERM_NUM_*; the constant
    uint64_t cpayload;
};

N1K_DEV bool num_term_true(uint32_t op, uint32_t tg, uint64_t p, uint32_t ct, uint64_t cp, double cf) {
    if (tg <= T_NULL) return false;  // MISSING / NULL: the comparison is not TRUE
    int c;                           // collation of the value against the constant (eval_term, TERM_NUM_*)
    if (tg == T_INT && ct == T_INT) {
        const int64_t x = (int64_t)p, y = (int64_t)cp;
        c = x < y ? -1 : (x > y ? 1 : 0);
    } else if (tg == T_INT || tg == T_FLOAT)
        c = collate_f64(num_actual(tg, p), cf);
    else
        c = tg < T_INT ? -1 : 1;  // BOOLEAN sorts below NUMBER, STRING / ARRAY / OBJECT above
    return op == TERM_NUM_LT ? c < 0 : op == TERM_NUM_LE ? c <= 0 : op == TERM_NUM_GT ? c > 0 : op == TERM_NUM_GE ? c >= 0
           : (c == 0 && (tg == T_INT || tg == T_FLOAT));
}

template <int R, int BLOCK, int SUB, bool FAST>
__global__ __launch_bounds__(BLOCK) void filter_stream_kernel(const Program P, uint64_t nrows, uint64_t row_base, uint64_t* out_rows,
                                                             unsigned long long* tile_state, unsigned long long* tile_counter,
                                                             unsigned long long* total, uint32_t* err_flags, const FilterFast F) {
    constexpr int NW = BLOCK / 64;
    constexpr int RJ = FAST ? R / 2 : R;  // items per thread and pass (the same rows per pass either way)
    __shared__ uint32_t wcnt[SUB][RJ][NW];
    __shared__ unsigned long long s_tile, s_prefix;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t pass_rows = (uint64_t)BLOCK * R, tile_rows = pass_rows * SUB, ntiles = (nrows + tile_rows - 1) / tile_rows;
    const double cf = FAST ? num_actual(F.ctag, F.cpayload) : 0.0;
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t unsupported = 0;
    for (;;) {
        if (tid == 0) s_tile = atomicAdd(tile_counter, 1ull);
        __syncthreads();
        const uint64_t tile = s_tile;
        if (tile >= ntiles) break;  // (uniform: every thread reads the same LDS word)
        uint32_t bits[SUB];  // this thread's rows of pass u that passed (bit j * H + h)
#pragma unroll
        for (int u = 0; u < SUB; u++) {
            bits[u] = 0;
            if constexpr (FAST) {
                typedef unsigned long long n1k_u64x2 __attribute__((ext_vector_type(2)));
                n1k_u64x2 pp[RJ];
                uint32_t tt[RJ];
                uint64_t first[RJ];
#pragma unroll
                for (int j = 0; j < RJ; j++) {  // all loads of the pass first
                    first[j] = tile * tile_rows + (uint64_t)u * pass_rows + ((uint64_t)j * BLOCK + tid) * 2;
                    pp[j] = n1k_u64x2{0ull, 0ull};
                    tt[j] = 0;
                    if (first[j] + 1 < nrows) {
                        pp[j] = __builtin_nontemporal_load((const n1k_u64x2*)(F.payload + first[j]));
                        tt[j] = (uint32_t)__builtin_nontemporal_load((const uint16_t*)(F.tags + first[j]));
                    } else if (first[j] < nrows) {  // the last row of an odd count
                        pp[j].x = F.payload[first[j]];
                        tt[j] = F.tags[first[j]];
                    }
                }
#pragma unroll
                for (int j = 0; j < RJ; j++) {
                    const bool p0 = first[j] < nrows && num_term_true(F.op, tt[j] & 255u, pp[j].x, F.ctag, F.cpayload, cf);
                    const bool p1 = first[j] + 1 < nrows && num_term_true(F.op, tt[j] >> 8, pp[j].y, F.ctag, F.cpayload, cf);
                    const unsigned long long m0 = __ballot(p0), m1 = __ballot(p1);
                    bits[u] |= (p0 ? 1u : 0u) << (2 * j) | (p1 ? 2u : 0u) << (2 * j);
                    if (lane == 0) wcnt[u][j][wave] = (uint32_t)(__popcll(m0) + __popcll(m1));
                }
            } else {
                uint64_t row[R];
                bool valid[R], pass[R];
#pragma unroll
                for (int j = 0; j < R; j++) {
                    row[j] = tile * tile_rows + (uint64_t)u * pass_rows + (uint64_t)j * BLOCK + tid;
                    valid[j] = row[j] < nrows;
                }
                eval_predicate<R>(P, row, valid, pass, unsupported);
#pragma unroll
                for (int j = 0; j < R; j++) {
                    const unsigned long long m = __ballot(pass[j]);
                    if (pass[j]) bits[u] |= 1u << j;
                    if (lane == 0) wcnt[u][j][wave] = (uint32_t)__popcll(m);
                }
            }
        }
        __syncthreads();
        if (wave == 0) {
            // the tile's survivors: SUB x RJ x NW counts, summed by the wave
            uint32_t c = 0;
            for (uint32_t i = lane; i < (uint32_t)(SUB * RJ * NW); i += 64) c += (&wcnt[0][0][0])[i];
            for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
            const uint32_t block_total = __shfl(c, 0, 64);
            unsigned long long excl = 0;
            if (tile > 0) {
                if (lane == 0) __hip_atomic_store(&tile_state[tile], kTileAgg | block_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int64_t hi = (int64_t)tile - 1; hi >= 0; hi -= 64) {  // the window hi, hi - 1, ... hi - 63
                    const int64_t t = hi - (int64_t)lane;
                    unsigned long long v = kTilePrefix;  // (in front of tile 0: an empty prefix)
                    if (t >= 0) {
                        do {
                            v = __hip_atomic_load(&tile_state[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        } while ((v >> 62) == 0);  // handed out before this tile: it will publish
                    }
                    const unsigned long long pm = __ballot((v >> 62) == 2);
                    const int stop = pm ? __ffsll((long long)pm) - 1 : 63;  // nearest tile with an inclusive prefix
                    unsigned long long add = (int)lane <= stop ? (v & kTileValue) : 0ull;
                    for (int off = 32; off > 0; off >>= 1) add += __shfl_down(add, off, 64);
                    excl += __shfl(add, 0, 64);
                    if (pm) break;
                }
            }
            if (lane == 0) {
                __hip_atomic_store(&tile_state[tile], kTilePrefix | (excl + block_total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_prefix = excl;
                if (tile == ntiles - 1) *total = excl + block_total;
            }
        }
        __syncthreads();
        unsigned long long at = s_prefix;  // first output position of (pass u, item j, wave 0)
#pragma unroll
        for (int u = 0; u < SUB; u++) {
#pragma unroll
            for (int j = 0; j < RJ; j++) {
                uint32_t before = 0, all = 0;
#pragma unroll
                for (int w = 0; w < NW; w++) {
                    if (w < (int)wave) before += wcnt[u][j][w];
                    all += wcnt[u][j][w];
                }
                if constexpr (FAST) {
                    const bool p0 = (bits[u] >> (2 * j)) & 1u, p1 = (bits[u] >> (2 * j + 1)) & 1u;
                    const unsigned long long m0 = __ballot(p0), m1 = __ballot(p1);
                    const uint64_t r0 = tile * tile_rows + (uint64_t)u * pass_rows + ((uint64_t)j * BLOCK + tid) * 2;
                    const uint32_t mine = (uint32_t)(__popcll(m0 & lt) + __popcll(m1 & lt));  // survivors of the lanes below, both rows
                    if (p0) out_rows[at + before + mine] = row_base + r0;
                    if (p1) out_rows[at + before + mine + (p0 ? 1u : 0u)] = row_base + r0 + 1;
                } else {
                    const bool p = (bits[u] >> j) & 1u;
                    const unsigned long long m = __ballot(p);
                    if (p) out_rows[at + before + (uint32_t)__popcll(m & lt)] =
                               row_base + tile * tile_rows + (uint64_t)u * pass_rows + (uint64_t)j * BLOCK + tid;
                }
                at += all;
            }
        }
        __syncthreads();  // (wcnt / s_prefix / s_tile are written again by the next tile)
    }
    if (unsupported) atomicOr(err_flags, (uint32_t)ERR_UNSUPPORTED_VALUE);
}

hipError_t launch_filter_stream(const Program& P, uint64_t nrows, uint64_t row_base, uint64_t* out_rows, unsigned long long* tile_state,
                                unsigned long long* tile_counter, unsigned long long* total, uint32_t* err_flags, uint32_t grid,
                                hipStream_t st, bool fast) {
    const uint64_t ntiles = (nrows + kFilterStreamTile - 1) / kFilterStreamTile;
    hipError_t e = hipMemsetAsync(tile_state, 0, (size_t)(ntiles + 1) * sizeof(unsigned long long), st);  // (+1: the tile counter behind it)
    if (e != hipSuccess) return e;
    static_assert(kFilterStreamTile == 256 * 8 * 4, "tile = 4 passes of 256 threads x 8 rows");
    FilterFast F{};
    if (fast) {  // (the caller checked: one TERM_NUM_* term over a TAGGED64 column, payload 16-byte and tags 2-byte aligned)
        const Term& t = P.terms[P.logic[0].arg];
        F.tags = P.cols[t.a.col].tags;
        F.payload = P.cols[t.a.col].payload;
        F.op = t.op;
        F.ctag = t.b.ctag;
        F.cpayload = t.b.cpayload;
        hipLaunchKernelGGL((filter_stream_kernel<8, 256, 4, true>), dim3(grid), dim3(256), 0, st, P, nrows, row_base, out_rows, tile_state,
                           tile_counter, total, err_flags, F);
    } else
        hipLaunchKernelGGL((filter_stream_kernel<8, 256, 4, false>), dim3(grid), dim3(256), 0, st, P, nrows, row_base, out_rows, tile_state,
                           tile_counter, total, err_flags, F);
    return hipGetLastError();
}

// exclusive scan of the tile counts (single workgroup, looping) -> tile offsets + total
__global__ __launch_bounds__(1024) void tile_scan_kernel(const uint32_t* counts, uint64_t* offsets, uint64_t ntiles,
                                                        unsigned long long* total) {
    __shared__ uint64_t part[1024];
    __shared__ uint64_t carry;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (uint64_t base = 0; base < ntiles; base += 1024) {
        uint64_t i = base + tid;
        uint64_t v = i < ntiles ? counts[i] : 0;
        part[tid] = v;
        __syncthreads();
        for (uint32_t off = 1; off < 1024; off <<= 1) {
            uint64_t add = tid >= off ? part[tid - off] : 0;
            __syncthreads();
            part[tid] += add;
            __syncthreads();
        }
        if (i < ntiles) offsets[i] = carry + part[tid] - v;
        __syncthreads();
        if (tid == 1023) carry += part[1023];
        __syncthreads();
    }
    if (tid == 0) *total = carry;
}

// K2: one workgroup per tile of kFilterTile rows (64 mask words): word offsets through LDS, then every lane
// writes the ordinal of its set bit at its rank -> ascending, densely packed output
__global__ __launch_bounds__(256) void filter_compact_kernel(const uint64_t* mask_words, const uint64_t* tile_offsets,
                                                            uint64_t nrows, uint64_t row_base, uint64_t* out_rows) {
    __shared__ uint32_t word_off[kFilterTile / 64];
    __shared__ uint64_t words[kFilterTile / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t ntiles = (nrows + kFilterTile - 1) / kFilterTile;
    constexpr uint32_t kWords = kFilterTile / 64;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint64_t first_word = tile * kWords;
        if (tid < kWords) {
            uint64_t w = (first_word + tid) * 64 < nrows ? mask_words[first_word + tid] : 0ull;
            words[tid] = w;
            // exclusive scan of the 64 popcounts inside wave 0
            uint32_t c = (uint32_t)__popcll(w), incl = c;
            for (int off = 1; off < 64; off <<= 1) {
                uint32_t t = __shfl_up(incl, off, 64);
                if ((int)lane >= off) incl += t;
            }
            word_off[tid] = incl - c;
        }
        __syncthreads();
        const uint64_t out_base = tile_offsets[tile];
        for (uint32_t w = wave; w < kWords; w += 256 / 64) {
            uint64_t m = words[w];
            if ((m >> lane) & 1ull) {
                uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                out_rows[out_base + word_off[w] + rank] = row_base + (first_word + w) * 64 + lane;
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ synthetic columns

N1K_DEV uint64_t splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void synth_kernel(SynthArgs a) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= a.nrows) return;
    uint64_t i = a.first_row + j;
    uint64_t base = a.seed + i * 8ull;
    if (a.cat_codes) {
        uint64_t r = splitmix64(base + 0);
        uint32_t c;
        if (a.cat_cdf) {
            double u = (double)(r >> 11) * 0x1.0p-53;
            uint32_t lo = 0, hi = a.k_cat;
            while (lo < hi) {
                uint32_t mid = lo + ((hi - lo) >> 1);
                if (a.cat_cdf[mid] > u) hi = mid; else lo = mid + 1;
            }
            c = lo < a.k_cat ? lo : a.k_cat - 1;
        } else {
            c = (uint32_t)__umul64hi(r, (uint64_t)a.k_cat);
        }
        a.cat_codes[j] = c;
    }
    if (a.price_tags || a.price_payload) {
        uint64_t sel = splitmix64(base + 1) % 1000ull;
        uint64_t r2 = splitmix64(base + 2);
        uint8_t tag;
        uint64_t pay = 0;
        if (sel < 800) {
            uint64_t cents = r2 % 10000ull;
            if (cents % 100ull == 0) {
                tag = T_INT;
                pay = cents / 100ull;
            } else {
                tag = T_FLOAT;
                pay = f64_bits((double)cents / 100.0);
            }
        } else if (sel < 980) {
            tag = T_INT;
            pay = r2 % 101ull;
        } else if (sel < 990) {
            tag = T_NULL;
        } else if (sel < 995) {
            tag = T_MISSING;
        } else {
            tag = T_STRING;
            pay = a.k_cat;
        }
        if (a.price_tags) a.price_tags[j] = tag;
        if (a.price_payload) a.price_payload[j] = pay;
    }
    uint64_t urange = a.total_rows / 10;
    if (urange == 0) urange = 1;
    if (a.user_tags) a.user_tags[j] = T_INT;
    if (a.user_payload) a.user_payload[j] = splitmix64(base + 3) % urange;
    if (a.region_tags) a.region_tags[j] = T_INT;
    if (a.region_payload) a.region_payload[j] = splitmix64(base + 4) % 64ull;
}

// ------------------------------------------------------------------ launchers (called by the host engine)

hipError_t launch_init_table(const Program& P, const GlobalTable& G, uint64_t first, uint64_t count,
                             unsigned long long* counters, hipStream_t st) {
    if (count == 0) return hipSuccess;
    uint32_t blocks = (uint32_t)((count + 255) / 256);
    hipLaunchKernelGGL(init_table_kernel, dim3(blocks), dim3(256), 0, st, P, G, first, count, counters);
    return hipGetLastError();
}

hipError_t launch_rehash(const Program& P, const GlobalTable& oldt, const GlobalTable& newt, uint32_t* err_flags,
                         unsigned long long* ngroups_scratch, hipStream_t st) {
    uint32_t blocks = (uint32_t)((oldt.capacity + 255) / 256);
    hipLaunchKernelGGL(rehash_kernel, dim3(blocks), dim3(256), 0, st, P, oldt, newt, err_flags, ngroups_scratch);
    return hipGetLastError();
}

template <int R, int BLOCK>
static hipError_t launch_scan_variant(const Program& P, const ScanArgs& A, const GlobalTable& G, unsigned long long* ngroups,
                                      uint32_t grid, bool direct, size_t shmem, hipStream_t st) {
    if (direct) {
        auto k = scan_group_kernel<R, BLOCK, true>;
        if (shmem > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        hipLaunchKernelGGL(k, dim3(grid), dim3(BLOCK), shmem, st, P, A, G, ngroups);
    } else {
        auto k = scan_group_kernel<R, BLOCK, false>;
        if (shmem > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        hipLaunchKernelGGL(k, dim3(grid), dim3(BLOCK), shmem, st, P, A, G, ngroups);
    }
    return hipGetLastError();
}

hipError_t launch_scan_group(const Program& P, const ScanArgs& A, const GlobalTable& G, unsigned long long* ngroups,
                             uint32_t grid, uint32_t block, uint32_t rows_per_lane, bool direct, hipStream_t st) {
    size_t shmem = (size_t)A.lds_slots * P.lds_words * 8 + (size_t)A.dcache_slots * A.dcache_aggs * 8;
    if (block == 256) return launch_scan_variant<4, 256>(P, A, G, ngroups, grid, direct, shmem, st);
    if (block == 512) return launch_scan_variant<4, 512>(P, A, G, ngroups, grid, direct, shmem, st);
    if (rows_per_lane == 2) return launch_scan_variant<2, 1024>(P, A, G, ngroups, grid, direct, shmem, st);
    return launch_scan_variant<4, 1024>(P, A, G, ngroups, grid, direct, shmem, st);
}

template <int R, int BLOCK>
static hipError_t launch_fast_variant(const Program& P, const FastArgs& F, const GlobalTable& G, unsigned long long* ngroups,
                                      uint32_t grid, size_t shmem, hipStream_t st) {
    auto k = scan_fast_kernel<R, BLOCK>;
    if (shmem > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    hipLaunchKernelGGL(k, dim3(grid), dim3(BLOCK), shmem, st, P, F, G, ngroups);
    return hipGetLastError();
}

hipError_t launch_scan_fast(const Program& P, const FastArgs& F, const GlobalTable& G, unsigned long long* ngroups,
                            uint32_t grid, uint32_t block, uint32_t rows_per_lane, hipStream_t st) {
    size_t shmem = (size_t)F.lds_slots * P.lds_words * 8;
    if (block == 512) {
        if (rows_per_lane == 2) return launch_fast_variant<2, 512>(P, F, G, ngroups, grid, shmem, st);
        return launch_fast_variant<4, 512>(P, F, G, ngroups, grid, shmem, st);
    }
    if (rows_per_lane == 2) return launch_fast_variant<2, 1024>(P, F, G, ngroups, grid, shmem, st);
    return launch_fast_variant<4, 1024>(P, F, G, ngroups, grid, shmem, st);
}

hipError_t launch_merge_slabs(const Program& P, const FastArgs& F, const GlobalTable& G, uint32_t nblocks,
                              unsigned long long* ngroups, hipStream_t st, uint32_t ychunks_opt, const TailArgs* tail) {
    TailArgs T{};
    if (tail && G.capacity <= kFinalizeSmallMax) T = *tail;
    uint32_t blocks = (F.lds_slots + 63) / 64;
    // 16 * ychunks parallel chunks of workgroups; enough blocks for every CU (64 blocks read 25 MB of slabs at 1.25 TB/s)
    uint32_t ychunks = nblocks >= 512 ? 4 : (nblocks >= 128 ? 2 : 1);
    while (ychunks < 16 && blocks * ychunks < 256 && nblocks >= 32 * ychunks) ychunks *= 2;
    if (ychunks_opt) ychunks = ychunks_opt;
    hipLaunchKernelGGL(merge_slabs_kernel, dim3(blocks, ychunks), dim3(64, 16), 0, st, P, F, G, nblocks, ngroups, T);
    return hipGetLastError();
}

__global__ void add_counter_kernel(unsigned long long* p, unsigned long long v) { *p += v; }

hipError_t launch_add_counter(unsigned long long* p, unsigned long long v, hipStream_t st) {
    hipLaunchKernelGGL(add_counter_kernel, dim3(1), dim3(1), 0, st, p, v);
    return hipGetLastError();
}

hipError_t launch_finalize_region(const Program& P, const uint64_t* region, uint64_t cap, uint64_t count, OutValue* out_keys,
                                  OutValue* out_aggs, OutPartial* out_parts, uint64_t* out_rep, uint32_t* err_flags, hipStream_t st,
                                  const uint32_t* cand, OutValue* ord, bool ord_is_key, uint32_t ord_index, uint64_t* images, bool desc) {
    if (!count) return hipSuccess;
    hipLaunchKernelGGL(finalize_region_kernel, dim3((uint32_t)((count + 255) / 256)), dim3(256), 0, st, P, region, cap, count,
                       out_keys, out_aggs, out_parts, out_rep, err_flags, cand, ord, ord_is_key ? 1u : 0u, ord_index, images, desc ? 1u : 0u);
    return hipGetLastError();
}

hipError_t launch_probe_keys(const Program& P, uint64_t nrows, const GlobalTable& G, uint32_t* err_flags, unsigned long long* ngroups,
                             uint32_t grid, hipStream_t st) {
    hipLaunchKernelGGL((probe_keys_kernel<4, 256>), dim3(grid), dim3(256), 0, st, P, nrows, G, err_flags, ngroups);
    return hipGetLastError();
}

hipError_t launch_project_records(const Program& P, const ProjectArgs& A, uint32_t grid, hipStream_t st) {
    hipLaunchKernelGGL((project_records_kernel<4, 512>), dim3(grid), dim3(512), 0, st, P, A);
    return hipGetLastError();
}

hipError_t launch_agg_bins(const Program& P, const BinAggArgs& A, const GlobalTable& G, unsigned long long* ngroups, uint32_t grid,
                           hipStream_t st) {
    auto k = agg_bins_kernel<512>;
    size_t shmem = (size_t)A.lds_slots * P.lds_words * 8;
    if (shmem > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), shmem, st, P, A, G, ngroups);
    return hipGetLastError();
}

size_t topk_state_bytes() { return sizeof(TopkState); }

uint64_t topk_cand_entries(uint64_t n) { return n + (uint64_t)kTopkSegs * (n / kTopkSegs + 320); }  // the candidates + the segment lists
bool topk_can_sample(uint64_t n, uint64_t keep) { return n >= 4ull * kTopkSample && n < (1ull << 32) && keep * 2 * kTopkSample / n + 16 <= kTopkSample / 4; }

// sampled = true: the threshold from a sample (the caller checks that at least `keep` candidates came out, and calls again
// with sampled = false — the images are in place: images_done — when they did not)
hipError_t launch_topk_select(const Program& P, const OutValue* vals, uint32_t stride, uint32_t index, uint64_t n, bool desc,
                              uint64_t keep, uint64_t* images, void* state, uint32_t* cand, hipStream_t st, bool sampled,
                              bool images_done) {
    TopkState* S = (TopkState*)state;
    (void)hipMemsetAsync(S, 0, sizeof(TopkState), st);
    const uint32_t blocks = (uint32_t)((n + 255) / 256);
    if (!images_done)
        hipLaunchKernelGGL(topk_images_kernel, dim3(blocks), dim3(256), 0, st, P, vals, stride, index, n, desc ? 1u : 0u, images, S, keep);
    else
        hipLaunchKernelGGL(topk_state_kernel, dim3(1), dim3(1), 0, st, S, keep);
    if (sampled) {
        const uint32_t rank = (uint32_t)(keep * 2 * kTopkSample / n) + 16;  // ~ 2 keep + 16 n / kTopkSample groups expected below it
        auto k = topk_sample_kernel;
        (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kTopkSample * 8));
        hipLaunchKernelGGL(k, dim3(1), dim3(1024), kTopkSample * 8, st, images, n, rank, S);
        // (the segment lists live behind the n entries of `cand`: the caller sized it with topk_cand_entries)
        const uint64_t seg_cap = n / kTopkSegs + 320;  // every image of a segment's workgroups fits
        uint32_t* seg_lists = cand + n;
        hipLaunchKernelGGL(topk_gather_seg_kernel, dim3(blocks), dim3(256), 0, st, images, n, S, seg_lists, seg_cap);
        auto kr = topk_refine_kernel;
        (void)hipFuncSetAttribute((const void*)kr, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kTopkSample * 8));
        hipLaunchKernelGGL(kr, dim3(1), dim3(1024), kTopkSample * 8, st, images, S, seg_lists, seg_cap, cand, keep);
        return hipGetLastError();
    }
    const uint32_t hb = (uint32_t)std::min<uint64_t>(blocks, 1024);
    for (uint32_t pass = 0; pass < 8; pass++) {
        hipLaunchKernelGGL(topk_hist_kernel, dim3(hb), dim3(256), 0, st, images, n, pass, S);
        hipLaunchKernelGGL(topk_pick_kernel, dim3(1), dim3(64), 0, st, pass, S);
    }
    hipLaunchKernelGGL(topk_gather_kernel, dim3(blocks), dim3(256), 0, st, images, n, S, cand);
    return hipGetLastError();
}

size_t topk_ncand_offset() { return offsetof(TopkState, ncand); }

hipError_t launch_topk_compact(const uint32_t* cand, uint64_t ncand, uint32_t nk, uint32_t na, const OutValue* keys,
                               const OutValue* aggs, const OutPartial* parts, const uint64_t* rep, OutValue* okeys, OutValue* oaggs,
                               OutPartial* oparts, uint64_t* orep, hipStream_t st) {
    if (!ncand) return hipSuccess;
    hipLaunchKernelGGL(topk_compact_kernel, dim3((uint32_t)((ncand + 255) / 256)), dim3(256), 0, st, cand, ncand, nk, na, keys, aggs,
                       parts, rep, okeys, oaggs, oparts, orep);
    return hipGetLastError();
}

hipError_t launch_radix_pass(const RadixArgs& A, uint32_t slices, hipStream_t st, bool have_hist) {
    if (A.bin_cap) {  // bins of fixed capacity: cursors from zero, no histogram, no offsets
        (void)hipMemsetAsync(A.cursor, 0, (size_t)A.nseg * 256 * (A.cursor_stride ? A.cursor_stride : 1u) * sizeof(unsigned long long), st);
        hipLaunchKernelGGL(radix_scatter_kernel<false>, dim3(slices, A.nseg), dim3(kRadixBlock), 0, st, A);
        return hipGetLastError();
    }
    if (!have_hist) {  // (the producer of the words may have counted the digits already)
        (void)hipMemsetAsync(A.hist, 0, (size_t)A.nseg * 256 * sizeof(unsigned long long), st);
        hipLaunchKernelGGL(radix_hist_kernel, dim3(slices, A.nseg), dim3(kRadixBlock), 0, st, A);
    }
    hipLaunchKernelGGL(radix_offsets_kernel, dim3(A.nseg), dim3(256), 0, st, A);
    if (A.nextra) hipLaunchKernelGGL(radix_scatter_kernel<true>, dim3(slices, A.nseg), dim3(kRadixBlock), 0, st, A);
    else hipLaunchKernelGGL(radix_scatter_kernel<false>, dim3(slices, A.nseg), dim3(kRadixBlock), 0, st, A);
    return hipGetLastError();
}

// hash regions -> one contiguous word log (n1k_finish's exact path when a region overflowed, or for the forced
// partition depths of the tests): (sub-)region b's words go to dst[off[b] ..]
__global__ __launch_bounds__(256) void compact_regions_kernel(const uint64_t* region, uint64_t cap, const unsigned long long* count,
                                                              const uint64_t* off, uint64_t* dst) {
    const uint32_t b = blockIdx.y;
    const uint64_t n = count[b * kCursorStride] < cap ? count[b * kCursorStride] : cap;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
        dst[off[b] + i] = region[(size_t)b * cap + i];
}

hipError_t launch_compact_regions(const uint64_t* region, uint32_t nreg, uint64_t cap, const unsigned long long* count, const uint64_t* off,
                                  uint64_t* dst, hipStream_t st) {
    hipLaunchKernelGGL(compact_regions_kernel, dim3(8, nreg), dim3(256), 0, st, region, cap, count, off, dst);
    return hipGetLastError();
}

// hash regions grow with the rows pushed: the words of region b move to the same region of a wider layout
__global__ __launch_bounds__(256) void regrow_regions_kernel(const uint64_t* src, uint64_t src_cap, uint64_t* dst, uint64_t dst_cap,
                                                             const unsigned long long* count) {
    const uint32_t b = blockIdx.y;
    const uint64_t n = count[b * kCursorStride] < src_cap ? count[b * kCursorStride] : src_cap;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
        dst[(size_t)b * dst_cap + i] = src[(size_t)b * src_cap + i];
}

// a region that overflowed counted words it never took (they went to the plain log): its cursor goes back to the
// capacity it had, so that the wider region continues right behind the words that are really there
__global__ void clamp_cursors_kernel(unsigned long long* count, uint64_t cap) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (count[b * kCursorStride] > cap) count[b * kCursorStride] = cap;
}

hipError_t launch_regrow_regions(const uint64_t* src, uint32_t nreg, uint64_t src_cap, uint64_t* dst, uint64_t dst_cap, unsigned long long* count,
                                 hipStream_t st) {
    hipLaunchKernelGGL(regrow_regions_kernel, dim3(8, nreg), dim3(256), 0, st, src, src_cap, dst, dst_cap, count);
    hipLaunchKernelGGL(clamp_cursors_kernel, dim3(nreg / 256), dim3(256), 0, st, count, src_cap);  // (nreg: a multiple of 256)
    return hipGetLastError();
}

size_t distinct_dedupe_lds(const DedupeArgs& D) {
    return (size_t)D.set_slots * 8 + (D.direct_keys ? (size_t)D.direct_keys * 4 : (size_t)D.lds_counters * 12);
}

hipError_t launch_distinct_dedupe(const Program& P, const GlobalTable& G, const DedupeArgs& D, uint32_t grid, uint32_t block, hipStream_t st) {
    const size_t shmem = distinct_dedupe_lds(D);
    grid = std::max(grid, (D.nbins + kDedupeOwn - 1) / kDedupeOwn);  // (the bounds of a workgroup's bins live in LDS)
    const bool together = !(block & 1u), four = (block & 2u) != 0;  // (tuning bits in the block size)
    block &= ~3u;
#define N1K_DEDUPE(B, U)                                                                                              \
    do {                                                                                                              \
        auto k = together ? distinct_dedupe_kernel<B, U, true> : distinct_dedupe_kernel<B, U, false>;                 \
        if (shmem > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem); \
        hipLaunchKernelGGL(k, dim3(grid), dim3(B), shmem, st, P, G, D);                                               \
    } while (0)
    if (block == 256) N1K_DEDUPE(256, 8);
    else if (block == 1024 && four) N1K_DEDUPE(1024, 4);
    else if (block == 1024) N1K_DEDUPE(1024, 2);
    else N1K_DEDUPE(512, 4);
#undef N1K_DEDUPE
    return hipGetLastError();
}

hipError_t launch_distinct_words_global(const GlobalTable& G, const uint64_t* words, uint64_t n, uint64_t* table, uint64_t mask,
                                        uint32_t key_shift, unsigned long long* counts, uint32_t* err_flags, uint32_t grid,
                                        hipStream_t st) {
    hipLaunchKernelGGL(distinct_words_global_kernel, dim3(grid), dim3(512), 0, st, G, words, n, table, mask, key_shift, counts, err_flags);
    return hipGetLastError();
}

hipError_t launch_distinct_add_counts(const Program& P, const GlobalTable& G, const unsigned long long* counts, uint32_t glob_off,
                                      hipStream_t st, const uint32_t* veto) {
    uint32_t blocks = (uint32_t)((G.capacity + 255) / 256);
    hipLaunchKernelGGL(distinct_add_counts_kernel, dim3(blocks), dim3(256), 0, st, P, G, counts, glob_off, veto);
    return hipGetLastError();
}

hipError_t launch_distinct_layout(const Program& P, const GlobalTable& G, const DistinctArgs& D, hipStream_t st) {
    uint32_t blocks = (uint32_t)((G.capacity + 255) / 256);
    hipLaunchKernelGGL(distinct_layout_kernel, dim3(blocks), dim3(256), 0, st, P, G, D);
    return hipGetLastError();
}

hipError_t launch_distinct_insert(const Program& P, const GlobalTable& G, const DistinctArgs& D, uint32_t* err_flags,
                                  hipStream_t st) {
    if (D.npairs == 0) return hipSuccess;
    uint32_t lds_counters = G.capacity <= 16384 ? (uint32_t)G.capacity : 0u;
    uint32_t blocks = (uint32_t)std::min<uint64_t>((D.npairs + 511) / 512, 2048);
    hipLaunchKernelGGL((distinct_insert_kernel<512>), dim3(blocks), dim3(512), lds_counters * 4, st, P, G, D, err_flags,
                       lds_counters);
    return hipGetLastError();
}

hipError_t launch_export_partials(const Program& P, const GlobalTable& G, uint32_t nparts, uint64_t cap, uint64_t* out,
                                  uint64_t region_words, uint32_t* err_flags, hipStream_t st) {
    uint32_t blocks = (uint32_t)((G.capacity + 255) / 256);
    hipLaunchKernelGGL(export_partials_kernel, dim3(blocks), dim3(256), 0, st, P, G, nparts, cap, out, region_words, err_flags);
    return hipGetLastError();
}

hipError_t launch_merge_partials(const Program& P, const GlobalTable& G, uint32_t nregions, uint64_t cap, const uint64_t* in,
                                 uint64_t region_words, uint32_t* err_flags, unsigned long long* ngroups, hipStream_t st,
                                 uint64_t limit, bool unique_keys) {
    uint64_t total = (uint64_t)nregions * cap;
    if (nregions == 1 && limit && limit < total) total = limit;  // one region whose count the caller knows
    if (!total) return hipSuccess;
    uint32_t blocks = (uint32_t)((total + 255) / 256);
    hipLaunchKernelGGL(merge_partials_kernel, dim3(blocks), dim3(256), 0, st, P, G, nregions, cap, in, region_words, err_flags, ngroups,
                       unique_keys ? 1u : 0u);
    return hipGetLastError();
}

hipError_t launch_arith(const ArithArgs& A, hipStream_t st) {
    if (A.nrows == 0) return hipSuccess;
    uint32_t blocks = (uint32_t)((A.nrows + 255) / 256);
    hipLaunchKernelGGL(arith_kernel, dim3(blocks), dim3(256), 0, st, A);
    return hipGetLastError();
}

// Received row regions (per source: kRowSubs sub-region counts kCursorStride words apart, the verdict in word 1): a sender
// whose region overflowed said so in every header — the receiver then aggregates nothing (all counts to zero) and its
// n1k_finish reports it, on every rank alike.
__global__ void exchange_verdict_kernel(const HeaderList H, uint32_t nregions, uint32_t* err_flags) {
    unsigned long long v = 0, status = 0;
    for (uint32_t r = 0; r < nregions; r++) {
        const unsigned long long x = H.h[r][1];
        v |= x;
        status = (x >> VD_STATUS_SHIFT) > status ? (x >> VD_STATUS_SHIFT) : status;
    }
    if (!v) return;
    if (threadIdx.x == 0) {
        raise_verdict(v, err_flags);
        if (status) atomicMax((unsigned long long*)err_flags + kPeerStatusFromErr, status & 0xFFull);
    }
    for (uint32_t i = threadIdx.x; i < nregions * kRowSubs; i += blockDim.x) H.h[i / kRowSubs][(size_t)(i % kRowSubs) * kCursorStride] = 0;
}

hipError_t launch_exchange_verdict(const HeaderList& H, uint32_t nregions, uint32_t* err_flags, hipStream_t st) {
    hipLaunchKernelGGL(exchange_verdict_kernel, dim3(1), dim3(64), 0, st, H, nregions, err_flags);
    return hipGetLastError();
}

// The sending side: behind the partition (or the export) of a step, this sender's own error flags — rows it dropped because
// their key does not pack, values its Filter cannot order — and, if its host part failed, that status join the verdict word of
// EVERY region it ships (n1k_types.h VD_*): the step then fails on every rank alike instead of returning short groups.
__global__ void stamp_verdict_kernel(unsigned long long* headers, uint32_t nregions, uint64_t stride_words, const uint32_t* err_flags,
                                     uint32_t host_status) {
    const uint32_t f = err_flags ? *err_flags : 0u;
    const unsigned long long v = ((f & ERR_UNPACKABLE_KEY) ? VD_UNPACKABLE : 0ull) | ((f & ERR_UNSUPPORTED_VALUE) ? VD_UNSUPPORTED : 0ull) |
                                 ((unsigned long long)(host_status & 0xFFu) << VD_STATUS_SHIFT);
    if (!v) return;
    for (uint32_t r = threadIdx.x; r < nregions; r += blockDim.x) atomicOr(&headers[(size_t)r * stride_words + 1], v);
}

hipError_t launch_stamp_verdict(unsigned long long* headers, uint32_t nregions, uint64_t stride_words, const uint32_t* err_flags,
                                uint32_t host_status, hipStream_t st) {
    hipLaunchKernelGGL(stamp_verdict_kernel, dim3(1), dim3(64), 0, st, headers, nregions, stride_words, err_flags, host_status);
    return hipGetLastError();
}

// A region written as ONE dense run of header[0] rows (the interpreting partition kernel; small batches) read as kRowSubs
// segments of sub_rows rows: the first ones full, then a partial one, then empty ones.
struct DestCaps { uint32_t per_dest, cap[kMaxParts]; };
__global__ void dense_to_segments_kernel(unsigned long long* headers, uint32_t nregions, uint64_t stride_words, uint64_t sub_rows, const DestCaps C) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nregions * kRowSubs) return;
    unsigned long long* h = headers + (size_t)(i / kRowSubs) * stride_words;
    const uint32_t x = i % kRowSubs;
    if (C.per_dest) sub_rows = C.cap[i / kRowSubs] / kRowSubs;
    const unsigned long long total = h[0];
    __syncthreads();  // (a region's kRowSubs threads sit in one workgroup: everybody has read the total before word 0 changes)
    const unsigned long long lo = (unsigned long long)x * sub_rows;
    h[(size_t)x * kCursorStride] = total <= lo ? 0ull : (total - lo < sub_rows ? total - lo : sub_rows);
}

hipError_t launch_dense_to_segments(unsigned long long* headers, uint32_t nregions, uint64_t stride_words, uint64_t sub_rows, hipStream_t st,
                                    const uint32_t* dest_cap) {
    const uint32_t n = nregions * kRowSubs;
    DestCaps C{};
    if (dest_cap) {
        C.per_dest = 1;
        for (uint32_t d = 0; d < nregions && d < kMaxParts; d++) C.cap[d] = dest_cap[d];
    }
    hipLaunchKernelGGL(dense_to_segments_kernel, dim3((n + 63) / 64), dim3(64), 0, st, headers, nregions, stride_words, sub_rows, C);
    return hipGetLastError();
}

hipError_t launch_partition(const Program& P, const PartArgs& A, uint32_t grid, hipStream_t st) {
    hipLaunchKernelGGL((partition_kernel<4, 512>), dim3(grid), dim3(512), 0, st, P, A);
    return hipGetLastError();
}

// the device counters into pinned host memory, behind the kernels that update them (n1k_finish)
__global__ void publish_counters_kernel(const unsigned long long* src, unsigned long long* dst, uint32_t n) {
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
}
hipError_t launch_publish_counters(const unsigned long long* src, unsigned long long* dst, uint32_t n, hipStream_t st) {
    hipLaunchKernelGGL(publish_counters_kernel, dim3(1), dim3(64), 0, st, src, dst, n);
    return hipGetLastError();
}

hipError_t launch_finalize(const Program& P, const GlobalTable& G, OutValue* out_keys, OutValue* out_aggs,
                           OutPartial* out_parts, uint64_t* out_rep, unsigned long long* out_count, uint64_t max_out,
                           uint32_t* err_flags, hipStream_t st) {
    // small tables: one 256-slot step per workgroup (no serial loop); big tables: fewer, longer chunks so that the
    // position counter sees thousands of atomics, not millions
    uint32_t chunk = (uint32_t)std::min<uint64_t>(kFinalizeChunkMax, std::max<uint64_t>(256, (G.capacity / 4096 + 255) / 256 * 256));
    uint32_t blocks = (uint32_t)((G.capacity + chunk - 1) / chunk);
    hipLaunchKernelGGL(finalize_kernel, dim3(blocks), dim3(256), 0, st, P, G, out_keys, out_aggs, out_parts, out_rep,
                       out_count, max_out, err_flags, chunk);
    return hipGetLastError();
}

hipError_t launch_filter_mask(const Program& P, uint64_t nrows, uint64_t* mask_words, uint32_t* tile_counts,
                              uint32_t* err_flags, uint32_t grid, hipStream_t st) {
    hipLaunchKernelGGL((filter_mask_kernel<4, 256>), dim3(grid), dim3(256), 0, st, P, nrows, mask_words, tile_counts,
                       err_flags);
    return hipGetLastError();
}

hipError_t launch_tile_scan(const uint32_t* counts, uint64_t* offsets, uint64_t ntiles, unsigned long long* total,
                            hipStream_t st) {
    hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, st, counts, offsets, ntiles, total);
    return hipGetLastError();
}

hipError_t launch_filter_compact(const uint64_t* mask_words, const uint64_t* tile_offsets, uint64_t nrows,
                                 uint64_t row_base, uint64_t* out_rows, uint32_t grid, hipStream_t st) {
    hipLaunchKernelGGL(filter_compact_kernel, dim3(grid), dim3(256), 0, st, mask_words, tile_offsets, nrows, row_base,
                       out_rows);
    return hipGetLastError();
}

hipError_t launch_synth(const SynthArgs& a, hipStream_t st) {
    if (a.nrows == 0) return hipSuccess;
    uint32_t blocks = (uint32_t)((a.nrows + 255) / 256);
    hipLaunchKernelGGL(synth_kernel, dim3(blocks), dim3(256), 0, st, a);
    return hipGetLastError();
}

}  // namespace n1k
