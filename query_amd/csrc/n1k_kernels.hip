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
#include "n1k_kernels.h"

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
        default:  // TERM_TRUTH
#pragma unroll
            for (int j = 0; j < R; j++) out[j] = truth_l(ta[j], pa[j], P.empty_str_code, P.empty_arr_code, P.empty_obj_code);
            break;
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

// ------------------------------------------------------------------ hash tables

// open-addressed LDS table: slot s occupies words [s*W, (s+1)*W); word 0 is the packed key
N1K_DEV int lds_find_or_insert(uint64_t* lds, uint32_t S, uint32_t W, uint64_t key, uint32_t* fill, uint32_t max_fill) {
    uint32_t h = (uint32_t)(((mix64(key) >> 32) * (uint64_t)S) >> 32);  // S need not be a power of two
    for (int probe = 0; probe < 32; probe++) {
        volatile uint64_t* p = &lds[(size_t)h * W];
        uint64_t cur = *p;
        if (cur == key) return (int)h;
        if (cur == kEmptyKey) {
            if (*(volatile uint32_t*)fill >= max_fill) return -1;  // table is kept sparse: new keys bypass LDS
            unsigned long long old = atomicCAS((unsigned long long*)p, (unsigned long long)kEmptyKey, (unsigned long long)key);
            if (old == kEmptyKey) {
                atomicAdd(fill, 1u);
                return (int)h;
            }
            if (old == key) return (int)h;
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
                atomicAdd(ngroups, 1ull);
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

N1K_DEV void lds_slot_init(const Program& P, uint64_t* s) {
    s[0] = kEmptyKey;
    if (P.want_rep_row) s[P.rep_lds_word] = ~0ull;
    for (uint32_t a = 0; a < P.naggs; a++) {
        const AggSpec& ag = P.aggs[a];
        if (ag.distinct) continue;
        uint64_t* w = s + ag.lds_off;
        switch (ag.kind) {
            case AGG_COUNT:
            case AGG_COUNTN: w[0] = 0; break;
            case AGG_SUM:
            case AGG_AVG: w[0] = 0; w[1] = 0; w[2] = 0; w[3] = 0; break;
            case AGG_MIN: w[0] = 0; w[1] = (uint64_t)INT64_MAX; w[2] = ~0ull; w[3] = ~0ull; break;
            default: w[0] = 0; w[1] = (uint64_t)INT64_MIN; w[2] = 0; w[3] = 0; break;
        }
    }
}

__device__ __forceinline__ void glob_row_init(const Program& P, uint64_t* g) {
    for (uint32_t a = 0; a < P.naggs; a++) {
        const AggSpec& ag = P.aggs[a];
        uint64_t* w = g + ag.glob_off;
        if (ag.distinct) { w[0] = 0; continue; }
        switch (ag.kind) {
            case AGG_COUNT:
            case AGG_COUNTN: w[0] = 0; break;
            case AGG_SUM:
            case AGG_AVG: for (int i = 0; i < 6; i++) w[i] = 0; break;
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
                atomicAdd(&w[x < 0 ? 4 : 3], 1ull);
            } else if (tag == T_FLOAT) {
                atomicAdd((double*)&w[2], as_f64(p));
                atomicAdd(&w[5], 1ull);
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

// CumulateInitial into the workgroup's LDS slot.  Returns false when the value must take the global path
// (|int| >= 2^40: the 64-bit LDS sum of a workgroup's share could overflow).
N1K_DEV bool acc_lds(const Program& P, const AggSpec& ag, uint64_t* s, uint32_t tag, uint64_t p) {
    unsigned long long* w = (unsigned long long*)(s + ag.lds_off);
    switch (ag.kind) {
        case AGG_COUNT:
            if (!ag.has_operand || tag > T_NULL) atomicAdd(&w[0], 1ull);
            return true;
        case AGG_COUNTN:
            if (tag == T_INT || tag == T_FLOAT) atomicAdd(&w[0], 1ull);
            return true;
        case AGG_SUM:
        case AGG_AVG:
            if (tag == T_INT) {
                int64_t x = (int64_t)p;
                if (x >= (1ll << 40) || x <= -(1ll << 40)) return false;
                atomicAdd(&w[0], (unsigned long long)x);
                atomicAdd(&w[2], x < 0 ? (1ull << 32) : 1ull);
            } else if (tag == T_FLOAT) {
                atomicAdd((double*)&w[1], as_f64(p));
                atomicAdd(&w[3], 1ull);
            }
            return true;
        default: {  // MIN / MAX
            if (tag <= T_NULL) return true;
            bool mn = ag.kind == AGG_MIN;
            volatile unsigned long long* vw = w;
            if (tag == T_FALSE || tag == T_TRUE) {
                unsigned long long bit = tag == T_TRUE ? (unsigned long long)MM_TRUE : (unsigned long long)MM_FALSE;
                if (!(vw[0] & bit)) atomicOr(&w[0], bit);
            } else if (tag == T_INT) {
                if (!(vw[0] & MM_INT)) atomicOr(&w[0], (unsigned long long)MM_INT);
                long long x = (long long)p, cur = (long long)vw[1];
                if (mn ? x < cur : x > cur) { if (mn) atomicMin((long long*)&w[1], x); else atomicMax((long long*)&w[1], x); }
            } else if (tag == T_FLOAT) {
                if (!(vw[0] & MM_FLOAT)) atomicOr(&w[0], (unsigned long long)MM_FLOAT);
                unsigned long long x = f64_sortable(as_f64(p)), cur = vw[2];
                if (mn ? x < cur : x > cur) { if (mn) atomicMin(&w[2], x); else atomicMax(&w[2], x); }
            } else if (tag == T_STRING) {
                if (!(vw[0] & MM_STRING)) atomicOr(&w[0], (unsigned long long)MM_STRING);
                unsigned long long x = ((unsigned long long)P.str_rank[(uint32_t)p] << 32) | (uint32_t)p, cur = vw[3];
                if (mn ? x < cur : x > cur) { if (mn) atomicMin(&w[3], x); else atomicMax(&w[3], x); }
            } else {
                if (!(vw[0] & MM_OTHER)) atomicOr(&w[0], (unsigned long long)MM_OTHER);
            }
            return true;
        }
    }
}

// CumulateIntermediate: fold one LDS slot into its global row (algebra/agg_*.go CumulateIntermediate)
N1K_DEV void merge_slot(const Program& P, const uint64_t* s, uint64_t* g) {
    for (uint32_t a = 0; a < P.naggs; a++) {
        const AggSpec& ag = P.aggs[a];
        if (ag.distinct) continue;
        const uint64_t* l = s + ag.lds_off;
        unsigned long long* w = (unsigned long long*)(g + ag.glob_off);
        switch (ag.kind) {
            case AGG_COUNT:
            case AGG_COUNTN:
                if (l[0]) atomicAdd(&w[0], (unsigned long long)l[0]);
                break;
            case AGG_SUM:
            case AGG_AVG: {
                if (l[2]) {
                    int64_t x = (int64_t)l[0];
                    atomicAdd(&w[0], (unsigned long long)(uint32_t)x);
                    atomicAdd(&w[1], (unsigned long long)(x >> 32));
                    uint64_t nn = l[2] & 0xFFFFFFFFull, ng = l[2] >> 32;
                    if (nn) atomicAdd(&w[3], (unsigned long long)nn);
                    if (ng) atomicAdd(&w[4], (unsigned long long)ng);
                }
                if (l[3]) {
                    atomicAdd((double*)&w[2], as_f64(l[1]));
                    atomicAdd(&w[5], (unsigned long long)l[3]);
                }
                break;
            }
            default: {
                if (!l[0]) break;
                bool mn = ag.kind == AGG_MIN;
                atomicOr(&w[0], (unsigned long long)l[0]);
                if (l[0] & MM_INT) { if (mn) atomicMin((long long*)&w[1], (long long)l[1]); else atomicMax((long long*)&w[1], (long long)l[1]); }
                if (l[0] & MM_FLOAT) { if (mn) atomicMin(&w[2], (unsigned long long)l[2]); else atomicMax(&w[2], (unsigned long long)l[2]); }
                if (l[0] & MM_STRING) { if (mn) atomicMin(&w[3], (unsigned long long)l[3]); else atomicMax(&w[3], (unsigned long long)l[3]); }
                break;
            }
        }
    }
}

// ------------------------------------------------------------------ K1+K2+K3(+K4): scan -> filter -> group

template <int R, int BLOCK>
__global__ __launch_bounds__(BLOCK) void scan_group_kernel(const Program P, const ScanArgs A, const GlobalTable G,
                                                          unsigned long long* ngroups) {
    extern __shared__ uint64_t lds[];
    __shared__ uint32_t lds_fill;
    const uint32_t S = A.lds_slots, W = P.lds_words;
    const uint32_t tid = threadIdx.x;

    for (uint32_t s = tid; s < S; s += BLOCK) lds_slot_init(P, &lds[(size_t)s * W]);
    if (tid == 0) lds_fill = 0;
    __syncthreads();

    uint32_t unsupported = 0, unpackable = 0;
    unsigned long long selected = 0;
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

        // group key -> one packed word per row (execution/group_util.go:18-35)
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
                uint64_t f = 0;
                if (pass[j] && !pack_key_field(ks, kt[j], kp[j], f)) {
                    unpackable = 1;
                    pass[j] = false;
                }
                key[j] |= f << ks.shift;
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
                slot[j] = lds_find_or_insert(lds, S, W, key[j], &lds_fill, A.lds_max_fill);
                if (slot[j] < 0) {
                    grow[j] = global_find_or_insert(G, key[j], A.err_flags, ngroups);
                    if (grow[j] < 0) pass[j] = false;
                }
                if (P.want_rep_row && pass[j]) {
                    unsigned long long ord = A.row_base + row[j];
                    if (slot[j] >= 0) atomicMin((unsigned long long*)&lds[(size_t)slot[j] * W + P.rep_lds_word], ord);
                    else atomicMin((unsigned long long*)&G.rep_row[grow[j]], ord);
                }
            }
        }

        // CumulateInitial of every aggregate (execution/group_initial.go:89-97)
        for (uint32_t a = 0; a < P.naggs; a++) {
            const AggSpec& ag = P.aggs[a];
            if (ag.distinct) continue;
            uint32_t vt[R];
            uint64_t vp[R];
            if (ag.has_operand) load_operand<R>(P, ag.src, row, pass, vt, vp);
#pragma unroll
            for (int j = 0; j < R; j++) {
                if (!pass[j]) continue;
                uint32_t t = ag.has_operand ? vt[j] : (uint32_t)T_NULL;
                uint64_t p = ag.has_operand ? vp[j] : 0ull;
                if (slot[j] >= 0) {
                    if (!acc_lds(P, ag, &lds[(size_t)slot[j] * W], t, p)) {
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
    // rows that passed the Filter (≙ Filter #itemsOut)
    for (int off = 32; off > 0; off >>= 1) selected += __shfl_down(selected, off, 64);
    if ((tid & 63) == 0 && selected) atomicAdd(A.rows_selected, selected);

    __syncthreads();
    // K4: merge this workgroup's partial groups into the global table
    for (uint32_t s = tid; s < S; s += BLOCK) {
        const uint64_t* sp = &lds[(size_t)s * W];
        if (sp[0] == kEmptyKey) continue;
        long long g = global_find_or_insert(G, sp[0], A.err_flags, ngroups);
        if (g < 0) continue;
        merge_slot(P, sp, &G.acc[(size_t)g * P.glob_words]);
        if (P.want_rep_row && sp[P.rep_lds_word] != ~0ull)
            atomicMin((unsigned long long*)&G.rep_row[g], (unsigned long long)sp[P.rep_lds_word]);
    }
}

__global__ void init_table_kernel(const Program P, const GlobalTable G, uint64_t first, uint64_t count) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint64_t s = first + i;
    G.keys[s] = kEmptyKey;
    if (G.rep_row) G.rep_row[s] = ~0ull;
    glob_row_init(P, &G.acc[(size_t)s * P.glob_words]);
}

// grow the global table: re-insert every occupied slot (keys keep their packed form)
__global__ void rehash_kernel(const Program P, const GlobalTable oldt, const GlobalTable newt, uint32_t* err_flags,
                              unsigned long long* scratch) {
    uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= oldt.capacity) return;
    uint64_t key = oldt.keys[s];
    if (key == kEmptyKey) return;
    long long g = global_find_or_insert(newt, key, err_flags, scratch);
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
    if (ag.distinct) {
        pt.distinct = (int64_t)w[0];
        put_value(fin, T_INT, w[0]);  // COUNT/COUNTN DISTINCT (SUM/AVG DISTINCT are finished by the distinct pass)
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
            uint64_t nn = w[3], ng = w[4], nf = w[5];
            uint64_t n = nn + ng + nf;
            // exact 128-bit integer total = hi * 2^32 + lo
            __int128 tot = ((__int128)(int64_t)w[1] << 32) + (__int128)(unsigned __int128)w[0];
            bool fits = tot >= (__int128)INT64_MIN && tot <= (__int128)INT64_MAX;
            // intValue.Add keeps an int only for same-sign operands without overflow (value/integer.go:266-277)
            bool int_exact = nf == 0 && !(nn > 0 && ng > 0) && fits;
            double itot = (double)(int64_t)w[1] * 4294967296.0 + (double)w[0];
            if (fits) itot = (double)(int64_t)tot;
            double fsum = as_f64(w[2]);
            pt.count = (int64_t)n;
            pt.isum = int_exact ? (int64_t)tot : 0;
            pt.fsum = int_exact ? fsum : fsum + itot;
            pt.flags = (int_exact ? 1u : 0u) | (nf ? 2u : 0u);
            if (n == 0) {
                put_value(fin, T_NULL, 0);  // Default(): NULL (agg_sum.go:77, agg_avg.go:77)
            } else if (ag.kind == AGG_SUM) {
                if (int_exact) put_value(fin, T_INT, (uint64_t)(int64_t)tot);
                else put_value(fin, T_FLOAT, f64_bits(fsum + itot));
            } else {
                double s = int_exact ? (double)(int64_t)tot : fsum + itot;
                double avg = s / (double)n;  // agg_avg.go:124-125 -> value.NewValue folds integral results
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

__global__ void finalize_kernel(const Program P, const GlobalTable G, OutValue* out_keys, OutValue* out_aggs,
                                OutPartial* out_parts, uint64_t* out_rep, unsigned long long* out_count,
                                uint64_t max_out, uint32_t* err_flags) {
    uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= G.capacity) return;
    uint64_t key = G.keys[s];
    if (key == kEmptyKey) return;
    unsigned long long idx = atomicAdd(out_count, 1ull);
    if (idx >= max_out) return;
    for (uint32_t k = 0; k < P.nkeys; k++) {
        const KeySpec& ks = P.keys[k];
        uint64_t field = ks.bits >= 64 ? key : ((key >> ks.shift) & ((1ull << ks.bits) - 1ull));
        uint32_t tag;
        uint64_t p;
        unpack_key_field(ks.mode, field, tag, p);
        put_value(&out_keys[idx * P.nkeys + k], tag, p);
    }
    const uint64_t* g = &G.acc[(size_t)s * P.glob_words];
    for (uint32_t a = 0; a < P.naggs; a++)
        finalize_agg(P, P.aggs[a], g, &out_aggs[idx * P.naggs + a], &out_parts[idx * P.naggs + a], err_flags);
    if (out_rep) out_rep[idx] = G.rep_row ? G.rep_row[s] : ~0ull;
}

// ------------------------------------------------------------------ Filter alone: mask, scan, compaction

// K1: one bit per row through a wave ballot; per-tile survivor counts.  Tile = FILTER_TILE rows.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void filter_mask_kernel(const Program P, uint64_t nrows, uint64_t* mask_words,
                                                           uint32_t* tile_counts, uint32_t* err_flags) {
    __shared__ uint32_t wave_cnt[BLOCK / 64];
    const uint32_t tid = threadIdx.x;
    uint32_t unsupported = 0;
    const uint64_t ntiles = (nrows + BLOCK - 1) / BLOCK;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        uint64_t row[1] = {tile * BLOCK + tid};
        bool valid[1] = {row[0] < nrows}, pass[1];
        eval_predicate<1>(P, row, valid, pass, unsupported);
        unsigned long long b = __ballot(pass[0]);
        if ((tid & 63) == 0) {
            mask_words[row[0] >> 6] = b;
            wave_cnt[tid >> 6] = (uint32_t)__popcll(b);
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t c = 0;
            for (int w = 0; w < BLOCK / 64; w++) c += wave_cnt[w];
            tile_counts[tile] = c;
        }
        __syncthreads();
    }
    if (unsupported) atomicOr(err_flags, (uint32_t)ERR_UNSUPPORTED_VALUE);
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

// K2: ordered compaction: row ordinals of the set bits, ascending
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void filter_compact_kernel(const uint64_t* mask_words, const uint64_t* tile_offsets,
                                                              uint64_t nrows, uint64_t row_base, uint64_t* out_rows) {
    __shared__ uint32_t wave_off[BLOCK / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t ntiles = (nrows + BLOCK - 1) / BLOCK;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        uint64_t row = tile * BLOCK + tid;
        uint64_t word = (tile * BLOCK + (uint64_t)wave * 64) < nrows ? mask_words[(tile * BLOCK >> 6) + wave] : 0ull;
        if (lane == 0) wave_off[wave] = (uint32_t)__popcll(word);
        __syncthreads();
        uint32_t before = 0;
        for (uint32_t w = 0; w < wave; w++) before += wave_off[w];
        bool set = (word >> lane) & 1ull;
        uint32_t rank = (uint32_t)__popcll(word & ((1ull << lane) - 1ull));
        if (set) out_rows[tile_offsets[tile] + before + rank] = row_base + row;
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

hipError_t launch_init_table(const Program& P, const GlobalTable& G, uint64_t first, uint64_t count, hipStream_t st) {
    if (count == 0) return hipSuccess;
    uint32_t blocks = (uint32_t)((count + 255) / 256);
    hipLaunchKernelGGL(init_table_kernel, dim3(blocks), dim3(256), 0, st, P, G, first, count);
    return hipGetLastError();
}

hipError_t launch_rehash(const Program& P, const GlobalTable& oldt, const GlobalTable& newt, uint32_t* err_flags,
                         unsigned long long* ngroups_scratch, hipStream_t st) {
    uint32_t blocks = (uint32_t)((oldt.capacity + 255) / 256);
    hipLaunchKernelGGL(rehash_kernel, dim3(blocks), dim3(256), 0, st, P, oldt, newt, err_flags, ngroups_scratch);
    return hipGetLastError();
}

hipError_t launch_scan_group(const Program& P, const ScanArgs& A, const GlobalTable& G, unsigned long long* ngroups,
                             uint32_t grid, hipStream_t st) {
    constexpr int R = 4, BLOCK = 256;
    size_t shmem = (size_t)A.lds_slots * P.lds_words * 8;
    hipLaunchKernelGGL((scan_group_kernel<R, BLOCK>), dim3(grid), dim3(BLOCK), shmem, st, P, A, G, ngroups);
    return hipGetLastError();
}

hipError_t launch_finalize(const Program& P, const GlobalTable& G, OutValue* out_keys, OutValue* out_aggs,
                           OutPartial* out_parts, uint64_t* out_rep, unsigned long long* out_count, uint64_t max_out,
                           uint32_t* err_flags, hipStream_t st) {
    uint32_t blocks = (uint32_t)((G.capacity + 255) / 256);
    hipLaunchKernelGGL(finalize_kernel, dim3(blocks), dim3(256), 0, st, P, G, out_keys, out_aggs, out_parts, out_rep,
                       out_count, max_out, err_flags);
    return hipGetLastError();
}

hipError_t launch_filter_mask(const Program& P, uint64_t nrows, uint64_t* mask_words, uint32_t* tile_counts,
                              uint32_t* err_flags, uint32_t grid, hipStream_t st) {
    hipLaunchKernelGGL((filter_mask_kernel<kFilterTile>), dim3(grid), dim3(kFilterTile), 0, st, P, nrows, mask_words,
                       tile_counts, err_flags);
    return hipGetLastError();
}

hipError_t launch_tile_scan(const uint32_t* counts, uint64_t* offsets, uint64_t ntiles, unsigned long long* total,
                            hipStream_t st) {
    hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, st, counts, offsets, ntiles, total);
    return hipGetLastError();
}

hipError_t launch_filter_compact(const uint64_t* mask_words, const uint64_t* tile_offsets, uint64_t nrows,
                                 uint64_t row_base, uint64_t* out_rows, uint32_t grid, hipStream_t st) {
    hipLaunchKernelGGL((filter_compact_kernel<kFilterTile>), dim3(grid), dim3(kFilterTile), 0, st, mask_words,
                       tile_offsets, nrows, row_base, out_rows);
    return hipGetLastError();
}

hipError_t launch_synth(const SynthArgs& a, hipStream_t st) {
    if (a.nrows == 0) return hipSuccess;
    uint32_t blocks = (uint32_t)((a.nrows + 255) / 256);
    hipLaunchKernelGGL(synth_kernel, dim3(blocks), dim3(256), 0, st, a);
    return hipGetLastError();
}

}  // namespace n1k
