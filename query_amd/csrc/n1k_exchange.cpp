// n1k_exchange.cpp — partial groups in and out of a handle, the hash partition of the row exchange, and the collectives
// (RCCL, or the loopback transport) behind n1k_comm_* / n1k_exchange_* / n1k_gather_groups.
#include "n1k_engine.h"

using namespace n1k;
using namespace n1k_eng;

#include <rccl/rccl.h>

// Filter + hash partition of a bound batch (bind_columns(h, b, defer = true) came first): the run-time-built kernel of the
// plan's shape when there is one (large batches, or jit = 2: n1k_spec.h scan_spec_partition_body — wide loads, arithmetic
// in registers, survivors written in runs), else the interpreting partition_kernel over materialised derived columns.
static n1k_status run_partition(n1k_handle* h, const n1k_batch* b, PartArgs& A) {
    h->device_clean = false;
    // packed regions (the row exchange): A.sub_rows = rows per sub-region; whoever writes dense runs converts the counts
    const uint64_t seg_rows = A.region_bytes ? A.sub_rows : 0;
    A.nsub = 1;
    if (b->nrows == 0) {
        if (seg_rows) HIP_TRY(h, launch_dense_to_segments(A.counts, A.nparts, A.count_stride, seg_rows, h->stream, A.per_dest ? A.dest_cap : nullptr));
        return materialize_derived(h, b);
    }
    const uint64_t n = b->nrows;
    FastArgs F;
    const JitKernel* jit = nullptr;
    const bool fuse = !h->derived.empty() && !h->derived_ready;
    if (h->opt_spec && h->opt_jit && (h->opt_jit == 2 || n >= h->opt_jit_min_rows) && n < (1ull << 31) && (!fuse || h->opt_fuse_arith) &&
        sizeof(Program) + sizeof(FastArgs) + sizeof(PartArgs) + 64 <= 4096 && build_fast_args(h, 1u << 15, F, fuse, true)) {
        // what the staging needs in LDS (n1k_spec.h PartLds: 2048 rows x (9 B per TAGGED64 column, 4 B per DICT32 column, 1))
        size_t lds = 2048 + 2048 + 4096;  // (+ the per-destination tables of PartLds)
        for (uint32_t c = 0; c < F.ncols; c++) lds += 2048u * (F.cols[c].kind == COLK_DICT32 ? 4u : 9u);
        if (lds <= 60 * 1024) {
            SpecSig sig = make_plan_sig(h, F);
            sig.mode = 1;
            sig.hashed = 0;  // (no table in this mode)
            jit = jit_get(sig);
            if (jit->failed || !jit->part_wide) {
                h->jit_log = jit->log;
                jit = nullptr;
            }
        }
    }
    hipEvent_t e0 = get_event(h), e1 = get_event(h);
    if (jit) {
        bool aligned = true;
        for (uint32_t c = 0; c < F.ncols; c++) {
            F.cols[c] = h->prog.cols[c];
            aligned &= ((uintptr_t)F.cols[c].tags % 2 == 0) && ((uintptr_t)F.cols[c].payload % 16 == 0) && ((uintptr_t)F.cols[c].codes % 8 == 0);
        }
        const bool wide = aligned && h->opt_wide && n >= 2;
        F.nrows = (uint32_t)n;
        F.row_base = h->row_base;
        F.err_flags = h->d_errp;
        // 256-thread workgroups (tiles of 1024 rows, one in flight, six per CU: many independent workgroups overlap the wait
        // for each tile's reservation) or 512-thread ones (2048 rows, two tiles in flight, two per CU)
        const uint32_t pblock = wide && h->opt_part_block == 256 ? 256u : 512u;
        const uint64_t tiles = (n + pblock * 4 - 1) / (pblock * 4);
        uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)h->num_cus * (h->opt_part_per_cu ? h->opt_part_per_cu : (pblock == 256 ? 6u : 2u)), tiles));
        // many tiles: every destination's region in kRowSubs sub-regions with their own counters, workgroups dealt round-robin
        // (tile t goes to sub-region t % kRowSubs: an even share of the rows whatever their order)
        if (seg_rows && h->opt_part_subs && (tiles >= 4096 || h->opt_part_subs == 2)) {
            A.nsub = kRowSubs;
            grid = (grid + kRowSubs - 1) / kRowSubs * kRowSubs;
        }
        if (e0) (void)hipEventRecord(e0, h->stream);
        HIP_TRY(h, jit_launch_partition(jit, h->prog, F, A, grid, wide, pblock, h->stream));
        h->stats.spec_kernel = F.nderived ? 3u : 2u;
    } else {
        n1k_status st = materialize_derived(h, b);
        if (st != N1K_OK) return st;
        const uint64_t ntiles = (n + 2047) / 2048;  // partition_kernel<4, 512>
        const uint32_t grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)h->num_cus * 4, ntiles));
        if (e0) (void)hipEventRecord(e0, h->stream);
        HIP_TRY(h, launch_partition(h->prog, A, grid, h->stream));
        h->stats.spec_kernel = 0;
    }
    if (seg_rows && A.nsub == 1) HIP_TRY(h, launch_dense_to_segments(A.counts, A.nparts, A.count_stride, seg_rows, h->stream, A.per_dest ? A.dest_cap : nullptr));
    if (e1) (void)hipEventRecord(e1, h->stream);
    h->events.emplace_back(e0, e1);
    return N1K_OK;
}

extern "C" {

n1k_status n1k_partition_device_batch(n1k_handle* h, const n1k_batch* batch, uint32_t nparts, uint64_t capacity_rows,
                                      const n1k_col* out_cols, uint64_t* out_counts) {
    return guarded(h, [&]() -> n1k_status {
    if (!h || !batch || !out_cols || !out_counts || nparts == 0) return N1K_INVALID;
    if (nparts > kMaxParts) return fail(h, N1K_INVALID, "at most %u destinations per partition call", kMaxParts);
    if (h->stop_flag.load()) return fail(h, N1K_STOPPED, "operator was stopped");
    if (!h->plan.has_group) return fail(h, N1K_INVALID, "partitioning needs group keys");
    if (sizeof(Program) + sizeof(PartArgs) + 64 > 4096) return fail(h, N1K_UNSUPPORTED, "kernel arguments exceed 4 KiB");
    n1k_status st = ensure_device(h);
    if (st != N1K_OK) return st;
    st = validate_batch(h, batch);
    if (st != N1K_OK) return st;
    if (!h->layout_fixed) {
        st = fix_layout(h, batch);
        if (st != N1K_OK) return st;
    }
    PartArgs A{};
    st = bind_columns(h, batch, true);
    if (st != N1K_OK) return st;
    A.ncopy = (uint32_t)h->plan.paths.size();  // derived columns are recomputed by the receiver
    for (uint32_t c = 0; c < A.ncopy; c++) {
        if (out_cols[c].kind != batch->cols[c].kind) return fail(h, N1K_INVALID, "output column %u has another kind", c);
        A.out_tags[c] = (uint8_t*)out_cols[c].tags;
        A.out_payload[c] = (uint64_t*)out_cols[c].payload;
        A.out_codes[c] = (uint32_t*)out_cols[c].codes;
    }
    st = ensure_rank(h);
    if (st != N1K_OK) return st;
    A.nrows = batch->nrows;
    A.capacity = capacity_rows;
    A.nparts = nparts;
    A.counts = (unsigned long long*)out_counts;
    A.err_flags = h->d_errp;
    HIP_TRY(h, hipMemsetAsync(out_counts, 0, nparts * sizeof(uint64_t), h->stream));
    st = run_partition(h, batch, A);
    if (st != N1K_OK) return st;
    uint32_t err_flags = 0;
    HIP_TRY(h, hipMemcpyAsync(&err_flags, h->d_errp, 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (err_flags & ERR_TABLE_FULL) return fail(h, N1K_REGION_FULL, "partition region capacity (%llu rows) exceeded", (unsigned long long)capacity_rows);
    if (err_flags & ERR_UNPACKABLE_KEY) return fail(h, N1K_UNSUPPORTED_DATA, "a group key value does not fit the packed key");
    if (err_flags & ERR_UNSUPPORTED_VALUE) return fail(h, N1K_UNSUPPORTED_DATA, "a value outside the device subset was met");
    h->stats.rows_in += batch->nrows;
    h->stats.batches += 1;
    return N1K_OK;
    });
}

uint32_t n1k_partial_words(const n1k_handle* h) { return h ? h->prog.glob_words : 0; }

uint64_t n1k_partial_region_bytes(const n1k_handle* h, uint64_t capacity_groups) {
    if (!h) return 0;
    return 8ull * (2 + capacity_groups * (1 + (uint64_t)h->prog.glob_words));
}

n1k_status n1k_export_partials_async(n1k_handle* h, uint32_t nparts, uint64_t capacity_groups, void* out) {
    return guarded(h, [&]() -> n1k_status {
    if (!h || !out || nparts == 0 || capacity_groups == 0) return N1K_INVALID;
    if (h->pending.count) {
        n1k_status pst = flush_pending(h);
        if (pst != N1K_OK) return pst;
    }
    if (!h->plan.has_group) return fail(h, N1K_INVALID, "no groups in a Filter-only plan");
    if (h->has_distinct) return fail(h, N1K_UNSUPPORTED, "DISTINCT sets do not travel with partial groups");
    h->device_clean = false;  // (an overflow raises this handle's error flags)
    n1k_status st = ensure_device(h);
    if (st != N1K_OK) return st;
    uint64_t region_words = 2 + capacity_groups * (1 + (uint64_t)h->prog.glob_words);
    HIP_TRY(h, hipMemsetAsync(out, 0, (size_t)nparts * region_words * 8, h->stream));  // headers (and padding) to zero
    if (h->table.capacity)
        HIP_TRY(h, launch_export_partials(h->prog, h->table, nparts, capacity_groups, (uint64_t*)out, region_words,
                                          h->d_errp, h->stream));
    return N1K_OK;
    });
}

n1k_status n1k_export_partials_device(n1k_handle* h, uint32_t nparts, uint64_t capacity_groups, void* out) {
    return guarded(h, [&]() -> n1k_status {
    n1k_status st = n1k_export_partials_async(h, nparts, capacity_groups, out);
    if (st != N1K_OK) return st;
    uint32_t err_flags = 0;
    unsigned long long sel = 0, wide = 0;
    HIP_TRY(h, hipMemcpyAsync(&err_flags, h->d_errp, 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(&sel, h->d_counters.p, sizeof sel, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(&wide, h->d_counters.p + 13, sizeof wide, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->stats.rows_selected = sel;
    h->stats.wide_key_values = wide;
    // codes of the wide-value tables mean nothing on another device: such groups travel as rows instead
    if (wide) {
        HIP_TRY(h, hipMemsetAsync(h->d_errp, 0, 4, h->stream));  // a region overflow of the abandoned export is moot
        return fail(h, N1K_UNSUPPORTED, "group keys hold %llu float / wide integer values: use the row exchange", wide);
    }
    if (err_flags & ERR_TABLE_FULL) {
        HIP_TRY(h, hipMemsetAsync(h->d_errp, 0, 4, h->stream));
        return fail(h, N1K_REGION_FULL, "more than %llu groups for one destination: raise the region capacity",
                    (unsigned long long)capacity_groups);
    }
    return N1K_OK;
    });
}

n1k_status n1k_merge_partials_device(n1k_handle* h, uint32_t nregions, uint64_t capacity_groups, const void* in) {
    return guarded(h, [&]() -> n1k_status {
    if (!h || !in || nregions == 0 || capacity_groups == 0) return N1K_INVALID;
    if (h->has_distinct) return fail(h, N1K_UNSUPPORTED, "DISTINCT sets do not travel with partial groups");
    if (!h->layout_fixed) return fail(h, N1K_INVALID, "merge needs the key layout: push a batch (even an empty one) first");
    h->device_clean = false;
    if (h->pending.count) {
        n1k_status pst = flush_pending(h);
        if (pst != N1K_OK) return pst;
    }
    n1k_status st = ensure_device(h);
    if (st != N1K_OK) return st;
    // the incoming groups bound the growth of the table
    uint64_t saved = h->row_base;
    h->row_base += (uint64_t)nregions * capacity_groups;
    st = ensure_table(h, 0);
    h->row_base = saved;
    if (st != N1K_OK) return st;
    uint64_t region_words = 2 + capacity_groups * (1 + (uint64_t)h->prog.glob_words);
    HIP_TRY(h, launch_merge_partials(h->prog, h->table, nregions, capacity_groups, (const uint64_t*)in, region_words,
                                     h->d_errp, h->d_counters.p + 1, h->stream));
    h->merged_groups_bound += (uint64_t)nregions * capacity_groups;
    return N1K_OK;
    });
}

n1k_status n1k_export_groups(n1k_handle* h, const void** blob, size_t* len) {
    return guarded(h, [&]() -> n1k_status {
    if (!h || !blob || !len) return N1K_INVALID;
    n1k_status st = ensure_device(h);
    if (st != N1K_OK) return st;
    st = flush_pending(h);
    if (st != N1K_OK) return st;
    unsigned long long ng = 0;
    HIP_TRY(h, hipMemcpyAsync(&ng, h->d_counters.p + 1, sizeof ng, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    uint64_t cap = std::max<uint64_t>(ng, 1);
    uint64_t bytes = n1k_partial_region_bytes(h, cap);
    DevBuf<uint64_t> tmp;
    HIP_TRY(h, tmp.ensure(bytes / 8));
    st = n1k_export_partials_device(h, 1, cap, tmp.p);
    if (st == N1K_OK) {
        h->export_blob.resize(bytes + 16);
        uint64_t hdr[2] = {0x4e314b5041525431ull /* "N1KPART1" */, cap};
        memcpy(h->export_blob.data(), hdr, 16);
        hipError_t e = hipMemcpy(h->export_blob.data() + 16, tmp.p, bytes, hipMemcpyDeviceToHost);
        if (e != hipSuccess) st = fail(h, N1K_DEVICE_ERROR, "copy of exported groups failed: %s", hipGetErrorString(e));
    }
    tmp.release();
    if (st != N1K_OK) return st;
    *blob = h->export_blob.data();
    *len = h->export_blob.size();
    return N1K_OK;
    });
}

n1k_status n1k_merge_groups(n1k_handle* h, const void* blob, size_t len) {
    return guarded(h, [&]() -> n1k_status {
    if (!h || !blob || len < 32) return N1K_INVALID;
    uint64_t hdr[2];
    memcpy(hdr, blob, 16);
    if (hdr[0] != 0x4e314b5041525431ull) return fail(h, N1K_INVALID, "not an exported group blob");
    uint64_t cap = hdr[1];
    if (n1k_partial_region_bytes(h, cap) + 16 != len) return fail(h, N1K_INVALID, "blob does not match this plan");
    n1k_status st = ensure_device(h);
    if (st != N1K_OK) return st;
    DevBuf<uint64_t> tmp;
    HIP_TRY(h, tmp.ensure((len - 16) / 8));
    HIP_TRY(h, hipMemcpy(tmp.p, (const char*)blob + 16, len - 16, hipMemcpyHostToDevice));
    st = n1k_merge_partials_device(h, 1, cap, tmp.p);
    if (st == N1K_OK) HIP_TRY(h, hipStreamSynchronize(h->stream));
    tmp.release();
    return st;
    });
}

// ---------------------------------------------------------------- multi-GPU: RCCL behind the C ABI
//
// One communicator per rank (one rank per GPU; on one node every GPU pair has its own xGMI link, so the grouped
// send / recv of an all-to-all keeps all of a GPU's links busy at once).  Everything below is ordered on the sending
// handle's stream; the receiving handle's stream waits on an event; nothing waits on the host before n1k_finish.

}  // extern "C" (the communicator struct is C++)

// Loopback transport (n1k_comm_create_loopback): the ranks are threads of ONE process sharing one device — every
// collective is a rendezvous (barrier), device-to-device copies out of the peers' buffers, and a second rendezvous before
// anybody reuses its send buffer.  It exists so that the world_size > 1 code paths of the exchange (region offsets, header
// lists, verdicts, segmented receives, agreed capacities) can be run and checked on a single GPU; RCCL refuses two ranks on
// one device.
struct LoopHub {
    int world = 1;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    std::vector<const void*> ptr;
    std::vector<size_t> stride;  // all-to-all: distance between the regions a rank published (0: one region for every peer)
    std::vector<unsigned long long> val;
    int refs = 0;
    void barrier() {
        std::unique_lock<std::mutex> lk(mu);
        const uint64_t g = generation;
        if (++arrived == world) {
            arrived = 0;
            generation++;
            cv.notify_all();
        } else
            cv.wait(lk, [&] { return generation != g; });
    }
};

struct n1k_comm {
    ncclComm_t comm = nullptr;
    LoopHub* hub = nullptr;  // non-null: loopback transport
    int rank = 0, world = 1, device = 0;
    DevBuf<char> send, recv, gsend, grecv;
    DevBuf<char> void_send, void_recv;  // one region each: what a rank whose own part of a step failed ships / lets land (void_regions)
    DevBuf<unsigned long long> scalar;
    hipEvent_t ev = nullptr;
    hipEvent_t ev_consumed = nullptr;   // on the receiving stream, behind the last kernel that reads send / recv
    bool consumed_pending = false;
    bool failure_broadcast = false;     // the last exchange failed on this rank before the collective and its peers were told
    size_t sent_stride = 0;             // the last row exchange's regions in `send`: their distance and their capacities
    std::vector<uint64_t> sent_cap;
    std::string last_error;
    uint64_t gather_cap = 1024;    // records per slot of n1k_gather_groups (the same on every rank, see there)
    std::vector<char> ghost;       // gathered records on the host
    std::vector<n1k_value> gkeys, gaggs;
};

namespace {

n1k_status cfail(n1k_comm* c, n1k_status st, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->last_error = buf;
    g_create_error = buf;
    return st;
}

#define NCCL_TRY(c, expr)                                                                                   \
    do {                                                                                                    \
        ncclResult_t _r = (expr);                                                                           \
        if (_r != ncclSuccess) return cfail(c, N1K_DEVICE_ERROR, "%s failed: %s", #expr, ncclGetErrorString(_r)); \
    } while (0)
#define CHIP_TRY(c, expr)                                                                                   \
    do {                                                                                                    \
        hipError_t _e = (expr);                                                                             \
        if (_e != hipSuccess)                                                                               \
            return cfail(c, _e == hipErrorOutOfMemory ? N1K_OOM : N1K_DEVICE_ERROR, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

// all-to-all of equal regions: region p of `send` goes to rank p, region s of `recv` comes from rank s.  This rank's own
// region is not copied: the caller reads it where it lies (`self` returns its address).
// loopback: what every peer published, copied (or read) by everybody between two rendezvous
template <class Copy>
n1k_status loop_collective(n1k_comm* c, const void* send, hipStream_t st, Copy copy) {
    // (a rank whose own part fails still keeps both rendezvous: its peers must not be left waiting for it)
    hipError_t e = hipStreamSynchronize(st);
    c->hub->ptr[c->rank] = e == hipSuccess ? send : nullptr;
    c->hub->barrier();  // every rank's send buffer is complete and published
    if (e == hipSuccess) {
        for (int p = 0; p < c->world; p++)
            if (!c->hub->ptr[p]) e = hipErrorUnknown;  // (a peer failed before publishing)
    }
    if (e == hipSuccess) e = copy();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    c->hub->barrier();  // everybody has read everybody: the send buffers may be overwritten
    return e == hipSuccess ? N1K_OK : cfail(c, N1K_DEVICE_ERROR, "loopback: collective failed: %s", hipGetErrorString(e));
}

n1k_status all_gather_bytes(n1k_comm* c, const char* send, char* recv, size_t bytes, hipStream_t st) {
    if (c->hub)
        return loop_collective(c, send, st, [&]() -> hipError_t {
            for (int p = 0; p < c->world; p++) {
                hipError_t e = hipMemcpyAsync(recv + (size_t)p * bytes, c->hub->ptr[p], bytes, hipMemcpyDeviceToDevice, st);
                if (e != hipSuccess) return e;
            }
            return hipSuccess;
        });
    NCCL_TRY(c, ncclAllGather(send, recv, bytes, ncclChar, c->comm, st));
    return N1K_OK;
}

// All-to-all of equal regions: the region at send + p * send_stride goes to rank p, rank s's region lands at
// recv + s * recv_stride.  This rank's own region is not copied: the caller reads it where it lies (`self`).  A stride of 0
// is how a rank whose own part failed takes part (void_regions): one region for every peer, one sink for what arrives.
// Once the group is open it is always closed: no early return between ncclGroupStart and ncclGroupEnd.
n1k_status all_to_all_regions(n1k_comm* c, const char* send, size_t send_stride, char* recv, size_t recv_stride, size_t region, hipStream_t st,
                              const char** self, const size_t* send_bytes = nullptr) {
    // send_bytes: the region for rank p is send_bytes[p] long (regions of per-destination capacities: every rank sizes the regions
    // for destination p alike, so what this rank receives from everybody is `region` = send_bytes[its own rank] long)
    *self = send + (size_t)c->rank * send_stride;
    if (c->hub) {
        c->hub->stride[c->rank] = send_stride;  // (published with the pointer: read by the peers behind the first rendezvous)
        return loop_collective(c, send, st, [&]() -> hipError_t {
            for (int p = 0; p < c->world; p++) {
                if (p == c->rank) continue;
                hipError_t e = hipMemcpyAsync(recv + (size_t)p * recv_stride, (const char*)c->hub->ptr[p] + (size_t)c->rank * c->hub->stride[p], region,
                                              hipMemcpyDeviceToDevice, st);
                if (e != hipSuccess) return e;
            }
            return hipSuccess;
        });
    }
    NCCL_TRY(c, ncclGroupStart());
    ncclResult_t first = ncclSuccess;
    for (int p = 0; p < c->world; p++) {
        if (p == c->rank) continue;
        ncclResult_t r = ncclSend(send + (size_t)p * send_stride, send_bytes ? send_bytes[p] : region, ncclChar, p, c->comm, st);
        if (first == ncclSuccess) first = r;
        r = ncclRecv(recv + (size_t)p * recv_stride, region, ncclChar, p, c->comm, st);
        if (first == ncclSuccess) first = r;
    }
    const ncclResult_t end = ncclGroupEnd();
    if (first == ncclSuccess) first = end;
    if (first != ncclSuccess) return cfail(c, N1K_DEVICE_ERROR, "all-to-all of the regions failed: %s", ncclGetErrorString(first));
    return N1K_OK;
}

int sender_column(const n1k_handle* snd, const std::string& path) {
    for (size_t j = 0; j < snd->plan.paths.size(); j++)
        if (snd->plan.paths[j] == path) return (int)j;
    return -1;
}

// the receiving handle learns the key layout (column kinds) and the dictionary from the sending one: both were built
// from the same plan, in one process
n1k_status prepare_receiver(n1k_handle* snd, n1k_handle* rcv) {
    if (!snd->layout_fixed) return fail(snd, N1K_INVALID, "the sender has seen no batch yet");
    for (size_t i = rcv->dict.size(); i < snd->dict.size(); i++)
        if (intern(rcv, snd->dict[i]) != (uint32_t)i) return fail(rcv, N1K_INVALID, "sender and receiver dictionaries differ");
    n1k_status st = ensure_device(rcv);
    if (st != N1K_OK) return st;
    if (!rcv->layout_fixed) {
        // (the receiver has no Filter: its columns are the sender's in another order — matched by their path text)
        std::vector<n1k_col> cols(std::max<size_t>(1, rcv->plan.paths.size()));
        for (size_t i = 0; i < rcv->plan.paths.size(); i++) {
            const int j = sender_column(snd, rcv->plan.paths[i]);
            if (j < 0) return fail(rcv, N1K_INVALID, "the receiver's column %s is not a column of the sender", rcv->plan.paths[i].c_str());
            cols[i].kind = snd->col_kinds[j];
        }
        n1k_batch b{};
        b.ncols = (uint32_t)rcv->plan.paths.size();
        b.cols = cols.data();
        st = push_device(rcv, &b);  // an empty batch: fixes the layout, runs nothing
    }
    return st;
}

n1k_status order_streams(n1k_comm* c, n1k_handle* snd, n1k_handle* rcv) {
    if (snd->stream == rcv->stream) return N1K_OK;
    CHIP_TRY(c, hipEventRecord(c->ev, snd->stream));
    CHIP_TRY(c, hipStreamWaitEvent(rcv->stream, c->ev, 0));
    return N1K_OK;
}

// layout of one packed row region for `cap` rows (kRowSubs sub-regions of cap / kRowSubs rows) of the plan's input columns:
// the header (sub-region x's row count at word x * kCursorStride, the verdict in word 1), then per column its arrays, each
// starting on a 16-byte boundary
size_t row_region_layout(const n1k_handle* h, uint64_t cap, std::vector<size_t>& off_a, std::vector<size_t>& off_b) {
    size_t at = (size_t)kRowSubs * kCursorStride * 8;  // header: the sub-regions' counts, 128 bytes apart; verdict in word 1
    const size_t nc = h->plan.paths.size();
    off_a.assign(nc, 0);
    off_b.assign(nc, 0);
    auto pad = [](size_t x) { return (x + 15) / 16 * 16; };
    for (size_t i = 0; i < nc; i++) {
        if (h->col_kinds[i] == N1K_COL_DICT32) {
            off_a[i] = at;
            at = pad(at + cap * 4);
        } else {
            off_a[i] = at;  // payload
            at = pad(at + cap * 8);
            off_b[i] = at;  // tags
            at = pad(at + cap);
        }
    }
    return (at + 127) / 128 * 128;
}

}  // namespace

extern "C" {

n1k_status n1k_comm_unique_id(void* id) {
    return guarded(nullptr, [&]() -> n1k_status {
        if (!id) return N1K_INVALID;
        static_assert(sizeof(ncclUniqueId) == N1K_COMM_ID_BYTES, "N1K_COMM_ID_BYTES must be sizeof(ncclUniqueId)");
        ncclUniqueId u;
        NCCL_TRY(nullptr, ncclGetUniqueId(&u));
        memcpy(id, &u, sizeof u);
        return N1K_OK;
    });
}

n1k_status n1k_comm_create(const void* id, int rank, int world, int device, n1k_comm** out) {
    return guarded(nullptr, [&]() -> n1k_status {
        if (out) *out = nullptr;
        if (!id || !out || world < 1 || rank < 0 || rank >= world || world > (int)kMaxParts) return cfail(nullptr, N1K_INVALID, "bad communicator arguments");
        auto* c = new n1k_comm();
        c->rank = rank;
        c->world = world;
        c->device = device;
        auto bail = [&](n1k_status st) {
            delete c;
            return st;
        };
        if (hipSetDevice(device) != hipSuccess) return bail(cfail(nullptr, N1K_DEVICE_ERROR, "no HIP device %d", device));
        ncclUniqueId u;
        memcpy(&u, id, sizeof u);
        ncclResult_t r = ncclCommInitRank(&c->comm, world, u, rank);
        if (r != ncclSuccess) return bail(cfail(nullptr, N1K_DEVICE_ERROR, "ncclCommInitRank failed: %s", ncclGetErrorString(r)));
        if (hipEventCreateWithFlags(&c->ev, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_consumed, hipEventDisableTiming) != hipSuccess)
            return bail(cfail(nullptr, N1K_DEVICE_ERROR, "hipEventCreate failed"));
        if (c->scalar.ensure(2 * (size_t)kMaxParts) != hipSuccess) return bail(cfail(nullptr, N1K_OOM, "no device memory for the communicator"));  // (nothing of n1k_comm_max_u64 can fail before its collective)
        *out = c;
        return N1K_OK;
    });
}

n1k_status n1k_comm_create_loopback(int world, int device, n1k_comm** out) {
    return guarded(nullptr, [&]() -> n1k_status {
        if (!out || world < 1 || world > (int)kMaxParts) return cfail(nullptr, N1K_INVALID, "bad communicator arguments");
        if (hipSetDevice(device) != hipSuccess) return cfail(nullptr, N1K_DEVICE_ERROR, "no HIP device %d", device);
        auto* hub = new LoopHub();
        hub->world = world;
        hub->ptr.assign(world, nullptr);
        hub->stride.assign(world, 0);
        hub->val.assign(world, 0);
        hub->refs = world;
        for (int r = 0; r < world; r++) {
            auto* c = new n1k_comm();
            c->rank = r;
            c->world = world;
            c->device = device;
            c->hub = hub;
            (void)hipEventCreateWithFlags(&c->ev, hipEventDisableTiming);
            (void)hipEventCreateWithFlags(&c->ev_consumed, hipEventDisableTiming);
            out[r] = c;
        }
        return N1K_OK;
    });
}

void n1k_comm_destroy(n1k_comm* c) {
    if (!c) return;
    try {
        (void)hipSetDevice(c->device);
        if (c->hub) {
            bool last;
            {
                std::lock_guard<std::mutex> lk(c->hub->mu);
                last = --c->hub->refs == 0;
            }
            if (last) delete c->hub;
        }
        if (c->comm) (void)ncclCommDestroy(c->comm);
        if (c->ev) (void)hipEventDestroy(c->ev);
        if (c->ev_consumed) (void)hipEventDestroy(c->ev_consumed);
        c->void_send.release();
        c->void_recv.release();
        c->send.release();
        c->recv.release();
        c->gsend.release();
        c->grecv.release();
        c->scalar.release();
        delete c;
    } catch (...) {
    }
}

const char* n1k_comm_last_error(const n1k_comm* c) { return c ? c->last_error.c_str() : g_create_error.c_str(); }
int n1k_comm_rank(const n1k_comm* c) { return c ? c->rank : -1; }
int n1k_comm_world(const n1k_comm* c) { return c ? c->world : 0; }

n1k_status n1k_comm_max_u64(n1k_comm* c, n1k_handle* h, uint64_t value, uint64_t* out) {
    return guarded(h, [&]() -> n1k_status {
        if (!c || !h || !out) return N1K_INVALID;
        n1k_status st = ensure_device(h);
        if (st != N1K_OK) return st;
        if (c->hub) {  // loopback: values through the hub
            c->hub->val[c->rank] = value;
            c->hub->barrier();
            unsigned long long mx = 0;
            for (int p = 0; p < c->world; p++) mx = std::max(mx, c->hub->val[p]);
            c->hub->barrier();
            *out = mx;
            return N1K_OK;
        }
        unsigned long long v = value, m = 0;
        HIP_TRY(h, hipMemcpyAsync(c->scalar.p, &v, 8, hipMemcpyHostToDevice, h->stream));
        NCCL_TRY(c, ncclAllReduce(c->scalar.p, c->scalar.p + 1, 1, ncclUint64, ncclMax, c->comm, h->stream));
        HIP_TRY(h, hipMemcpyAsync(&m, c->scalar.p + 1, 8, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        *out = m;
        return N1K_OK;
    });
}

// ---- one exchange step, failure-safe -------------------------------------------------------------------------------------
//
// A collective has to be entered by every rank, whatever happened on that rank before it: a rank that returned early would
// leave its peers waiting in ncclRecv for good.  So the exchange calls are built in three parts:
//   1. the local part (validation, buffers, Filter + partition / export): anything may fail here;
//   2. the collective, ALWAYS entered — a rank whose local part failed ships regions that hold nothing but its status in
//      the verdict word (n1k_types.h VD_*), from its usual buffers or, when it has none, from one scratch region that every
//      peer gets; its own return value is its local status, and c->failure_broadcast says that the peers know;
//   3. the receiving part (verdict check, InitialGroup / merge on the owner): a failure here is this rank's alone and
//      travels in the gather (n1k_gather_groups_status).
// What cannot be carried: a rank with no device, a rank that cannot even allocate one region, and — for the row exchange
// only — a first batch whose column kinds are not the plan's (the region size is a function of them): see n1k.h.

namespace {

// behind the last reader of c->send / c->recv (the owner's scans of the previous exchange on this communicator): the next
// exchange overwrites both buffers on the sending stream
n1k_status wait_consumed(n1k_comm* c, n1k_handle* snd) {
    if (!c->consumed_pending) return N1K_OK;
    CHIP_TRY(c, hipStreamWaitEvent(snd->stream, c->ev_consumed, 0));
    return N1K_OK;
}
n1k_status mark_consumed(n1k_comm* c, n1k_handle* rcv) {
    CHIP_TRY(c, hipEventRecord(c->ev_consumed, rcv->stream));
    c->consumed_pending = true;
    return N1K_OK;
}

// the regions of a rank whose local part failed: headers that hold nothing but the status.  Its usual buffers when they are
// large enough (stride = region), else ONE scratch region that goes to every peer and one that every peer's region lands in.
n1k_status void_regions(n1k_comm* c, n1k_handle* snd, size_t region, size_t header_bytes, uint32_t nsend, uint32_t nrecv, n1k_status status,
                        const char** send, size_t* send_stride, char** recv, size_t* recv_stride, bool recv_contiguous) {
    if (c->send.n >= region * nsend && c->recv.n >= region * nrecv) {
        *send = c->send.p;
        *send_stride = region;
        *recv = c->recv.p;
        *recv_stride = region;
    } else {
        CHIP_TRY(c, c->void_send.ensure(region));
        CHIP_TRY(c, c->void_recv.ensure(recv_contiguous ? region * nrecv : region));
        *send = c->void_send.p;
        *send_stride = 0;
        *recv = c->void_recv.p;
        *recv_stride = recv_contiguous ? region : 0;
        nsend = 1;
    }
    for (uint32_t d = 0; d < nsend; d++) CHIP_TRY(c, hipMemsetAsync((char*)*send + (size_t)d * *send_stride, 0, header_bytes, snd->stream));
    CHIP_TRY(c, launch_stamp_verdict((unsigned long long*)*send, nsend, *send_stride / 8, nullptr, (uint32_t)status, snd->stream));
    return N1K_OK;
}

// test hook (option inject_failure, one shot): pretend that `site` failed on this handle
bool injected(n1k_handle* h, uint32_t site) {
    if (h->opt_inject_failure != site) return false;
    h->opt_inject_failure = 0;
    return true;
}

// n1k_exchange_partials; `failed`: what the caller's own preparation of this step returned (n1k_partials_step)
n1k_status exchange_partials_impl(n1k_comm* c, n1k_handle* sender, n1k_handle* receiver, uint64_t capacity_groups, int gathered, n1k_status failed) {
    c->failure_broadcast = false;
    sender->failure_global = false;
    n1k_status st = ensure_device(sender);
    if (st != N1K_OK) return st;  // (no device: this rank cannot take part in anything)
    st = wait_consumed(c, sender);
    if (st != N1K_OK) return fail(sender, st, "%s", c->last_error.c_str());
    const size_t region = (size_t)n1k_partial_region_bytes(sender, capacity_groups);  // (a function of the plan alone)
    const uint32_t P = (uint32_t)c->world, nsend = gathered ? 1u : P;
    // 1. local part
    n1k_status local = failed;
    if (local == N1K_OK && sender->stop_flag.load()) local = fail(sender, N1K_STOPPED, "operator was stopped");
    if (local == N1K_OK) {
        local = prepare_receiver(sender, receiver);
        if (local == N1K_INVALID && sender->last_error.empty()) fail(sender, local, "%s", receiver->last_error.c_str());
    }
    if (local == N1K_OK && injected(sender, 1)) local = fail(sender, N1K_OOM, "injected failure: buffers of the exchange");
    if (local == N1K_OK) {
        hipError_t e = c->send.ensure(region * nsend);
        if (e == hipSuccess) e = c->recv.ensure(region * (size_t)P);
        if (e != hipSuccess) local = fail(sender, e == hipErrorOutOfMemory ? N1K_OOM : N1K_DEVICE_ERROR, "buffers of the exchange: %s", hipGetErrorString(e));
    }
    if (local == N1K_OK) local = n1k_export_partials_async(sender, nsend, capacity_groups, c->send.p);
    if (local == N1K_OK && injected(sender, 2)) local = fail(sender, N1K_DEVICE_ERROR, "injected failure: export of the partial groups");
    // 2. the collective, always
    const char* send = c->send.p;
    char* recv = c->recv.p;
    size_t sstride = region, rstride = region;
    if (local != N1K_OK) {
        st = void_regions(c, sender, region, 16, nsend, P, local, &send, &sstride, &recv, &rstride, gathered != 0);
        if (st != N1K_OK) return local;  // (not even one region: the peers are not told — see the comment above)
    }
    const char* self = nullptr;
    if (gathered) st = all_gather_bytes(c, send, recv, region, sender->stream);
    else st = all_to_all_regions(c, send, sstride, recv, rstride, region, sender->stream, &self);
    if (local != N1K_OK) {
        c->failure_broadcast = st == N1K_OK;
        sender->failure_global = st == N1K_OK;
        return local;
    }
    if (st != N1K_OK) return fail(sender, st, "%s", c->last_error.c_str());
    // 3. the receiving part
    if (!gathered)  // (this rank's own region joins the received ones by a device copy of G groups, not through the fabric)
        HIP_TRY(sender, hipMemcpyAsync(c->recv.p + (size_t)c->rank * region, self, region, hipMemcpyDeviceToDevice, sender->stream));
    st = order_streams(c, sender, receiver);
    if (st != N1K_OK) return fail(sender, st, "%s", c->last_error.c_str());
    if (injected(sender, 3)) return fail(sender, N1K_DEVICE_ERROR, "injected failure: the receiving part");
    st = n1k_merge_partials_device(receiver, P, capacity_groups, c->recv.p);
    (void)mark_consumed(c, receiver);
    if (st != N1K_OK) return fail(sender, st, "receiver: %s", receiver->last_error.c_str());
    return N1K_OK;
}

// the column kinds of a batch are the plan's (and the ones seen before): what the size of a row region hangs on
bool batch_kinds_ok(const n1k_handle* h, const n1k_batch* b) {
    if (!b || b->ncols != (uint32_t)h->plan.paths.size() || (b->ncols && !b->cols)) return false;
    for (uint32_t i = 0; i < b->ncols; i++) {
        if (b->cols[i].kind != N1K_COL_DICT32 && b->cols[i].kind != N1K_COL_TAGGED64) return false;
        if (h->layout_fixed && b->cols[i].kind != h->col_kinds[i]) return false;
    }
    return true;
}

n1k_status exchange_rows_impl(n1k_comm* c, n1k_handle* sender, const n1k_batch* batch, n1k_handle* receiver, const uint64_t* capacity_rows,
                              n1k_status failed) {
    c->failure_broadcast = false;
    sender->failure_global = false;
    if (!sender->plan.has_group) return fail(sender, N1K_INVALID, "the row exchange partitions on group keys");
    n1k_status st = ensure_device(sender);
    if (st != N1K_OK) return st;  // (no device: this rank cannot take part in anything)
    // 0. the size of a region: from the column kinds (of this batch, or as fixed by an earlier one)
    n1k_status local = failed;
    if (!batch_kinds_ok(sender, batch)) {
        if (!sender->layout_fixed)
            return fail(sender, N1K_INVALID, "the first batch of a row exchange does not have the plan's columns: the region size is unknown "
                                             "and this rank cannot enter the collective");
        if (local == N1K_OK) local = fail(sender, N1K_INVALID, "the batch does not have the columns of the earlier ones");
    } else if (!sender->layout_fixed) {
        st = fix_layout(sender, batch);  // (sets the column kinds first, then the key layout)
        if (st != N1K_OK && local == N1K_OK) local = st;
    }
    // Every destination has its own capacity (the same vector on every rank): the regions for owner d — one per sender — are
    // sized for what THAT owner receives, so a hot owner (skewed keys) does not inflate the regions of the other P - 1.
    const uint32_t P = (uint32_t)c->world;
    const uint64_t quantum = 16ull * kRowSubs;  // kRowSubs sub-regions of whole 16-row groups
    std::vector<uint64_t> cap(P);
    uint64_t cap_max = 0;
    for (uint32_t d = 0; d < P; d++) {
        if (capacity_rows[d] == 0) return fail(sender, N1K_INVALID, "a region capacity of zero rows");  // (the same on every rank)
        cap[d] = (capacity_rows[d] + quantum - 1) / quantum * quantum;
        cap_max = std::max(cap_max, cap[d]);
    }
    if (cap_max >= (1ull << 31)) return fail(sender, N1K_INVALID, "row regions hold fewer than 2^31 rows");  // (the same on every rank)
    std::vector<size_t> off_a, off_b, wire(P);
    const size_t stride = row_region_layout(sender, cap_max, off_a, off_b);  // distance between the regions in the send buffer
    for (uint32_t d = 0; d < P; d++) wire[d] = row_region_layout(sender, cap[d], off_a, off_b);
    const uint64_t cap_me = cap[(uint32_t)c->rank];
    const size_t region = row_region_layout(sender, cap_me, off_a, off_b);  // what this rank receives from every sender (off_*: its layout)
    const size_t header = (size_t)kRowSubs * kCursorStride * 8;
    st = wait_consumed(c, sender);
    if (st != N1K_OK && local == N1K_OK) local = fail(sender, st, "%s", c->last_error.c_str());
    // 1. local part: Filter + hash partition on the group key values into the packed regions (headers zeroed first)
    auto local_part = [&]() -> n1k_status {
        if (sender->stop_flag.load()) return fail(sender, N1K_STOPPED, "operator was stopped");
        n1k_status s = validate_batch(sender, batch);
        if (s != N1K_OK) return s;
        s = prepare_receiver(sender, receiver);
        if (s != N1K_OK) return s;
        if (injected(sender, 1)) return fail(sender, N1K_OOM, "injected failure: buffers of the exchange");
        HIP_TRY(sender, c->send.ensure(stride * P));
        HIP_TRY(sender, c->recv.ensure(region * P));
        for (uint32_t d = 0; d < P; d++) HIP_TRY(sender, hipMemsetAsync(c->send.p + (size_t)d * stride, 0, header, sender->stream));
        s = bind_columns(sender, batch, true);
        if (s != N1K_OK) return s;
        s = ensure_rank(sender);
        if (s != N1K_OK) return s;
        PartArgs A{};
        A.nrows = batch->nrows;
        A.capacity = cap_max;
        A.nparts = P;
        A.ncopy = (uint32_t)sender->plan.paths.size();
        A.counts = (unsigned long long*)c->send.p;
        A.count_stride = (uint32_t)(stride / 8);
        A.region_bytes = stride;
        A.sub_rows = cap_max / kRowSubs;
        bool uniform = true;
        for (uint32_t d = 0; d < P; d++) uniform &= cap[d] == cap_max;
        A.hdr_bytes = (uint32_t)header;
        if (uniform) {
            // every destination alike (uniform keys; one rank): the kernels' plain addressing — one layout, region d a whole number
            // of strides further on — which spares them the per-destination tables (0.43 against 0.46 ms per 100 M rows)
            std::vector<size_t> oa, ob;
            (void)row_region_layout(sender, cap_max, oa, ob);
            for (uint32_t i = 0; i < A.ncopy; i++) {
                if (sender->col_kinds[i] == N1K_COL_DICT32) A.out_codes[i] = (uint32_t*)(c->send.p + oa[i]);
                else {
                    A.out_payload[i] = (uint64_t*)(c->send.p + oa[i]);
                    A.out_tags[i] = (uint8_t*)(c->send.p + ob[i]);
                }
            }
        } else {
            A.per_dest = 1;
            for (uint32_t d = 0; d < P; d++) A.dest_cap[d] = (uint32_t)cap[d];
        }
        A.err_flags = sender->d_errp;
        s = run_partition(sender, batch, A);
        if (s != N1K_OK) return s;
        if (injected(sender, 2)) return fail(sender, N1K_DEVICE_ERROR, "injected failure: the partition");
        sender->stats.rows_in += batch->nrows;
        sender->stats.batches += 1;
        // what the partition itself found (rows whose key does not pack, values the Filter cannot order) joins the verdicts
        HIP_TRY(sender, launch_stamp_verdict((unsigned long long*)c->send.p, P, stride / 8, sender->d_errp, 0, sender->stream));
        return N1K_OK;
    };
    if (local == N1K_OK) local = local_part();
    c->sent_stride = stride;
    c->sent_cap = cap;
    // 2. ONE all-to-all, always: counts, verdicts and rows of every column travel in the same region
    const char* send = c->send.p;
    char* recv = c->recv.p;
    size_t sstride = stride, rstride = region;
    if (local != N1K_OK) {
        c->sent_cap.clear();
        if (c->send.n >= stride * P && c->recv.n >= region * P) {
            for (uint32_t d = 0; d < P; d++) (void)hipMemsetAsync(c->send.p + (size_t)d * stride, 0, header, sender->stream);
            (void)launch_stamp_verdict((unsigned long long*)c->send.p, P, stride / 8, nullptr, (uint32_t)local, sender->stream);
        } else {
            st = void_regions(c, sender, stride, header, 1, 1, local, &send, &sstride, &recv, &rstride, false);
            if (st != N1K_OK) return local;  // (not even one region: the peers are not told — see the comment above)
        }
    }
    const char* self = nullptr;
    st = all_to_all_regions(c, send, sstride, recv, rstride, region, sender->stream, &self, wire.data());
    if (local != N1K_OK) {
        c->failure_broadcast = st == N1K_OK;
        sender->failure_global = st == N1K_OK;
        return local;
    }
    if (st != N1K_OK) return fail(sender, st, "%s", c->last_error.c_str());
    st = order_streams(c, sender, receiver);
    if (st != N1K_OK) return fail(sender, st, "%s", c->last_error.c_str());
    // 3. the owner's InitialGroup over what it received: one batch per source, each with its row count on the device.
    //    (Headers are checked first: a sender that overflowed, dropped rows or failed voids the step on every rank.)
    std::vector<const char*> src(P);
    for (uint32_t sidx = 0; sidx < P; sidx++) src[sidx] = (int)sidx == c->rank ? self : c->recv.p + (size_t)sidx * region;
    {
        HeaderList H{};
        for (uint32_t sidx = 0; sidx < P; sidx++) H.h[sidx] = (unsigned long long*)src[sidx];
        HIP_TRY(receiver, launch_exchange_verdict(H, P, receiver->d_errp, receiver->stream));
    }
    if (injected(sender, 3)) return fail(sender, N1K_DEVICE_ERROR, "injected failure: the receiving part");
    for (uint32_t sidx = 0; sidx < P; sidx++) {
        const uint32_t rnc = (uint32_t)receiver->plan.paths.size();
        std::vector<n1k_col> cols(std::max<size_t>(1, rnc));
        for (uint32_t i = 0; i < rnc; i++) {
            const int j = sender_column(sender, receiver->plan.paths[i]);  // (prepare_receiver checked that it exists)
            cols[i].kind = sender->col_kinds[j];
            if (cols[i].kind == N1K_COL_DICT32) cols[i].codes = (const uint32_t*)(src[sidx] + off_a[j]);
            else {
                cols[i].payload = (const uint64_t*)(src[sidx] + off_a[j]);
                cols[i].tags = (const uint8_t*)(src[sidx] + off_b[j]);
            }
        }
        n1k_batch rb{};
        rb.nrows = cap_me;
        rb.ncols = rnc;
        rb.cols = cols.data();
        receiver->push_seg_counts = (const unsigned long long*)src[sidx];  // (the region's header)
        receiver->push_nseg = kRowSubs;
        receiver->push_seg_rows = cap_me / kRowSubs;
        st = push_device(receiver, &rb);
        receiver->push_seg_counts = nullptr;
        receiver->push_nseg = 0;
        if (st != N1K_OK) {
            (void)mark_consumed(c, receiver);
            return fail(sender, st, "receiver: %s", receiver->last_error.c_str());
        }
    }
    (void)mark_consumed(c, receiver);
    return N1K_OK;
}

}  // namespace

n1k_status n1k_comm_max_u64_v(n1k_comm* c, n1k_handle* h, uint32_t n, const uint64_t* values, uint64_t* out) {
    return guarded(h, [&]() -> n1k_status {
        if (!c || !h || !values || !out || n == 0 || n > kMaxParts) return N1K_INVALID;
        n1k_status st = ensure_device(h);
        if (st != N1K_OK) return st;
        if (c->hub) {  // loopback: values through the hub, one at a time
            for (uint32_t i = 0; i < n; i++) {
                c->hub->val[c->rank] = values[i];
                c->hub->barrier();
                unsigned long long mx = 0;
                for (int p = 0; p < c->world; p++) mx = std::max(mx, c->hub->val[p]);
                c->hub->barrier();
                out[i] = mx;
            }
            return N1K_OK;
        }
        HIP_TRY(h, c->scalar.ensure(2 * (size_t)kMaxParts));
        std::vector<unsigned long long> m(n);
        HIP_TRY(h, hipMemcpyAsync(c->scalar.p, values, (size_t)n * 8, hipMemcpyHostToDevice, h->stream));
        NCCL_TRY(c, ncclAllReduce(c->scalar.p, c->scalar.p + kMaxParts, n, ncclUint64, ncclMax, c->comm, h->stream));
        HIP_TRY(h, hipMemcpyAsync(m.data(), c->scalar.p + kMaxParts, (size_t)n * 8, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        for (uint32_t i = 0; i < n; i++) out[i] = m[i];
        return N1K_OK;
    });
}

n1k_status n1k_exchange_partials(n1k_comm* c, n1k_handle* sender, n1k_handle* receiver, uint64_t capacity_groups, int gathered) {
    return guarded(sender, [&]() -> n1k_status {
        if (!c || !sender || !receiver || capacity_groups == 0) return N1K_INVALID;
        return exchange_partials_impl(c, sender, receiver, capacity_groups, gathered, N1K_OK);
    });
}

n1k_status n1k_exchange_rows_v(n1k_comm* c, n1k_handle* sender, const n1k_batch* batch, n1k_handle* receiver, const uint64_t* capacity_rows) {
    return guarded(sender, [&]() -> n1k_status {
        if (!c || !sender || !receiver || !capacity_rows) return N1K_INVALID;
        return exchange_rows_impl(c, sender, batch, receiver, capacity_rows, N1K_OK);
    });
}

n1k_status n1k_exchange_rows(n1k_comm* c, n1k_handle* sender, const n1k_batch* batch, n1k_handle* receiver, uint64_t capacity_rows) {
    if (!c || capacity_rows == 0) return N1K_INVALID;
    const std::vector<uint64_t> caps((size_t)c->world, capacity_rows);
    return n1k_exchange_rows_v(c, sender, batch, receiver, caps.data());
}

// rows this rank's last n1k_exchange_rows wrote for every destination (its own included): what the ranks size the regions of
// the following steps from (the largest over the senders, per destination: n1k_comm_max_u64_v)
n1k_status n1k_exchange_sent_rows(n1k_comm* c, n1k_handle* sender, uint64_t* out) {
    return guarded(sender, [&]() -> n1k_status {
        if (!c || !sender || !out) return N1K_INVALID;
        const uint32_t P = (uint32_t)c->world;
        if (c->sent_cap.size() != P || !c->send.p) return fail(sender, N1K_INVALID, "no completed row exchange on this communicator");
        std::vector<unsigned long long> hdr((size_t)P * kRowSubs * kCursorStride);
        for (uint32_t d = 0; d < P; d++)
            HIP_TRY(sender, hipMemcpyAsync(hdr.data() + (size_t)d * kRowSubs * kCursorStride, c->send.p + (size_t)d * c->sent_stride,
                                           (size_t)kRowSubs * kCursorStride * 8, hipMemcpyDeviceToHost, sender->stream));
        HIP_TRY(sender, hipStreamSynchronize(sender->stream));
        for (uint32_t d = 0; d < P; d++) {
            uint64_t n = 0;
            for (uint32_t x = 0; x < kRowSubs; x++)
                n += std::min<uint64_t>(hdr[((size_t)d * kRowSubs + x) * kCursorStride], c->sent_cap[d] / kRowSubs);
            out[d] = n;
        }
        return N1K_OK;
    });
}


int n1k_failure_is_global(const n1k_handle* h) { return h && h->failure_global ? 1 : 0; }

// what follows the exchange in one step: the owner's n1k_finish, then — unless the step failed on every rank alike — the
// gather, entered with this rank's own status
static n1k_status finish_and_gather(n1k_comm* c, n1k_status st, n1k_handle* receiver, n1k_handle* merger, bool gather, n1k_result* out,
                                    int* worst_status) {
    if (st != N1K_OK && c->failure_broadcast) return st;  // this rank's own failure, told to every peer in the headers: no gather anywhere
    n1k_result local;
    // (the step's result leaves the device here and the next step starts with a reset: a small table's tail kernel leaves the
    //  receiver as that reset would — no reopen kernel in front of the next step)
    receiver->clear_on_finish = true;
    const n1k_status fs = n1k_finish(receiver, &local);  // (also after a failure of the receiving part: the verdicts decide)
    receiver->clear_on_finish = false;
    if (fs != N1K_OK && receiver->failure_global) return fs;  // learnt from the headers, by every rank alike: no gather anywhere
    if (st == N1K_OK) st = fs;
    if (!gather) {
        if (st == N1K_OK) *out = local;
        return st;
    }
    const n1k_status gs = n1k_gather_groups_status(c, merger, st == N1K_OK ? &local : nullptr, (int)st, out, worst_status);
    return gs != N1K_OK ? gs : st;
}

n1k_status n1k_rows_step(n1k_comm* c, n1k_handle* sender, const n1k_batch* batch, n1k_handle* receiver, n1k_handle* merger,
                         uint64_t capacity_rows, n1k_result* out, int* worst_status) {
    if (!c || capacity_rows == 0) return N1K_INVALID;
    const std::vector<uint64_t> caps((size_t)c->world, capacity_rows);
    return n1k_rows_step_v(c, sender, batch, receiver, merger, caps.data(), out, worst_status);
}

n1k_status n1k_rows_step_v(n1k_comm* c, n1k_handle* sender, const n1k_batch* batch, n1k_handle* receiver, n1k_handle* merger,
                           const uint64_t* capacity_rows, n1k_result* out, int* worst_status) {
    if (!c || !sender || !batch || !receiver || !merger || !out || !worst_status || !capacity_rows) return N1K_INVALID;
    *worst_status = N1K_OK;
    n1k_status st = n1k_reset(receiver);
    if (st == N1K_OK) st = n1k_reset(sender);  // (the sender holds no groups in this mode; its counters and timers start over)
    // (a reset that failed is this rank's own failure: it still enters the exchange, with that status)
    const n1k_status prep = st;
    st = guarded(sender, [&]() -> n1k_status { return exchange_rows_impl(c, sender, batch, receiver, capacity_rows, prep); });
    return finish_and_gather(c, st, receiver, merger, true, out, worst_status);
}

n1k_status n1k_partials_step(n1k_comm* c, n1k_handle* sender, const n1k_batch* batch, n1k_handle* receiver, n1k_handle* merger,
                             uint64_t capacity_groups, int gathered, n1k_result* out, int* worst_status) {
    if (!c || !sender || !batch || !receiver || !merger || !out || !worst_status || capacity_groups == 0) return N1K_INVALID;
    *worst_status = N1K_OK;
    n1k_status st = n1k_reset(receiver);
    if (st == N1K_OK) st = n1k_reset(sender);
    if (st == N1K_OK) st = n1k_push_device_batch(sender, batch);  // InitialGroup over the shard: the single-GPU kernels
    const n1k_status prep = st;
    st = guarded(sender, [&]() -> n1k_status { return exchange_partials_impl(c, sender, receiver, capacity_groups, gathered, prep); });
    // gathered: every rank merged every rank's groups and `receiver` (the handle that carries the plan's tail) holds the result
    return finish_and_gather(c, st, receiver, merger, !gathered, out, worst_status);
}


n1k_status n1k_gather_groups(n1k_comm* c, n1k_handle* h, const n1k_result* local, n1k_result* out) {
    int worst = 0;
    n1k_status st = n1k_gather_groups_status(c, h, local, N1K_OK, out, &worst);
    if (st == N1K_OK && worst != N1K_OK) return fail(h, (n1k_status)worst, "a peer rank's step failed with status %d", worst);
    return st;
}

n1k_status n1k_gather_groups_status(n1k_comm* c, n1k_handle* h, const n1k_result* local, int local_status, n1k_result* out, int* worst_status) {
    return guarded(h, [&]() -> n1k_status {
        if (!c || !h || !out || !worst_status || (!local && local_status == N1K_OK)) return N1K_INVALID;
        *worst_status = local_status;
        static const n1k_result kNone{};
        if (!local || local_status != N1K_OK) local = &kNone;  // a rank whose step failed contributes no groups, only its status
        if (h->has_array_agg) return fail(h, N1K_UNSUPPORTED, "array_agg values are interned per rank: gather the rows on the host");
        n1k_status st = ensure_device(h);
        if (st != N1K_OK) return st;
        const size_t nk = h->plan.keys.size(), na = h->plan.aggs.size();
        const size_t rec = (nk + na) * sizeof(n1k_value);
        // ONE all-gather of fixed-size slots [count][records]: the slot size is part of the collective's shape, so it is the
        // same on every rank by construction — it starts at 1024 records and only ever changes on what ALL ranks read in
        // the gathered headers (a count beyond the slot: everybody doubles to fit the largest and gathers again)
        unsigned long long mine = local->ngroups;
        std::vector<char> stage;
        size_t slot = 0;
        for (;;) {
            const uint64_t capg = c->gather_cap;
            slot = 16 + (size_t)capg * rec;
            stage.assign(slot, 0);
            memcpy(stage.data(), &mine, 8);
            const unsigned long long my_status = (unsigned long long)(unsigned)local_status;
            memcpy(stage.data() + 8, &my_status, 8);
            for (uint64_t g = 0; g < std::min<uint64_t>(local->ngroups, capg); g++) {
                char* p = stage.data() + 16 + (size_t)g * rec;
                if (nk) memcpy(p, local->keys + g * nk, nk * sizeof(n1k_value));
                if (na) memcpy(p + nk * sizeof(n1k_value), local->aggs + g * na, na * sizeof(n1k_value));
            }
            HIP_TRY(h, c->gsend.ensure(slot));
            HIP_TRY(h, c->grecv.ensure(slot * (size_t)c->world));
            HIP_TRY(h, hipMemcpyAsync(c->gsend.p, stage.data(), slot, hipMemcpyHostToDevice, h->stream));
            st = all_gather_bytes(c, c->gsend.p, c->grecv.p, slot, h->stream);
            if (st != N1K_OK) return fail(h, st, "%s", c->last_error.c_str());
            c->ghost.resize(slot * (size_t)c->world);
            HIP_TRY(h, hipMemcpyAsync(c->ghost.data(), c->grecv.p, c->ghost.size(), hipMemcpyDeviceToHost, h->stream));
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            unsigned long long most = 0, bad = 0;
            for (int r = 0; r < c->world; r++) {
                unsigned long long n = 0, s = 0;
                memcpy(&n, c->ghost.data() + (size_t)r * slot, 8);
                memcpy(&s, c->ghost.data() + (size_t)r * slot + 8, 8);
                most = std::max(most, n);
                if (s && !bad) bad = s;  // (the lowest rank's failure: the same on every rank)
            }
            if (bad) {  // some rank's step failed: every rank learns it here, in the collective it would otherwise hang in
                *worst_status = (int)bad;
                memset(out, 0, sizeof *out);
                return N1K_OK;
            }
            if (most <= capg) break;
            while (c->gather_cap < most) c->gather_cap *= 2;
        }
        // 3. the union, in rank order; the plan's grouped tail (ORDER BY / OFFSET / LIMIT, projection) over it
        c->gkeys.clear();
        c->gaggs.clear();
        for (int r = 0; r < c->world; r++) {
            const char* base = c->ghost.data() + (size_t)r * slot;
            unsigned long long n = 0;
            memcpy(&n, base, 8);
            for (unsigned long long g = 0; g < n; g++) {
                const n1k_value* v = (const n1k_value*)(base + 16 + (size_t)g * rec);
                c->gkeys.insert(c->gkeys.end(), v, v + nk);
                c->gaggs.insert(c->gaggs.end(), v + nk, v + nk + na);
            }
        }
        const uint64_t total = nk ? c->gkeys.size() / nk : (na ? c->gaggs.size() / na : 0);
        return n1k_order_rows(h, total, c->gkeys.data(), c->gaggs.data(), out);
    });
}

}  // extern "C"
