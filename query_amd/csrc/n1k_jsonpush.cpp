// n1k_jsonpush.cpp — n1k_push_json through the device extractor (n1k_jsondev.hip): the documents' bytes cross PCIe once, the
// leaf columns are born in HBM, only the batch's DISTINCT strings and the documents the kernel left alone visit the host.
#include "n1k_engine.h"

using namespace n1k;
using namespace n1k_eng;

namespace n1k_eng {

// can the device extractor serve this handle's leaf paths?  (chains of at most kJsonMaxSteps field names)
bool json_device_paths(n1k_handle* h, JsonDevArgs& A) {
    const size_t np = h->json_paths.size();
    if (np == 0 || np > (size_t)kMaxCols) return false;
    memset(&A, 0, sizeof A);
    A.npaths = (uint32_t)np;
    uint32_t at = 0;
    for (size_t c = 0; c < np; c++) {
        const JsonPath& p = h->json_paths[c];
        if (p.names.empty() || p.names.size() > kJsonMaxSteps) return false;
        A.paths[c].nsteps = (uint32_t)p.names.size();
        for (size_t s = 0; s < p.names.size(); s++) {
            if (p.names[s].is_index) return false;  // (element navigation: the host's extractor)
            const std::string& nm = p.names[s].name;
            if (at + nm.size() > sizeof A.names) return false;
            A.paths[c].name_off[s] = at;
            A.paths[c].name_len[s] = (uint32_t)nm.size();
            memcpy(A.names + at, nm.data(), nm.size());
            at += (uint32_t)nm.size();
        }
    }
    return true;
}

// *done = false: nothing was pushed (the caller takes the host path); else the batch is in the handle
n1k_status push_json_device(n1k_handle* h, uint64_t ndocs, const uint64_t* offsets, const char* bytes, bool* done) {
    *done = false;
    JsonDevArgs A;
    if (!h->opt_json_device || ndocs < h->opt_json_device_min_docs || !json_device_paths(h, A)) return N1K_OK;
    if (offsets[ndocs] < offsets[0]) return N1K_OK;
    const uint64_t nbytes = offsets[ndocs] - offsets[0];
    if (nbytes >= (1ull << 39)) return N1K_OK;
    n1k_status st = ensure_device(h);
    if (st != N1K_OK) return st;
    const size_t np = A.npaths;
    // the previous batch's kernels may still read the column buffers of this path: one set, reused behind a wait
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    constexpr uint32_t kTabBits = 20;
    HIP_TRY(h, h->jd_bytes.ensure(nbytes + 64));
    HIP_TRY(h, h->jd_offsets.ensure(ndocs + 1));
    HIP_TRY(h, h->jd_status.ensure(ndocs));
    h->jd_tags.resize(std::max(h->jd_tags.size(), np));
    h->jd_payload.resize(std::max(h->jd_payload.size(), np));
    for (size_t c = 0; c < np; c++) {
        HIP_TRY(h, h->jd_tags[c].ensure(ndocs));
        HIP_TRY(h, h->jd_payload[c].ensure(ndocs));
    }
    HIP_TRY(h, h->jd_tab.ensure(2ull << kTabBits));         // hashes, then first occurrences
    HIP_TRY(h, h->jd_new_list.ensure((1ull << kTabBits) + 4));  // [0 .. cap) slots, then the counter (8 bytes, aligned)
    HIP_TRY(h, h->jd_code_of.ensure(1ull << kTabBits));
    HIP_TRY(h, hipMemcpyAsync(h->jd_bytes.p, bytes + offsets[0], nbytes, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->jd_offsets.p, offsets, (ndocs + 1) * 8, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemsetAsync(h->jd_tab.p, 0, (2ull << kTabBits) * 8, h->stream));
    unsigned long long* d_new_count = (unsigned long long*)(h->jd_new_list.p + (((1ull << kTabBits) + 1) & ~1ull));
    HIP_TRY(h, hipMemsetAsync(d_new_count, 0, 8, h->stream));
    A.bytes = (const uint8_t*)h->jd_bytes.p;
    A.offsets = h->jd_offsets.p;
    A.base = offsets[0];
    A.ndocs = ndocs;
    A.tab_bits = kTabBits;
    for (size_t c = 0; c < np; c++) {
        A.out_tags[c] = h->jd_tags[c].p;
        A.out_payload[c] = h->jd_payload[c].p;
    }
    A.status = h->jd_status.p;
    A.tab_hash = (unsigned long long*)h->jd_tab.p;
    A.tab_first = (unsigned long long*)h->jd_tab.p + (1ull << kTabBits);
    A.new_list = h->jd_new_list.p;
    A.new_count = d_new_count;
    A.new_cap = 1u << kTabBits;
    HIP_TRY(h, launch_json_extract(A, (uint32_t)h->num_cus, h->stream));
    // what comes back: the documents the kernel left alone, and the batch's new strings
    h->jd_host_status.resize(ndocs);
    st = ensure_pinned_counters(h);
    if (st != N1K_OK) return st;
    HIP_TRY(h, hipMemcpyAsync(h->jd_host_status.data(), h->jd_status.p, ndocs, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->pin_counters + kCounters, d_new_count, 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const uint64_t nnew = h->pin_counters[kCounters];
    std::vector<uint64_t> left;
    for (uint64_t d = 0; d < ndocs; d++)
        if (h->jd_host_status[d]) left.push_back(d);
    if (nnew > A.new_cap / 2 || left.size() > ndocs * h->opt_json_device_left_pct / 100 + 16) return N1K_OK;  // not this kernel's kind of batch: the host path takes all of it
    // new strings -> dictionary codes -> code_of[slot]
    if (nnew) {
        HIP_TRY(h, h->jd_new_first.ensure(nnew));
        HIP_TRY(h, h->jd_codes.ensure(nnew));
        HIP_TRY(h, launch_json_gather_first(A, nnew, (unsigned long long*)h->jd_new_first.p, h->stream));
        std::vector<uint64_t> first(nnew);
        HIP_TRY(h, hipMemcpyAsync(first.data(), h->jd_new_first.p, nnew * 8, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        std::vector<uint32_t> codes(nnew);
        const char* base = bytes + offsets[0];
        for (uint64_t i = 0; i < nnew; i++) {
            const uint64_t off = first[i] >> 24, len = first[i] & 0xFFFFFFull;
            if (off + len > nbytes) return fail(h, N1K_DEVICE_ERROR, "device JSON extractor: a string lies outside the batch");
            codes[i] = intern(h, std::string(base + off, (size_t)len));
        }
        HIP_TRY(h, hipMemcpyAsync(h->jd_codes.p, codes.data(), nnew * 4, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, launch_json_scatter_codes(A.new_list, h->jd_codes.p, nnew, h->jd_code_of.p, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));  // (`codes` lives on this frame)
    }
    HIP_TRY(h, launch_json_remap(A, h->jd_code_of.p, h->stream));
    // the documents left to the host: its scalar extractor gives the values, or names the malformed one
    if (!left.empty()) {
        std::vector<uint8_t> ptags(left.size() * np);
        std::vector<uint64_t> ppay(left.size() * np);
        JsonColumns one;
        std::string err;
        for (size_t i = 0; i < left.size(); i++) {
            const uint64_t d = left[i];
            const long long bad = extract_json_range(h->json_paths, offsets, bytes, d, d + 1, one, err);
            if (bad >= 0 && err.find("nested deeper than 256") != std::string::npos)
                return fail(h, N1K_UNSUPPORTED_DATA, "document %llu: %s", (unsigned long long)d, err.c_str());
            if (bad >= 0) return fail(h, N1K_INVALID, "document %llu is not valid JSON: %s", (unsigned long long)d, err.c_str());
            for (size_t c = 0; c < np; c++) {
                const uint8_t t = one.tags[c][0];
                uint64_t v = one.payload[c][0];
                if (t >= N1K_T_STRING) v = intern(h, one.strings[(size_t)v]);
                ptags[i * np + c] = t;
                ppay[i * np + c] = v;
            }
        }
        HIP_TRY(h, h->jd_patch_docs.ensure(left.size()));
        HIP_TRY(h, h->jd_patch_tags.ensure(ptags.size()));
        HIP_TRY(h, h->jd_patch_pay.ensure(ppay.size()));
        HIP_TRY(h, hipMemcpyAsync(h->jd_patch_docs.p, left.data(), left.size() * 8, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipMemcpyAsync(h->jd_patch_tags.p, ptags.data(), ptags.size(), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, hipMemcpyAsync(h->jd_patch_pay.p, ppay.data(), ppay.size() * 8, hipMemcpyHostToDevice, h->stream));
        HIP_TRY(h, launch_json_patch(A, h->jd_patch_docs.p, h->jd_patch_tags.p, h->jd_patch_pay.p, left.size(), h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));  // (the vectors live on this frame)
    }
    if (h->dict.size() >= 0xFFFFFFF0ull) return fail(h, N1K_OOM, "dictionary too large");
    // the columns are where the scan kernels read them
    std::vector<n1k_col> cols(np);
    for (size_t c = 0; c < np; c++) {
        cols[c].kind = N1K_COL_TAGGED64;
        cols[c].tags = h->jd_tags[c].p;
        cols[c].payload = h->jd_payload[c].p;
    }
    n1k_batch b{};
    b.nrows = ndocs;
    b.ncols = (uint32_t)np;
    b.cols = cols.data();
    st = push_device(h, &b);
    if (st != N1K_OK) return st;
    h->stats.json_device_docs += ndocs - left.size();
    *done = true;
    return N1K_OK;
}

}  // namespace n1k_eng
