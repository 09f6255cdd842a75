// n1k_kernels.h — launch interface between the host engine and n1k_kernels.hip
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include "n1k_types.h"

namespace n1k {

constexpr int kFilterStreamTile = 8192;  // rows per tile of the one-pass Filter-only kernel (4 passes of 256 threads x 8 rows)
constexpr int kFilterTile = 4096;  // rows per tile of the Filter-only path (64 ballot words)

struct OutValue {  // same 16-byte layout as n1k_value
    uint64_t tag;  // low byte = tag (little endian), rest zero
    uint64_t payload;
};

struct OutPartial {
    int64_t count;
    int64_t isum;
    double fsum;
    uint32_t flags;  // bit0 int_exact, bit1 has_float
    uint32_t ext_tag;
    uint64_t ext_payload;
    int64_t distinct;
};

struct SynthArgs {
    uint64_t seed, first_row, nrows, total_rows;
    uint32_t k_cat, pad;
    const double* cat_cdf;
    uint32_t* cat_codes;
    uint8_t* price_tags;
    uint64_t* price_payload;
    uint8_t* user_tags;
    uint64_t* user_payload;
    uint8_t* region_tags;
    uint64_t* region_payload;
};

hipError_t launch_init_table(const Program& P, const GlobalTable& G, uint64_t first, uint64_t count,
                             unsigned long long* counters, hipStream_t st);
hipError_t launch_restamp_ranks(const Program& P, const GlobalTable& G, hipStream_t st);
hipError_t launch_rehash(const Program& P, const GlobalTable& oldt, const GlobalTable& newt, uint32_t* err_flags,
                         unsigned long long* ngroups_scratch, hipStream_t st);
hipError_t launch_scan_group(const Program& P, const ScanArgs& A, const GlobalTable& G, unsigned long long* ngroups,
                             uint32_t grid, uint32_t block, uint32_t rows_per_lane, bool direct, hipStream_t st);
hipError_t launch_scan_fast(const Program& P, const FastArgs& F, const GlobalTable& G, unsigned long long* ngroups,
                            uint32_t grid, uint32_t block, uint32_t rows_per_lane, hipStream_t st);
// (tail: the query's last kernel — launch_finalize_small's work — done by the merge's last workgroup; null: merge only)
struct TailArgs {
    OutValue *out_keys, *out_aggs;
    OutPartial* out_parts;
    uint64_t* out_rep;
    unsigned long long *counters, *host_counters;
    uint64_t max_out;
    unsigned int* done;  // workgroups finished (zero before the launch; the last workgroup leaves it zero)
    uint32_t clear, enabled;
};
hipError_t launch_merge_slabs(const Program& P, const FastArgs& F, const GlobalTable& G, uint32_t nblocks,
                              unsigned long long* ngroups, hipStream_t st, uint32_t ychunks_opt = 0, const TailArgs* tail = nullptr);
// ORDER BY ... LIMIT over the finalised groups: order images of the first sort term, radix select of the keep-th image,
// candidate indices (image <= threshold) and compaction of their records
// high-cardinality GROUP BY: rows -> records (key + operands), [radix passes], per-bin LDS aggregation
hipError_t launch_finalize_region(const Program& P, const uint64_t* region, uint64_t cap, uint64_t count, OutValue* out_keys,
                                  OutValue* out_aggs, OutPartial* out_parts, uint64_t* out_rep, uint32_t* err_flags, hipStream_t st,
                                  const uint32_t* cand = nullptr, OutValue* ord = nullptr, bool ord_is_key = false, uint32_t ord_index = 0,
                                  uint64_t* images = nullptr, bool desc = false);  // (images: order images of the ORDER BY term instead of ord)
hipError_t launch_add_counter(unsigned long long* p, unsigned long long v, hipStream_t st);
hipError_t launch_probe_keys(const Program& P, uint64_t nrows, const GlobalTable& G, uint32_t* err_flags, unsigned long long* ngroups,
                             uint32_t grid, hipStream_t st);
hipError_t launch_project_records(const Program& P, const ProjectArgs& A, uint32_t grid, hipStream_t st);
hipError_t launch_agg_bins(const Program& P, const BinAggArgs& A, const GlobalTable& G, unsigned long long* ngroups, uint32_t grid,
                           hipStream_t st);
size_t topk_state_bytes();
size_t topk_ncand_offset();
hipError_t launch_topk_select(const Program& P, const OutValue* vals, uint32_t stride, uint32_t index, uint64_t n, bool desc,
                              uint64_t keep, uint64_t* images, void* state, uint32_t* cand, hipStream_t st, bool sampled = false, bool images_done = false);
bool topk_can_sample(uint64_t n, uint64_t keep);
uint64_t topk_cand_entries(uint64_t n);  // entries of the candidate buffer launch_topk_select needs for n groups  // enough groups for the sampled threshold (else the exact radix select)
hipError_t launch_topk_compact(const uint32_t* cand, uint64_t ncand, uint32_t nk, uint32_t na, const OutValue* keys,
                               const OutValue* aggs, const OutPartial* parts, const uint64_t* rep, OutValue* okeys, OutValue* oaggs,
                               OutPartial* oparts, uint64_t* orep, hipStream_t st);
hipError_t launch_distinct_layout(const Program& P, const GlobalTable& G, const DistinctArgs& D, hipStream_t st);
// COUNT(DISTINCT) over one-word members: one 256-bin partition pass (histogram, offsets, LDS-staged scatter) ...
hipError_t launch_radix_pass(const RadixArgs& A, uint32_t slices, hipStream_t st, bool have_hist = false);
// ... the per-bin LDS sets, the global-memory fallback, and the hand-over of the member counts to the set sizes
hipError_t launch_distinct_dedupe(const Program& P, const GlobalTable& G, const DedupeArgs& D, uint32_t grid, uint32_t block, hipStream_t st);
size_t distinct_dedupe_lds(const DedupeArgs& D);
// (`nreg` regions of fixed capacity: the 256 x kRecSubs sub-regions of a DISTINCT aggregate's member words)
hipError_t launch_compact_regions(const uint64_t* region, uint32_t nreg, uint64_t cap, const unsigned long long* count, const uint64_t* off,
                                  uint64_t* dst, hipStream_t st);
hipError_t launch_regrow_regions(const uint64_t* src, uint32_t nreg, uint64_t src_cap, uint64_t* dst, uint64_t dst_cap,
                                 unsigned long long* count, hipStream_t st);
hipError_t launch_distinct_words_global(const GlobalTable& G, const uint64_t* words, uint64_t n, uint64_t* table, uint64_t mask,
                                        uint32_t key_shift, unsigned long long* counts, uint32_t* err_flags, uint32_t grid,
                                        hipStream_t st);
hipError_t launch_distinct_add_counts(const Program& P, const GlobalTable& G, const unsigned long long* counts, uint32_t glob_off,
                                      hipStream_t st, const uint32_t* veto = nullptr);
hipError_t launch_distinct_insert(const Program& P, const GlobalTable& G, const DistinctArgs& D, uint32_t* err_flags,
                                  hipStream_t st);
hipError_t launch_export_partials(const Program& P, const GlobalTable& G, uint32_t nparts, uint64_t cap, uint64_t* out,
                                  uint64_t region_words, uint32_t* err_flags, hipStream_t st);
hipError_t launch_merge_partials(const Program& P, const GlobalTable& G, uint32_t nregions, uint64_t cap, const uint64_t* in,
                                 uint64_t region_words, uint32_t* err_flags, unsigned long long* ngroups, hipStream_t st,
                                 uint64_t limit = 0, bool unique_keys = false);
hipError_t launch_arith(const ArithArgs& A, hipStream_t st);
hipError_t launch_publish_counters(const unsigned long long* src, unsigned long long* dst, uint32_t n, hipStream_t st);
hipError_t launch_partition(const Program& P, const PartArgs& A, uint32_t grid, hipStream_t st);
struct HeaderList { unsigned long long* h[kMaxParts]; };  // the headers of the row regions a rank received (one per source)
hipError_t launch_exchange_verdict(const HeaderList& H, uint32_t nregions, uint32_t* err_flags, hipStream_t st);
hipError_t launch_stamp_verdict(unsigned long long* headers, uint32_t nregions, uint64_t stride_words, const uint32_t* err_flags,
                                uint32_t host_status, hipStream_t st);
hipError_t launch_dense_to_segments(unsigned long long* headers, uint32_t nregions, uint64_t stride_words, uint64_t sub_rows, hipStream_t st,
                                    const uint32_t* dest_cap = nullptr);  // dest_cap: regions of their own capacities (PartArgs::per_dest)
struct SpecEntry {
    const char* name;
    SpecSig sig;
    hipError_t (*launch)(const Program& P, const FastArgs& F, const GlobalTable& G, unsigned long long* ngroups,
                         uint32_t grid, uint32_t block, bool wide, const WordLogArgs& L, hipStream_t st);
    // records mode (partitioned GROUP BY): Filter + packed key -> 16-byte records in the hash regions
    hipError_t (*launch_records)(const Program& P, const FastArgs& F, uint32_t grid, bool wide, const WordLogArgs& L, hipStream_t st);
};
size_t spec_records_lds_bytes();
// n1k_bins.hip: the partitioned GROUP BY over 16-byte records — second partition pass (`bins_per_seg` bins of fixed capacity
// per hash region, a power of two <= 256) and the per-bin LDS tables (`block` threads per bin, `per_thread` = records a thread loads at once)
hipError_t launch_radix_scatter16(const RadixArgs& A, uint32_t wpr, uint32_t bins_per_region, hipStream_t st);
hipError_t launch_radix_scatter_words(const RadixArgs& A, uint32_t wpr, uint32_t bins_per_region, hipStream_t st);  // the same for 8-byte member words
size_t agg_bins16_lds_bytes(const Program& P, uint32_t slots);  // LDS of one agg_bins16 workgroup with `slots` table slots
hipError_t launch_agg_bins16(const Program& P, const BinAggArgs& A, uint32_t grid, uint32_t block, uint32_t per_thread, hipStream_t st,
                             bool specialise = true);  // (specialise: the plan's one aggregate as a compile-time constant)
const std::vector<SpecEntry>& spec_registry();
hipError_t launch_finalize(const Program& P, const GlobalTable& G, OutValue* out_keys, OutValue* out_aggs,
                           OutPartial* out_parts, uint64_t* out_rep, unsigned long long* out_count, uint64_t max_out,
                           uint32_t* err_flags, hipStream_t st);
// one workgroup, tables of at most 8192 slots: FinalGroup + the counters published to the host (+ table and counters left as
// n1k_reset leaves them): the one last kernel of a query
hipError_t launch_finalize_small(const Program& P, const GlobalTable& G, OutValue* out_keys, OutValue* out_aggs, OutPartial* out_parts,
                                 uint64_t* out_rep, unsigned long long* counters, unsigned long long* host_counters, uint64_t max_out,
                                 uint32_t* err_flags, bool clear, hipStream_t st);
hipError_t launch_filter_mask(const Program& P, uint64_t nrows, uint64_t* mask_words, uint32_t* tile_counts,
                              uint32_t* err_flags, uint32_t grid, hipStream_t st);
hipError_t launch_tile_scan(const uint32_t* counts, uint64_t* offsets, uint64_t ntiles, unsigned long long* total,
                            hipStream_t st);
hipError_t launch_filter_compact(const uint64_t* mask_words, const uint64_t* tile_offsets, uint64_t nrows,
                                 uint64_t row_base, uint64_t* out_rows, uint32_t grid, hipStream_t st);
// Filter alone in one pass: predicate + ordered compaction, tile offsets by a chained scan (tile_state: ntiles + 1 words, the
// last one the tile counter; zeroed here)
hipError_t launch_filter_stream(const Program& P, uint64_t nrows, uint64_t row_base, uint64_t* out_rows, unsigned long long* tile_state,
                                unsigned long long* tile_counter, unsigned long long* total, uint32_t* err_flags, uint32_t grid,
                                hipStream_t st, bool fast = false);  // fast: ONE TERM_NUM_* comparison of a TAGGED64 column, wide loads
// ---- raw JSON documents -> leaf columns on the device (n1k_jsondev.hip)
constexpr uint32_t kJsonMaxSteps = 4;  // field names of a leaf path below the keyspace alias
struct JsonDevPath {
    uint32_t nsteps;
    uint32_t name_off[kJsonMaxSteps], name_len[kJsonMaxSteps];  // the names' bytes in JsonDevArgs::names
};
struct JsonDevArgs {
    const uint8_t* bytes;     // the batch's documents, back to back (+ 32 spare bytes); document d = [offsets[d] - base, offsets[d + 1] - base)
    const uint64_t* offsets;  // ndocs + 1
    uint64_t base, ndocs;
    uint32_t npaths, tab_bits;
    JsonDevPath paths[kMaxCols];
    char names[1024];
    uint8_t* out_tags[kMaxCols];
    uint64_t* out_payload[kMaxCols];
    uint8_t* status;                    // per document: 0 extracted here, 1 left to the host's scalar extractor
    unsigned long long* tab_hash;       // the batch's string table: 1 << tab_bits slots, 0 = free
    unsigned long long* tab_first;      // [63] valid [offset of the first occurrence in `bytes` : 39][length : 24]
    uint32_t* new_list;                 // slots taken by this batch, in no particular order
    unsigned long long* new_count;
    uint32_t new_cap;
};
hipError_t launch_json_extract(const JsonDevArgs& A, uint32_t num_cus, hipStream_t st);
hipError_t launch_json_remap(const JsonDevArgs& A, const uint32_t* code_of, hipStream_t st);
hipError_t launch_json_gather_first(const JsonDevArgs& A, uint64_t n, unsigned long long* out, hipStream_t st);
hipError_t launch_json_scatter_codes(const uint32_t* new_list, const uint32_t* codes, uint64_t n, uint32_t* code_of, hipStream_t st);
hipError_t launch_json_patch(const JsonDevArgs& A, const uint64_t* docs, const uint8_t* tags, const uint64_t* payload, uint64_t n, hipStream_t st);
hipError_t launch_synth(const SynthArgs& a, hipStream_t st);

}  // namespace n1k
