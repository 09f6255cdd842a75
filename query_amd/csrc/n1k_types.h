// n1k_types.h — structures shared by the host engine and the HIP kernels.
//
// The host compiles the reference's plan JSON (plan/filter.go:46-53,
// plan/group.go:54-70) into a small, wave-uniform "program"; the kernels
// interpret it with scalar control flow (every lane runs the same op on its
// own rows), so the interpretive overhead stays on the scalar unit.
#pragma once
#ifdef __HIPCC_RTC__
// hiprtc has no <stdint.h>: same widths and underlying types as the LP64 host headers
typedef unsigned char uint8_t;
typedef unsigned short uint16_t;
typedef unsigned int uint32_t;
typedef unsigned long uint64_t;
typedef int int32_t;
typedef long int64_t;
typedef unsigned long uintptr_t;
typedef unsigned long size_t;
#define INT64_MAX 9223372036854775807L
#define INT64_MIN (-9223372036854775807L - 1)
#else
#include <stdint.h>
#endif

namespace n1k {

constexpr int kMaxCols = 16;
constexpr int kMaxTerms = 12;
constexpr int kMaxLogic = 32;
constexpr int kMaxKeys = 4;
constexpr int kMaxAggs = 8;
constexpr int kMaxLdsWords = 40;   // 1 key word + accumulators

constexpr uint64_t kEmptyKey = 0xFFFFFFFFFFFFFFFFull;
// device counters of a handle: [0] rows selected [1] groups [2] out count [3] filter total [4] rehash scratch
// [5] DISTINCT region words [8..11] pair-log cursors [12] error flags [13] wide key values [16..19] word-log cursors
// [25] survivor count saved by the optimistic partitioned path [26] its region's group count (copied for the host)
// [20] COUNT(DISTINCT) optimistic path: set / bin overflow flags [21] records of the partitioned path [22] its singleton
// partial groups [24] COUNT(DISTINCT) exact path: LDS-set overflow flag [27] the n1k_status a peer's verdict carried (ERR_PEER_FAILED)
constexpr uint32_t kCounters = 32;

// value tags == n1k_tag (include/n1k.h)
enum : uint32_t { T_MISSING = 0, T_NULL, T_FALSE, T_TRUE, T_INT, T_FLOAT, T_STRING, T_ARRAY, T_OBJECT };

// 4-valued logic of N1QL (expression/logic_and.go:64-89, logic_or.go:98-123)
enum : uint32_t { L_FALSE = 0, L_TRUE = 1, L_NULL = 2, L_MISSING = 3 };

enum : uint32_t { COLK_TAGGED64 = 0, COLK_DICT32 = 1 };

struct DevCol {
    const uint8_t* tags;
    const uint64_t* payload;
    const uint32_t* codes;
    uint32_t kind;
    uint32_t pad;
};

struct Operand {
    uint32_t is_const;  // 0: column, 1: constant
    uint32_t col;
    uint32_t ctag;
    uint32_t pad;
    uint64_t cpayload;
};

// predicate terms (leaves of the 4-valued logic tree)
enum : uint32_t {
    TERM_EQ = 0,       // expression/comp_eq.go:76-78
    TERM_LT,           // expression/comp_lt.go:57-65
    TERM_LE,           // expression/comp_le.go:57-65
    TERM_BETWEEN,      // expression/comp_between.go:58-78
    TERM_IS_NULL,      // expression/comp_null.go:58-67
    TERM_IS_NOT_NULL,  // comp_null.go:116-125
    TERM_IS_MISSING,   // comp_missing.go:62-69
    TERM_IS_NOT_MISSING,
    TERM_IS_VALUED,    // comp_valued.go:61-68
    TERM_IS_NOT_VALUED,
    TERM_TRUTH,        // a bare value used as a condition: type + Truth()
    // fast forms of LT/LE/EQ when operand b is a NUMBER constant (same semantics, fewer instructions):
    // a <op> const with op in {<, <=, >, >=, =}   ("(50 < x)" is stored as x GT 50)
    TERM_NUM_LT, TERM_NUM_LE, TERM_NUM_GT, TERM_NUM_GE, TERM_NUM_EQ,
    TERM_STR_EQ        // a = STRING constant (dictionary code compare)
};

struct Term {
    uint32_t op;
    uint32_t pad;
    Operand a, b, c;
};

enum : uint8_t { LOGIC_PUSH = 0, LOGIC_AND, LOGIC_OR, LOGIC_NOT };
struct LogicOp {
    uint8_t op;
    uint8_t arg;  // PUSH: term index; AND/OR: arity
};

// group key packing: every key becomes a bit field of one 63-bit word
enum : uint32_t { KEYM_DICT = 0, KEYM_TAGGED = 1 };
struct KeySpec {
    Operand src;
    uint32_t mode;
    uint32_t bits;
    uint32_t shift;
    uint32_t pad;
};

enum : uint32_t { AGG_SUM = 0, AGG_COUNT, AGG_COUNTN, AGG_AVG, AGG_MIN, AGG_MAX,
                  AGG_ARRAY };  // ARRAY_AGG: every operand but MISSING is logged per group, the arrays are built at finish
struct AggSpec {
    uint32_t kind;
    uint32_t distinct;
    uint32_t has_operand;
    uint32_t lds_off;   // first accumulator word inside an LDS slot
    uint32_t glob_off;  // first accumulator word inside a global-table row
    uint32_t log_index; // DISTINCT: which pair log this aggregate appends to
    uint32_t lds_n;     // accumulator words inside an LDS slot (0: a DISTINCT aggregate whose kernel counts nothing in LDS)
    uint32_t pad;
    Operand src;
};

// LDS accumulator words per aggregate (the LDS table is word-major: word w of slot s lives at lds[w * S + s])
//   COUNT/COUNTN : [cnt]
//   SUM          : [isum i64][fsum f64][flags]          one atomic per row; flags are read-mostly
//   AVG          : [isum i64][fsum f64][flags][n]
//   MIN/MAX      : [flags][ival i64][fval sortable u64][sval rank<<32|code]
//   * DISTINCT   : [n_int][n_float][n_other]   operands per value class (sizes the dedupe regions); the
//                  (group key, class, value) pairs go to the aggregate's pair log
// global accumulator words per aggregate
//   COUNT/COUNTN : [cnt]
//   SUM          : [isum_lo][isum_hi][fsum][flags]
//   AVG          : [isum_lo][isum_hi][fsum][flags][n]
//   MIN/MAX      : [flags][ival][fval][sval]
//   * DISTINCT   : [distinct count (filled at finish)][n_int][n_float][n_other][special]
//                  SUM/AVG(DISTINCT) append the SUM words [isum_lo][isum_hi][fsum][flags] of the distinct members
constexpr uint32_t kLdsWordsDistinct = 3, kGlobWordsDistinct = 5, kGlobWordsDistinctSum = 9, kMaxDistinct = 4;
constexpr uint32_t kLdsWordsSum = 3, kLdsWordsAvg = 4, kGlobWordsSum = 4, kGlobWordsAvg = 5, kWordsMinMax = 4;

// SUM/AVG flag bits: which kinds of NUMBER operands were met.  intValue.Add keeps an int64 only for same-sign
// operands (value/integer.go:266-277), so the sign mix decides the representation of the result.
enum : uint64_t { SF_NONNEG_INT = 1, SF_NEG_INT = 2, SF_FLOAT = 4 };

// MIN/MAX flag bits
enum : uint64_t { MM_FALSE = 1, MM_TRUE = 2, MM_INT = 4, MM_FLOAT = 8, MM_STRING = 16, MM_OTHER = 32 };

// run-time problem bits reported by the kernels
enum : uint32_t {
    ERR_UNPACKABLE_KEY = 1,   // a group key value does not fit its bit field (wide ints, non-integral floats)
    ERR_TABLE_FULL = 2,       // global group table capacity exceeded
    ERR_UNSUPPORTED_VALUE = 4, // e.g. ordering two arrays/objects
    ERR_EXCHANGE_OVERFLOW = 16, // a sender's partial-group region overflowed (seen by the merge on every rank)
    ERR_EXCHANGE_WIDE = 32,     // a sender's keys hold wide-value codes: partial groups cannot travel
    // learnt from the verdict word of a received region (multi-GPU exchange): some SENDER's step failed — seen by every
    // receiver alike, so every rank's n1k_finish fails in the same step
    ERR_PEER_UNPACKABLE = 64,   // a sender dropped rows whose group key does not pack (its ERR_UNPACKABLE_KEY)
    ERR_PEER_UNSUPPORTED = 128, // a sender met a value outside the device subset (its ERR_UNSUPPORTED_VALUE)
    ERR_PEER_FAILED = 256       // a sender failed on the host before the collective; its n1k_status is in counter [27]
};
constexpr uint32_t kErrFromVerdict = ERR_EXCHANGE_OVERFLOW | ERR_EXCHANGE_WIDE | ERR_PEER_UNPACKABLE | ERR_PEER_UNSUPPORTED | ERR_PEER_FAILED;

// Verdict word of a region that travels between GPUs (word 1 of a packed row region's or a partial-group region's header).
// A sender writes the same verdict into EVERY region it ships, so all receivers read the same set of verdicts.
enum : uint64_t {
    VD_OVERFLOW = 1,     // some region of this sender overflowed
    VD_WIDE = 2,         // partial groups: the keys hold device-local wide-value codes
    VD_UNPACKABLE = 4,   // rows: the sender's partition dropped rows whose key does not pack
    VD_UNSUPPORTED = 8,  // rows: the sender's Filter met a value outside the device subset
    VD_STATUS_SHIFT = 8  // bits 8..15: the n1k_status of a sender that failed on the host before the collective
};
// the error flags live in counter [12]; the largest status a peer's verdict carried goes to counter [27]
constexpr uint32_t kPeerStatusFromErr = 15;
#ifndef __HIPCC_RTC__
static_assert(12 + kPeerStatusFromErr == 27, "peer status counter");
#endif

struct Program {
    uint32_t ncols, nterms, nlogic, nkeys, naggs;
    uint32_t emit_packed_key;  // FinalGroup writes the packed group key where the representative row would go (ARRAY_AGG: the host
                               // matches the logged operands to the groups by it)
    uint32_t lds_words;     // words per LDS slot (key + accumulators)
    uint32_t glob_words;    // words per global row (accumulators only)
    uint32_t nan_code, pinf_code, ninf_code;  // dictionary codes of "NaN" / "+Infinity" / "-Infinity": what such FLOAT group keys marshal to
    uint32_t want_rep_row;  // keep min row ordinal per group
    uint32_t rep_lds_word;  // LDS word of the rep row (when wanted)
    uint32_t dict_size;
    uint32_t empty_str_code, empty_arr_code, empty_obj_code;  // codes of "", "[]", "{}" or 0xFFFFFFFF
    const uint32_t* str_rank;  // rank[code]: bytewise order of the dictionary strings (value/string.go:116-130)
    // wide key values: numbers that do not fit a key bit field (non-integral floats, big ints) are replaced by
    // their slot in one of these per-handle open-addressed value tables (2^wide_bits entries, kEmptyKey = free)
    uint64_t* wide_int;
    uint64_t* wide_flt;
    unsigned long long* wide_count;  // distinct wide values met so far
    uint32_t wide_bits, pad1;
    DevCol cols[kMaxCols];
    Term terms[kMaxTerms];
    LogicOp logic[kMaxLogic];
    KeySpec keys[kMaxKeys];
    AggSpec aggs[kMaxAggs];
};

struct GlobalTable {
    uint64_t* keys;     // capacity entries, kEmptyKey when free
    uint64_t* acc;      // capacity * glob_words
    uint64_t* rep_row;  // capacity entries (or null)
    uint64_t capacity;  // power of two
};

struct ScanArgs {
    uint64_t nrows;
    uint64_t row_base;        // ordinal of row 0 of this batch
    uint32_t lds_slots;       // S: slots of the LDS table (any value >= 2)
    uint32_t lds_max_fill;    // HASH mode: stop inserting new keys beyond this many occupied slots
    uint32_t direct_stride[kMaxKeys];  // DIRECT mode: slot = sum(field_k * stride_k), field_k < radix_k
    uint32_t direct_radix[kMaxKeys];
    uint32_t* err_flags;      // device word, OR of ERR_*
    unsigned long long* rows_selected;  // device counter
    // DISTINCT pair logs (one per DISTINCT aggregate): (group key, value, class) of every qualifying operand
    uint64_t* log_key[kMaxDistinct];
    uint64_t* log_val[kMaxDistinct];
    uint8_t* log_cls[kMaxDistinct];
    unsigned long long* log_cursor;  // kMaxDistinct counters
    uint64_t log_capacity;
    // COUNT(DISTINCT): pairs whose group key and value fit one 64-bit MEMBER WORD
    //   [key : nw_key_bits][class : 3][value : nw_val_bits]      (nw_key_bits + 3 + nw_val_bits == 64)
    // go to a second log of single words (8 B/pair) that n1k_finish radix-partitions and de-duplicates in LDS.
    uint64_t* log_word[kMaxDistinct];
    unsigned long long* word_cursor;  // kMaxDistinct counters
    uint32_t nw_key_bits, nw_val_bits;
    uint32_t dcache_slots;  // per DISTINCT aggregate: slots (power of two) of the workgroup's "already logged" cache, 0 = none
    uint32_t dcache_aggs;   // number of such caches (== DISTINCT aggregates of the plan)
    unsigned long long* word_hist;  // kMaxDistinct x 256 counters: first radix digit of the logged words (n1k_finish's first pass)
    const unsigned long long* nrows_dev;  // see FastArgs::nrows_dev
};

// COUNT(DISTINCT) inside the plan-specialised scan (n1k_spec.h): the member words of aggregate d are scattered by the
// first radix digit of mix64(word) straight into 256 HASH REGIONS of fixed capacity (the first partition pass of
// n1k_finish, fused into the scan: no log round trip, no histogram pass).  mix64 spreads distinct words evenly, so a
// region holds words / 256 plus slack; a word that finds its region full goes to the plain word log instead (and
// n1k_finish then takes the exact path over everything).
constexpr uint32_t kSpecDistinct = 2;  // DISTINCT aggregates a specialised kernel handles
// Write cursors that every workgroup bumps once per tile sit one per 128-byte line: 256 neighbouring counters share 16
// lines (a handful of L2 channels), and the atomics on them then bound the whole scatter.
constexpr uint32_t kCursorStride = 16;
struct WordLogArgs {
    uint64_t* region[kSpecDistinct];              // 256 regions x region_cap words
    unsigned long long* region_cursor[kSpecDistinct];  // 256 counters each, kCursorStride apart (words offered to a region, may exceed the capacity)
    uint64_t region_cap;
    uint64_t* over_word[kSpecDistinct];           // the plain word log (ScanArgs::log_word) and its cursors / histogram
    unsigned long long* over_cursor;              // kMaxDistinct counters, indexed by log_index
    unsigned long long* over_hist;                // kMaxDistinct x 256
    uint64_t over_capacity;
    // members that do not fit one word (non-integral floats, wide values): the pair log of ScanArgs
    uint64_t* log_key[kSpecDistinct];
    uint64_t* log_val[kSpecDistinct];
    uint8_t* log_cls[kSpecDistinct];
    unsigned long long* log_cursor;               // kMaxDistinct counters, indexed by log_index
    uint64_t log_capacity;
    uint32_t log_index[kSpecDistinct];
    uint32_t nw_key_bits, nw_val_bits;
    uint32_t dcache_slots, pad;                   // per aggregate: "already logged" cache in LDS (power of two), 0 = none
    // records mode (the partitioned GROUP BY's 16-byte records, region[0] / region_cursor[0]): set when a region is full —
    // the engine then redoes the batch on the exact path
    uint32_t* rec_overflow;
};

// radix partition of a word log by bits of mix64(word) (finish step of COUNT(DISTINCT), see n1k_kernels.hip)
struct RadixArgs {
    const uint64_t* src;
    uint64_t* dst;
    const uint64_t* seg_start;  // nseg + 1 entries: segment s is src[seg_start[s] .. seg_start[s+1])
    uint32_t nseg;
    uint32_t shift;             // bin = (mix64(word) >> shift) & 255
    unsigned long long* hist;   // nseg * 256 counters
    unsigned long long* cursor; // nseg * 256 write cursors (absolute positions in dst)
    uint64_t* out_start;        // nseg * 256 + 1 entries: starts of the finer segments
    // Segments of fixed capacity instead (hash regions): segment s is src[s * seg_stride .. + min(seg_count[s], seg_stride))
    const unsigned long long* seg_count;  // (counter of segment s at seg_count[s * kCursorStride])
    uint64_t seg_stride;
    uint32_t cursor_stride, pad1;         // `cursor` entry i lives at cursor[i * cursor_stride] (0 = 1)
    // Output bins of fixed capacity (no histogram pass): bin (s, b) is dst[(s * 256 + b) * bin_cap ..], `cursor` counts
    // from zero, words beyond the capacity are dropped and *overflow is set (the caller then takes the exact path)
    uint64_t bin_cap;
    uint32_t* overflow;
    // records: arrays that travel with the word (same permutation), see RecArrays
    uint32_t nextra, pad;
    const uint64_t* src_pay[2];
    uint64_t* dst_pay[2];
    const uint8_t* src_tag[2];
    uint8_t* dst_tag[2];
};

// High-cardinality GROUP BY: rows are projected to records (packed key + the aggregates' operands), radix
// partitioned by mix64(key) with the same passes as the COUNT(DISTINCT) words, and aggregated bin by bin in LDS.
constexpr uint32_t kRecOperands = 2;  // distinct operand sources a record carries
constexpr uint32_t kRecSubs = 8;      // sub-regions per hash region of the 16-byte records: one per workgroup label (n1k_spec.h)
struct RecArrays {
    uint64_t* key;
    uint64_t* pay[kRecOperands];
    uint8_t* tag[kRecOperands];
};
struct ProjectArgs {
    uint64_t nrows;
    uint64_t capacity;             // records the output arrays take
    RecArrays out;
    unsigned long long* cursor;    // records written
    Operand src[kRecOperands];
    uint32_t nsrc, pad;
    uint32_t* err_flags;
    unsigned long long* hist;  // 256 counters of the records' first radix digit (or null)
};
struct BinAggArgs {
    RecArrays in;
    // ... or 16-byte records (Rec16, n1k_tables.h) in bins of fixed capacity: bin i is rec[i * bin_stride .. + min(
    // bin_count[i], bin_stride)); entries whose key is kEmptyKey are padding
    const void* rec;
    const unsigned long long* bin_count;  // (entry i at bin_count[i * bin_count_stride])
    uint64_t bin_stride;
    uint32_t bin_count_stride, pad1;
    const uint64_t* bin_start;  // nbins + 1
    uint32_t nbins;
    uint32_t nsrc;
    uint32_t lds_slots, lds_max_fill;
    uint32_t agg_src[kMaxAggs];  // operand slot of every aggregate, 0xFFFFFFFF = none (count(*))
    uint32_t* err_flags;
    // when set: the bins' groups are appended to this region ([count][0][keys: emit_cap][accumulators]) instead of
    // being merged into the global table
    uint64_t* emit;
    uint64_t emit_cap;
    unsigned long long* emit_singletons;  // records emitted outside the bins' tables: their keys may repeat
};

struct DedupeArgs {
    const uint64_t* words;
    const uint64_t* bin_start;  // nbins + 1 (null with bin_count)
    // bins of fixed capacity: bin i is words[i * bin_stride .. + min(bin_count[i], bin_stride))
    const unsigned long long* bin_count;
    uint64_t bin_stride;
    uint32_t count_stride, pad1;  // bin_count entry i lives at bin_count[i * count_stride] (0 = 1)
    uint32_t direct_keys;       // > 0: packed keys are < direct_keys, the LDS member counters are indexed by the key itself
    uint32_t pad0;
    uint32_t nbins;
    uint32_t set_slots;         // LDS set size (power of two)
    uint32_t key_shift;         // word >> key_shift = packed group key
    uint32_t glob_off;
    unsigned long long* counts; // per global-table slot: distinct members found (added to the set size afterwards)
    uint32_t* overflow;         // set when a bin holds more distinct words than the LDS set takes
    uint32_t lds_counters;      // == table capacity when per-group LDS counters are used, else 0
    uint32_t pad;
};

// value classes of the DISTINCT sets (value/set.go:22-35 keeps one map per type; integral floats join the ints)
enum : uint32_t { DC_INT = 0, DC_FLOAT = 1, DC_OTHER = 2 };

struct DistinctArgs {
    const uint64_t* log_key;
    const uint64_t* log_val;
    const uint8_t* log_cls;
    uint64_t npairs;
    uint64_t* regions;     // capacity * 3 * 2 words: (offset, mask) per (slot, class)
    uint64_t* set_table;   // concatenated per-(group, class) open-addressed value sets, kEmptyKey when free
    unsigned long long* total_words;  // out: words needed by the regions
    uint32_t glob_off;     // first global word of the aggregate
    uint32_t kind;         // AGG_* of the DISTINCT aggregate
};

// ---- "fast" scan kernel: bounded plan shapes with every descriptor at a static index ------------------------
// <= 4 columns (each loaded once per row into registers), <= 2 cheap terms combined by AND, <= 2 dictionary
// keys addressed by perfect hash (DIRECT), <= 5 non-DISTINCT aggregates over columns.
constexpr int kFastCols = 3, kFastTerms = 2, kFastKeys = 2, kFastAggs = 5;
// Arithmetic nodes a plan-specialised kernel evaluates in registers (fused derived columns): column slots
// [ncols, ncols + nderived) of the shape; kSpecCols bounds the per-row register arrays of those kernels.
constexpr int kFastDerived = 3, kSpecCols = kFastCols + kFastDerived;

struct FastTerm {
    uint32_t op;    // TERM_NUM_* / TERM_IS_* / TERM_STR_EQ
    uint32_t col;   // column slot of operand a
    uint32_t ctag;
    uint32_t pad;
    uint64_t cpayload;
};
struct FastKey {
    uint32_t col, stride, radix, shift;
};
struct FastArgs {
    uint32_t ncols, nterms, nkeys, naggs;
    uint32_t nrows;      // rows of this launch (< 2^32)
    uint32_t lds_slots;  // S
    uint32_t hashed;     // 0: DIRECT perfect hash (dictionary keys, small domain); 1: open-addressed LDS table on the packed key
    uint32_t lds_max_fill;
    uint64_t row_base;
    DevCol cols[kFastCols];
    FastTerm terms[kFastTerms];
    FastKey keys[kFastKeys];
    uint32_t agg_col[kFastAggs];  // column slot of each aggregate's operand (unused when has_operand == 0)
    uint32_t pad_f;
    uint32_t nderived;            // fused arithmetic nodes (plan-specialised kernels only; their shape is in the SpecSig)
    uint64_t dconst[kFastDerived][4];  // payloads of the nodes' constant operands (the tags are part of the shape)
    uint32_t* err_flags;
    unsigned long long* rows_selected;
    uint64_t* slabs;  // when non-null: workgroup b stores its LDS table at slabs[b * lds_words * S ..] instead of merging
    unsigned long long* block_selected;  // with slabs: per-workgroup survivor counts (summed by the merge kernel)
    // when non-null: the batch really holds min(nrows, *nrows_dev) rows (a region received from another GPU: its row count
    // arrived with it and never visits the host)
    const unsigned long long* nrows_dev;
    // Segmented batch (plan-specialised scan only; nseg > 1): the columns hold nseg segments of seg_rows rows capacity each
    // (a multiple of 16), segment s holding min(seg_rows, seg_counts[s * seg_count_stride]) rows — the sub-regions of a
    // row region received from another GPU (n1k_exchange_rows), aggregated by ONE launch.
    uint32_t nseg, seg_rows, seg_count_stride, pad_seg;
    const unsigned long long* seg_counts;
};
constexpr uint32_t kMaxSegments = 8;  // (= kRowSubs: the counts live in scalar registers)
constexpr uint32_t kRowSubs = 8;  // sub-regions (segments) of a packed row region of the row exchange; their counts kCursorStride words apart

// derived columns: arithmetic nodes of the plan are evaluated once per batch by an element-wise kernel into a
// temporary TAGGED64 column; the scan kernels then see them as ordinary columns
enum : uint32_t { AR_ADD = 0, AR_MULT, AR_SUB, AR_DIV, AR_MOD, AR_NEG, AR_IDIV, AR_IMOD,
                  // expression/func_num.go: ROUND / TRUNC (value [, digits]), ABS, CEIL, FLOOR, SIGN, SQRT
                  AR_ROUND, AR_TRUNC, AR_ABS, AR_CEIL, AR_FLOOR, AR_SIGN, AR_SQRT,
                  AR_GREATEST, AR_LEAST };  // expression/func_comp.go: by value.Collate (derived columns only: they need the string ranks)
struct ArithArgs {
    uint32_t op, nops;
    Operand ops[4];
    DevCol cols[kMaxCols];
    uint64_t nrows;
    uint8_t* out_tags;
    uint64_t* out_payload;
    const uint32_t* str_rank;  // AR_GREATEST / AR_LEAST: bytewise rank of every dictionary string
    uint32_t* err_flags;       // ... and where ordering two arrays / objects is reported
};

constexpr uint32_t kMaxParts = 64;  // destinations of one partition launch (ranks of a node, with room)
struct PartArgs {
    uint64_t nrows;
    uint64_t capacity;  // rows per destination region
    uint32_t nparts, ncopy;      // ncopy: number of (input) columns shipped
    unsigned long long* counts;  // nparts counters (rows written per destination), count_stride words apart
    // Packed regions (n1k_exchange_rows): destination d's rows of every column live inside ONE region of region_bytes
    // bytes (so that one send per peer moves them): out_* then point at the column's place inside region 0 and row r of
    // destination d is element r of the array that starts d * region_bytes further on.  0 = one array per column with
    // `capacity` rows per destination.  With regions the count sits in the region header ([count][verdict]), and an
    // overflow raises verdict bit 0 in EVERY region (each receiver learns it from the exchange itself).
    uint64_t region_bytes;
    uint32_t count_stride;
    // Packed regions written by the run-time-built partition kernel: every destination's rows in `nsub` sub-regions of
    // `sub_rows` rows capacity (a multiple of 16), one per workgroup label blockIdx.x % nsub, each with its own count
    // kCursorStride words further on in the region header — tens of thousands of tiles reserving their runs through ONE
    // counter per destination serialise on it (~12 ns per same-address atomic: 0.6 ms per 100 M rows).  nsub <= 1: one
    // dense run per destination.
    uint32_t nsub;
    uint64_t sub_rows;
    // Regions of DIFFERENT capacities (per_dest: the row exchange under skew — every destination's regions are sized for what
    // THAT owner receives, so that a hot owner does not inflate the other P - 1 regions of every sender): region d still starts
    // d * region_bytes into the send buffer, but is laid out for dest_cap[d] rows (a multiple of 16 * nsub): a header of
    // hdr_bytes, then per input column its arrays, each padded to 16 bytes (part_region_next below; the host's
    // row_region_layout).  capacity / sub_rows / out_* are not used then.
    uint32_t per_dest, hdr_bytes;
    uint32_t dest_cap[kMaxParts];
    uint32_t* err_flags;
    uint8_t* out_tags[kMaxCols];
    uint64_t* out_payload[kMaxCols];
    uint32_t* out_codes[kMaxCols];
};

// a region's arrays, one after the other: the array of `width`-byte elements that starts at `off` ends at the next 16-byte boundary
#if defined(__HIPCC__) || defined(__HIPCC_RTC__)
__host__ __device__
#endif
inline uint64_t part_region_next(uint64_t off, uint64_t cap, uint32_t width) { return (off + cap * width + 15ull) / 16ull * 16ull; }

// compile-time shape of a plan handled by scan_spec_kernel (see n1k_spec.h)
struct SpecTerm {
    uint32_t op;         // TERM_NUM_* / TERM_IS_* / TERM_STR_EQ
    uint32_t col;        // column slot
    uint32_t const_int;  // TERM_NUM_*: 1 = the constant is an INT, 0 = FLOAT
};
struct SpecAgg {
    uint32_t kind;  // AGG_*
    uint32_t has_operand;
    uint32_t col;
    uint32_t distinct;  // COUNT(DISTINCT col): one-word members scattered into the hash regions (WordLogArgs)
};
struct SpecOperand {
    uint32_t is_const;  // 1: constant (tag below, payload in FastArgs::dconst); 0: column slot (input or earlier derived)
    uint32_t v;         // column slot, or the constant's tag
};
struct SpecDerived {
    uint32_t op, nops;  // AR_*
    SpecOperand ops[4];
};
struct SpecSig {
    int ncols, nterms, nkeys, naggs;
    int hashed, nderived;  // hashed 1: keys go through the open-addressed LDS table; nderived: fused arithmetic nodes
    int mode, seg;         // mode 0: scan kernels (+ records front end); 1: partition kernels of the row exchange (aggregates left out); seg 1: the scan kernels take segmented batches (scan_spec_body's SEG)
    uint32_t col_kind[kFastCols];
    SpecTerm terms[kFastTerms];
    uint32_t key_col[kFastKeys];
    SpecAgg aggs[kFastAggs];
    SpecDerived derived[kFastDerived];
};

}  // namespace n1k
