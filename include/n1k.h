/*
 * n1k.h — C ABI of the MI355X-native Filter -> Group -> Aggregate pipeline.
 *
 * This is the drop-in boundary for ONE hot path of the N1QL engine
 * (pavel-paulau/query): the chain
 *
 *     Parallel{ Sequence[ Filter, InitialGroup ] } -> IntermediateGroup -> FinalGroup
 *
 * Every entry point names the reference interface it replaces (paths are
 * relative to the reference tree).  The reference has no FFI of its own (it is
 * 100 % Go); the cgo stub a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C, no C++/torch types; all pointers are caller-owned unless stated;
 *   - every function returns an n1k_status; the message for the last failure
 *     of a handle is n1k_last_error(h);
 *   - one handle == one operator copy (reference: Operator.Copy(),
 *     execution/parallel.go:67-73).  A handle is NOT re-entrant, different
 *     handles may be driven from different OS threads concurrently;
 *     n1k_stop() alone may be called concurrently with a push on the same
 *     handle (reference: SendStop, execution/base.go:313-338).
 *   - the library never falls back to a CPU implementation: if no HIP device
 *     is usable every compute call fails with N1K_DEVICE_ERROR.
 */
#ifndef N1K_H
#define N1K_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define N1K_ABI_VERSION 3 /* 2: n1k_result.nproj / proj, projection and communicator entry points; 3: n1k_stats.query_ms,
                             N1K_REGION_FULL, n1k_partials_step, n1k_failure_is_global */

/* ---------------------------------------------------------------- status -- */

typedef enum n1k_status {
    N1K_OK = 0,
    N1K_UNSUPPORTED = 1,      /* plan/expression outside the device subset: caller keeps the reference operators */
    N1K_EVAL_ERROR = 2,       /* ≙ errors.NewEvaluationError (execution/filter.go:51-54, group_initial.go:62-65) */
    N1K_DEVICE_ERROR = 3,     /* HIP failure or no device; ≙ context.Fatal */
    N1K_OOM = 4,              /* host/device allocation failed or group-table capacity exceeded */
    N1K_STOPPED = 5,          /* n1k_stop() was observed (≙ processItem returning false after SendStop) */
    N1K_INVALID = 6,          /* bad argument / batch shape */
    N1K_UNSUPPORTED_DATA = 7, /* a value met at run time is outside the device subset (see n1k_tag) */
    N1K_REGION_FULL = 8       /* multi-GPU: a fixed-capacity region of an exchange overflowed; the step is void on EVERY rank alike —
                                 repeat it with a larger capacity on every rank (never an allocation failure: that is N1K_OOM) */
} n1k_status;

/* ------------------------------------------------------------ value tags -- */

/*
 * Value tags.  Order == N1QL type order (value/value.go:69-79:
 * MISSING < NULL < BOOLEAN < NUMBER < STRING < ARRAY < OBJECT < BINARY), with
 * BOOLEAN and NUMBER split into their two representations:
 * FALSE<TRUE (value/boolean.go:99-113), INT/FLOAT = intValue/floatValue
 * (value/integer.go, value/float.go).
 */
typedef enum n1k_tag {
    N1K_T_MISSING = 0,
    N1K_T_NULL = 1,
    N1K_T_FALSE = 2,
    N1K_T_TRUE = 3,
    N1K_T_INT = 4,    /* payload = int64 */
    N1K_T_FLOAT = 5,  /* payload = IEEE-754 binary64 bits; never integral-in-int64-range for parsed input
                         (NewValue folds those to INT, value/value.go:377-382) */
    N1K_T_STRING = 6, /* payload = code in the handle's string dictionary */
    N1K_T_ARRAY = 7,  /* payload = dictionary code of the canonical JSON text (group keys / COUNT only) */
    N1K_T_OBJECT = 8  /* idem */
} n1k_tag;

/* DICT32 columns carry no tag byte: these two codes stand for MISSING / NULL */
#define N1K_CODE_MISSING 0xFFFFFFFFu
#define N1K_CODE_NULL 0xFFFFFFFEu

/* A tagged scalar as returned in results (16 bytes). */
typedef struct n1k_value {
    uint8_t tag; /* n1k_tag */
    uint8_t pad[7];
    union {
        int64_t i;
        double f;
        uint64_t code; /* STRING/ARRAY/OBJECT: dictionary code */
    } v;
} n1k_value;

/* --------------------------------------------------------- column batches -- */

typedef enum n1k_col_kind {
    N1K_COL_TAGGED64 = 0, /* tags[nrows] (1 B) + payload[nrows] (8 B) */
    N1K_COL_DICT32 = 1    /* codes[nrows] (4 B): string code, N1K_CODE_MISSING or N1K_CODE_NULL */
} n1k_col_kind;

typedef struct n1k_col {
    uint32_t kind;           /* n1k_col_kind */
    uint32_t reserved;
    const uint8_t *tags;     /* TAGGED64 */
    const uint64_t *payload; /* TAGGED64 */
    const uint32_t *codes;   /* DICT32 */
} n1k_col;

/*
 * One batch of rows, column-wise.  Column i holds, for every row, the value of
 * leaf path i of the plan (n1k_num_columns / n1k_column_path): what
 * Field.Apply / Identifier.Evaluate would return for that row
 * (expression/nav_field.go:134-160, identifier.go:48-51).
 */
typedef struct n1k_batch {
    uint64_t nrows;
    uint32_t ncols;
    uint32_t reserved;
    const n1k_col *cols;
} n1k_batch;

/* ----------------------------------------------------------------- results -- */

/* Raw, mergeable accumulators of one aggregate of one group (the reference's
 * "partial" values: algebra/aggregate.go:25-40). */
typedef struct n1k_partial {
    int64_t count;      /* COUNT/COUNTN: the count.  SUM/AVG: number of NUMBER operands */
    int64_t isum;       /* SUM/AVG: exact integer part of the sum when int_exact != 0 */
    double fsum;        /* SUM/AVG: sum of the FLOAT operands (plus the integer part when int_exact == 0) */
    uint8_t int_exact;  /* 1: no FLOAT operand seen and the integer sum fits int64 */
    uint8_t has_float;  /* 1: at least one FLOAT operand */
    uint8_t pad[6];
    n1k_value extreme;  /* MIN/MAX: current winner (NULL when none) */
    int64_t distinct;   /* COUNT(DISTINCT)/COUNTN(DISTINCT): set size; SUM/AVG(DISTINCT): set size */
} n1k_partial;

typedef struct n1k_result {
    uint64_t ngroups;
    uint32_t nkeys;
    uint32_t naggs;
    const n1k_value *keys;       /* [ngroups][nkeys]  group key values (MISSING keys stay MISSING) */
    const n1k_value *aggs;       /* [ngroups][naggs]  ComputeFinal() values, plan order */
    const n1k_partial *partials; /* [ngroups][naggs]  */
    const uint64_t *rep_row;     /* [ngroups] smallest input row ordinal of the group (the reference keeps the
                                    first row it met as carrier, execution/group_initial.go:69-72) */
    uint64_t nselected;          /* Filter-only plans: number of rows that passed */
    const uint64_t *selected;    /* Filter-only plans: their row ordinals, ascending */
    uint32_t nproj;              /* plans with an InitialProject: number of result terms, else 0 */
    uint32_t reserved1;
    const n1k_value *proj;       /* [ngroups][nproj]  value of every result term (≙ the "projection" attachment,
                                    execution/project_initial.go:100-144; a MISSING term is left out of the row) */
} n1k_result;

typedef struct n1k_stats {
    uint64_t rows_in;       /* ≙ #itemsIn  (execution/base.go:32-46) */
    uint64_t rows_selected; /* rows that passed the Filter (≙ Filter #itemsOut) */
    uint64_t groups_out;    /* ≙ FinalGroup #itemsOut */
    uint64_t batches;
    double device_ms;       /* hipEvent time of all device work (≙ execTime) */
    uint64_t bytes_scanned; /* algorithmic column bytes read by the scan kernel */
    uint32_t agg_mode;      /* n1k_agg_mode actually used by the last batch */
    uint32_t spec_kernel;   /* plan-specialised kernel that ran the last batch: 0 none, 1 prebuilt, 2 built at run time, 3 built at run time with the plan's arithmetic nodes evaluated in registers */
    uint64_t wide_key_values; /* distinct key numbers held in the wide-value tables (floats, |int| beyond the field); set by finish/export */
    uint32_t distinct_path;  /* how the last finish built the DISTINCT sets: bit 0 per-group sets in global memory (two-word
                                pairs), bit 1 radix partition + LDS sets (one-word members), bit 2 global one-word set (fallback) */
    uint32_t reserved0;
    uint64_t topk_candidates; /* ORDER BY ... LIMIT: groups that left the device after the top-k filter (0 = filter not used) */
    uint64_t json_device_docs; /* documents of n1k_push_json whose leaf values the DEVICE extractor produced (the rest: the host's) */
    double query_ms;        /* hipEvent time of the last WHOLE query on the handle's stream: from n1k_reset (reopen) to the last
                               kernel / copy of n1k_finish — every kernel of the query and the gaps between them; 0 until a
                               finish has completed.  device_ms above covers the batches' kernels only. */
} n1k_stats;

typedef enum n1k_agg_mode {
    N1K_MODE_AUTO = 0,
    N1K_MODE_LDS_HASH = 1,  /* open-addressed LDS table per workgroup, global atomic merge */
    N1K_MODE_LDS_DIRECT = 2,/* perfect-hash LDS table (small dictionary-coded key domain), slab merge */
    N1K_MODE_GLOBAL = 3,    /* no LDS stage: global open-addressed table only */
    N1K_MODE_PARTITIONED = 4 /* high cardinality: rows -> records, radix partition on the key hash, per-bin LDS tables */
} n1k_agg_mode;

typedef struct n1k_handle n1k_handle;

/* ------------------------------------------------------------ life cycle -- */

/*
 * Build an operator from the reference's own plan JSON.
 * Replaces: execution.builder.VisitParallel / VisitFilter / VisitInitialGroup
 *           (execution/build.go:281-296, 457-491) + NewFilter / NewInitialGroup /
 *           NewIntermediateGroup / NewFinalGroup.
 * plan_json is ONE of
 *   {"#operator":"Sequence","~children":[{"#operator":"Filter","condition":"…"},
 *                                        {"#operator":"InitialGroup","group_keys":[…],"aggregates":[…]}]}
 *   {"#operator":"InitialGroup", …}            (no Filter)
 *   {"#operator":"Filter","condition":"…"}     (Filter only: result = selected row ordinals)
 *   {"#operator":"Parallel","~child":<one of the above>, "maxParallelism":n}
 *   {"#operator":"Sequence","~children":[<one of the above>, IntermediateGroup, FinalGroup,      (the grouped tail:
 *        {"#operator":"Filter","condition":"…"},                                       HAVING, plan/filter.go
 *        {"#operator":"Order","sort_terms":[{"expr":"…","desc":true}],"offset":"…","limit":"…"},  plan/order.go:51-79
 *        {"#operator":"Offset","expr":"…"}, {"#operator":"Limit","expr":"…"}]}               plan/limit.go:46-53)
 *        {"#operator":"Parallel","~child":{"#operator":"Sequence","~children":[                   (HAVING and projection
 *             {"#operator":"Filter",…}, {"#operator":"InitialProject","result_terms":[{"expr":"…","as":"…"}]}]}},    as the planner
 *        {"#operator":"FinalProject"}]}                                                            nests them, plan/project.go:73-110)
 *     IntermediateGroup / FinalGroup must repeat the InitialGroup's lists (plan/group.go:106-273) and are subsumed;
 *     the result terms of an InitialProject are expressions over group keys and aggregates (constants, arithmetic,
 *     round / trunc / abs / ceil / floor / sign / sqrt): n1k_result.proj then holds their values per group, and a sort
 *     term may name a term's alias (`alias`) or repeat its expression; star / raw / distinct projections are N1K_UNSUPPORTED;
 *     the HAVING condition and the sort terms may name only group keys and aggregates of the plan (by their text);
 *     offset / limit are integer constants.  n1k_finish then returns the groups filtered, ordered and cut.
 * exactly as plan.(*Filter).MarshalJSON (plan/filter.go:46-53),
 * plan.(*InitialGroup).MarshalJSON (plan/group.go:54-70), plan/sequence.go:48-57 and
 * plan/parallel.go:54-67 emit them; expressions are expression.Stringer text.
 * Returns N1K_UNSUPPORTED (and *out == NULL) for anything outside the device subset.
 */
n1k_status n1k_create(const char *plan_json, size_t len, n1k_handle **out);

/* ≙ Operator.Done() (execution/execution.go:26-64): frees host + device state. */
void n1k_destroy(n1k_handle *h);

/* ≙ reopen() (execution/group_initial.go:117-120): drop all groups, keep plan + dictionary. */
n1k_status n1k_reset(n1k_handle *h);

/* ≙ SendStop() (execution/base.go:313-338): safe from any thread. */
void n1k_stop(n1k_handle *h);

const char *n1k_last_error(const n1k_handle *h);
/* message of a failed n1k_create (thread-local) */
const char *n1k_create_error(void);

/* --------------------------------------------------------------- binding -- */

/* Leaf paths of the plan, in column order: stringer text such as
 * "(`default`.`price`)".  The caller evaluates exactly these per row. */
uint32_t n1k_num_columns(const n1k_handle *h);
const char *n1k_column_path(const n1k_handle *h, uint32_t i);
uint32_t n1k_num_keys(const n1k_handle *h);
uint32_t n1k_num_aggregates(const n1k_handle *h);
/* agg.String() of aggregate i — the key of the reference's "aggregates"
 * attachment map (execution/group_initial.go:74-79). */
const char *n1k_aggregate_name(const n1k_handle *h, uint32_t i);

/* Result terms of the plan's InitialProject (0 when it has none): expression text and explicit alias ("" = none; the
 * caller derives one as algebra.ResultTerm does: the last name of a path, else $1, $2, ...). */
uint32_t n1k_num_projection_terms(const n1k_handle *h);
const char *n1k_projection_expr(const n1k_handle *h, uint32_t i);
const char *n1k_projection_alias(const n1k_handle *h, uint32_t i);

/* ------------------------------------------------------------ dictionary -- */

/* Intern n strings (bytes[offsets[i]..offsets[i+1])) into the handle's
 * dictionary; out_codes[i] receives the code.  Equal byte strings always get
 * equal codes (string equality ≙ value/string.go:82-96).  Codes are handed out in
 * interning order starting at 0; the plan's own string constants are interned at the
 * first push, so a dictionary interned right after n1k_create keeps code == index. */
n1k_status n1k_dict_intern(n1k_handle *h, uint32_t n, const uint64_t *offsets, const char *bytes,
                           uint32_t *out_codes);
uint32_t n1k_dict_size(const n1k_handle *h);
n1k_status n1k_dict_get(const n1k_handle *h, uint32_t code, const char **ptr, size_t *len);

/* ---------------------------------------------------------------- options -- */

/* name ∈ {"device" (ordinal) | "stream" (hipStream_t) | "rep_row" (0/1): before the first push;
 *         "max_groups" (group-table capacity bound, default 1<<26; the table itself grows with the rows pushed), "agg_mode" (n1k_agg_mode),
 *         "grid_blocks" / "block" / "rows_per_lane" / "lds_bytes" (launch tuning, 0 = auto),
 *         "fast" / "spec" / "wide" / "slabs" (0/1: kernel selection switches used by the ablation tests),
 *         "jit" (0 off, 1 = compile a specialised kernel for large batches of unregistered shapes, 2 = always),
 *         "jit_min_rows", "distinct_words" (0/1, before the first push: COUNT(DISTINCT) members that fit one word are
 *         de-duplicated by radix partition + LDS sets), "partition_min_rows" / "partition_probe_rows" / "partition_min_groups" / "partition_levels" (when AUTO takes the
 *         partitioned high-cardinality path: batches of at least min_rows (8 Mi) rows whose first probe_rows (512 Ki) rows
 *         bring at least min_groups (4096) new groups; forced number of passes), "topk_min_groups" (ORDER BY ... LIMIT: the device top-k filter runs from this many groups on, default 65536),
 *         "distinct_set_slots" / "distinct_levels" (LDS set size, forced number of partition passes: tests),
 *         "distinct_region_cap" (forced capacity of the specialised scan's hash regions: tests), "dedupe_block" (256 / 512 / 1024: tuning), "wide_values" (before the first push: how many distinct float / wide-integer group key
 *         values the handle can code, default 1<<20; 0 = such keys are N1K_UNSUPPORTED_DATA),
 *         "inject_failure" (tests of the multi-GPU failure rules, one shot, on the SENDING handle: the next exchange call
 *         pretends that 1 = its buffers, 2 = its partition / export, 3 = its receiving part failed)} */
n1k_status n1k_set_option(n1k_handle *h, const char *name, int64_t value);

/* ----------------------------------------------------------------- data ---- */

/*
 * Raw documents -> the plan's leaf columns.
 * Replaces: the per-row, per-field leaf accesses of the operators over a parsedValue — Field.Apply
 *           (expression/nav_field.go:134-160) -> parsedValue.Field (value/parsed.go:159-207: go_json FirstFind of the
 *           field in the raw bytes) -> value.NewValue typing (value/value.go:367-430: integer literals that fit
 *           int64 are INT, a float64 with no fraction folds to INT) — done once per batch.
 * Document d is bytes[offsets[d] .. offsets[d+1]) (one UTF-8 JSON value, the file datastore's <key>.json content,
 * datastore/file/file.go:732-749).  Leaf paths must be chains of field names; the first field of a name counts;
 * a field of a non-object is MISSING; strings and the canonical text of arrays / objects (sorted names, compact:
 * value/object.go:30-78) are interned in the handle's dictionary.  The columns (TAGGED64) belong to the handle
 * until the next extract / destroy.  Parsing is multi-threaded ("json_threads" option, default = the cores, <= 16).
 * N1K_INVALID names the first malformed document.  Needs no device.
 */
n1k_status n1k_extract_json(n1k_handle *h, uint64_t ndocs, const uint64_t *offsets, const char *bytes, n1k_batch *out);

/* n1k_extract_json + n1k_push_batch */
n1k_status n1k_push_json(n1k_handle *h, uint64_t ndocs, const uint64_t *offsets, const char *bytes);

/*
 * ≙ processItem() over a batch (execution/filter.go:49-61 +
 * execution/group_initial.go:56-100).  Host pointers; the call copies them to
 * the device before returning (cgo rule: no Go pointer is retained).
 */
n1k_status n1k_push_batch(n1k_handle *h, const n1k_batch *batch);

/* Same, but every pointer in the batch is a DEVICE pointer on the handle's
 * device (columns already resident in HBM); nothing is copied.  The call is
 * asynchronous on the handle's stream; the buffers must stay valid until
 * n1k_finish / n1k_sync. */
n1k_status n1k_push_device_batch(n1k_handle *h, const n1k_batch *batch);

/* One whole execution of the operator over a device-resident batch: n1k_reset + n1k_push_device_batch + n1k_finish in one
 * call (≙ reopen() + processItem()* + afterItems() of a re-opened operator, execution/base.go:119-135: what a prepared
 * statement executed again over the same resident columns does).  Same results and errors as the three calls. */
n1k_status n1k_run_device_batch(n1k_handle *h, const n1k_batch *batch, n1k_result *out);

/* Wait for all queued device work of the handle. */
n1k_status n1k_sync(n1k_handle *h);

/*
 * ≙ afterItems() of InitialGroup + the whole IntermediateGroup and FinalGroup
 * (execution/group_intermediate.go:56-104, group_final.go:55-118): merges,
 * finalises (ComputeFinal) and returns all groups; emits the default row when
 * there are no keys and no input (group_final.go:108-117).  The result memory
 * is owned by the handle and valid until the next finish/reset/destroy.
 */
n1k_status n1k_finish(n1k_handle *h, n1k_result *out);

n1k_status n1k_get_stats(const n1k_handle *h, n1k_stats *out);

/*
 * Build (but do not run) the plan-specialised scan kernel of this plan for the given column kinds through the
 * in-process compiler (hiprtc): what the engine does lazily for large batches of a plan shape that has no
 * ahead-of-time instantiation.  Needs no GPU; a prepared-statement cache (plan/prepared.go) would call it at
 * PREPARE time.  `log` receives the compiler output.  N1K_UNSUPPORTED when the shape is outside the bounded family
 * (the interpreter kernel then runs the plan).
 */
n1k_status n1k_jit_check(n1k_handle *h, const uint32_t *col_kinds, uint32_t ncols, char *log, size_t loglen);

/* ------------------------------------------------- multi-GPU (one per rank) -- */

/*
 * Hash-partition step of the 8-GPU path (no reference analogue: replaces the
 * Parallel -> IntermediateGroup fan-in, execution/exchange.go:161-251).
 * Evaluates the Filter on a device-resident batch and scatters every surviving
 * row's referenced columns into `nparts` (<= 64) regions by a hash of the group key VALUES.
 * Each column keeps its own layout, so the receiving rank can hand the received
 * buffers straight to n1k_push_device_batch of a handle built from the same plan
 * without its Filter.
 *
 * out_cols[c] are device buffers with room for `capacity_rows` rows per part
 * (part p starts at row p*capacity_rows); out_counts (device, nparts x u64)
 * receives the number of rows written to each part.
 */
n1k_status n1k_partition_device_batch(n1k_handle *h, const n1k_batch *batch, uint32_t nparts,
                                      uint64_t capacity_rows, const n1k_col *out_cols, uint64_t *out_counts);

/*
 * Partial groups (the reference's Initial -> Intermediate hand-over, algebra/aggregate.go:25-40).
 *
 * n1k_export_partials_device: write every group of the handle (packed key + raw accumulators) into `nparts`
 * regions of `out` (device memory) by a hash of the packed group key.  Region d (region_bytes = n1k_partial_region_bytes(h, capacity_groups)) is
 *     [count u64][verdict u64][keys: capacity x u64][accumulators: capacity x n1k_partial_words(h) x u64]
 * so that ONE all-to-all with equal splits moves all regions.  N1K_REGION_FULL when a region overflows; N1K_UNSUPPORTED
 * when the keys hold float / wide-integer values (their codes are local to the handle).
 *
 * n1k_export_partials_async: the same, ordered on the handle's stream with no host synchronisation.  The two
 * failures above are written into the `verdict` word of EVERY region instead (bit 0 overflow, bit 1 wide values),
 * so that each receiver learns them from the exchange itself: n1k_merge_partials_device then merges nothing and
 * the receiver's n1k_finish returns N1K_REGION_FULL / N1K_UNSUPPORTED — on all ranks alike, which lets them retry in step.
 *
 * n1k_merge_partials_device: ≙ CumulateIntermediate (execution/group_intermediate.go:91-101) over `nregions`
 * regions of that layout (device memory), e.g. what the all-to-all delivered.  The handle must have the same
 * plan, column kinds and dictionary as the exporters.  DISTINCT aggregates cannot be merged this way
 * (N1K_UNSUPPORTED): their sets would have to travel; use the row exchange instead.
 *
 * n1k_export_groups / n1k_merge_groups: the same through a host blob (one region), for merging operator copies
 * inside one process.
 */
uint32_t n1k_partial_words(const n1k_handle *h);
uint64_t n1k_partial_region_bytes(const n1k_handle *h, uint64_t capacity_groups);
n1k_status n1k_export_partials_device(n1k_handle *h, uint32_t nparts, uint64_t capacity_groups, void *out);
n1k_status n1k_export_partials_async(n1k_handle *h, uint32_t nparts, uint64_t capacity_groups, void *out);
n1k_status n1k_merge_partials_device(n1k_handle *h, uint32_t nregions, uint64_t capacity_groups, const void *in);
n1k_status n1k_export_groups(n1k_handle *h, const void **blob, size_t *len);

/*
 * The multi-GPU tail of ORDER BY ... OFFSET ... LIMIT: when groups are owned by several handles of the same plan and
 * dictionary (disjoint key ranges after the hash partition), each owner returns its first offset+limit rows and
 * rank 0 applies the plan's Order / Offset / Limit to their union (≙ execution/order.go:121-169 over the gathered rows).
 * Host only; `out` belongs to the handle like n1k_finish's.
 */
n1k_status n1k_order_rows(n1k_handle *h, uint64_t ngroups, const n1k_value *keys, const n1k_value *aggs, n1k_result *out);
n1k_status n1k_merge_groups(n1k_handle *h, const void *blob, size_t len);

/* ------------------------------------------ multi-GPU: RCCL collectives behind the ABI -- */

/*
 * One communicator per rank, one rank per GPU (no reference analogue: the reference is single-process and fans its
 * Parallel copies in through one in-memory queue, execution/exchange.go:161-251).  Rank 0 makes an id
 * (n1k_comm_unique_id, N1K_COMM_ID_BYTES bytes = ncclUniqueId), hands it to the other ranks by whatever channel the
 * host has, and every rank calls n1k_comm_create — a collective, like ncclCommInitRank.
 *
 * The exchange calls are collectives too: every rank makes the same call.  They are enqueued on the sending handle's
 * stream with no host synchronisation; the receiving handle's n1k_finish is the step's one wait.  Several exchanges may
 * be issued on one communicator per step (one per batch): each waits, on the device, for the owner's kernels that still
 * read the previous one's regions.
 *
 * Failures.  A rank never leaves its peers waiting in a collective:
 *   - whatever fails on a rank BEFORE the collective (a stopped operator, a batch that does not validate, buffers that
 *     cannot be allocated, the partition or the export itself) does not keep it from entering: it ships regions that
 *     hold only its n1k_status in their verdict word, returns that status from the exchange call, and every receiver's
 *     n1k_finish then fails with the same status;
 *   - what a sender's kernels find (a region that overflows: N1K_REGION_FULL; rows whose group key does not pack, values its
 *     Filter cannot order: N1K_UNSUPPORTED_DATA) travels in the same verdict word;
 *   in both cases the step is void on EVERY rank alike — n1k_failure_is_global(handle) says so after the failing call —
 *   and no rank goes on to n1k_gather_groups: all of them retry (N1K_REGION_FULL: with a larger capacity) or give up in step;
 *   - what fails on a rank AFTER the collective (the owner's InitialGroup / merge / n1k_finish over what it received) is
 *     that rank's alone: it enters n1k_gather_groups_status with its status, and every rank learns it there.
 * n1k_rows_step / n1k_partials_step run one whole step by these rules.  Not carried (the peers are left waiting; bound
 * such a wait with the host's own watchdog): a rank without a usable device, a rank that cannot allocate even one
 * region, and the first n1k_exchange_rows of a handle whose batch does not have the plan's column count / kinds (the
 * region size is a function of the column kinds; N1K_INVALID).
 *
 * n1k_exchange_partials: per-GPU partial groups (≙ the Initial -> Intermediate hand-over, algebra/aggregate.go:25-40):
 *   the sender's groups are exported (n1k_export_partials_async), moved by ONE collective — all-gather when `gathered`
 *   (every rank merges every rank's groups and ends with the whole result), else all-to-all of regions hash-partitioned
 *   on the group key (each key is merged and finished by one owner) — and merged into the receiver
 *   (n1k_merge_partials_device).  Not for plans with DISTINCT aggregates (sets do not travel).
 * n1k_exchange_rows: the sender evaluates its Filter over `batch` (device-resident) and scatters the survivors' columns
 *   by a hash of the group key VALUES into one packed region per destination rank ([count][verdict] header + every
 *   column's rows); ONE all-to-all moves counts and rows together; the receiver — a handle of the same plan without the
 *   Filter — runs InitialGroup over each received region, whose row count stays on the device.  Each group then
 *   lives on exactly one rank: COUNT(DISTINCT) needs no set merge.  capacity_rows: rows a region takes — like
 *   capacity_groups of n1k_exchange_partials it is part of the collective's shape and MUST be the same on every rank
 *   (agree on it with n1k_comm_max_u64, e.g. from the largest shard's row count; the regions are split into 8
 *   sub-regions with their own counts, which a receiver aggregates as one segmented batch).
 * n1k_gather_groups: every rank's finished groups (`local`, its n1k_finish result) to every rank with one all-gather of
 *   fixed-size slots (the slot size only changes on counts every rank reads in the gathered headers), then the handle's Order / Offset / Limit / projection over the union
 *   (≙ n1k_order_rows).  `out` belongs to the handle like n1k_finish's.
 */
#define N1K_COMM_ID_BYTES 128
typedef struct n1k_comm n1k_comm;
n1k_status n1k_comm_unique_id(void *id);
n1k_status n1k_comm_create(const void *id, int rank, int world, int device, n1k_comm **out);
/* Loopback transport: `world` communicators (out[0 .. world)) whose ranks are THREADS of this process sharing one device;
 * every collective is a rendezvous plus device-to-device copies.  For tests of the world_size > 1 paths on a single GPU
 * (RCCL refuses two ranks on one device) and for hosts that run several operator copies per GPU.  Each rank must be driven
 * from its own thread (a collective blocks until every rank has entered it). */
n1k_status n1k_comm_create_loopback(int world, int device, n1k_comm **out);
void n1k_comm_destroy(n1k_comm *c);
const char *n1k_comm_last_error(const n1k_comm *c);
int n1k_comm_rank(const n1k_comm *c);
int n1k_comm_world(const n1k_comm *c);
/* largest `value` over the ranks (a collective; waits): how the ranks agree on a region capacity from what each one holds */
n1k_status n1k_comm_max_u64(n1k_comm *c, n1k_handle *h, uint64_t value, uint64_t *out);
n1k_status n1k_exchange_partials(n1k_comm *c, n1k_handle *sender, n1k_handle *receiver, uint64_t capacity_groups, int gathered);
n1k_status n1k_exchange_rows(n1k_comm *c, n1k_handle *sender, const n1k_batch *batch, n1k_handle *receiver, uint64_t capacity_rows);
/* The same with one capacity PER DESTINATION (capacity_rows[world], the same vector on every rank): every sender's region for
 * owner d holds capacity_rows[d] rows, so an owner that receives more than its share (skewed keys) only enlarges the regions
 * sent to IT, not all world x world of them.  n1k_exchange_sent_rows: the rows this rank's last exchange wrote per destination
 * (waits for the stream); the ranks size the next steps' regions from the largest over the senders, per destination
 * (n1k_comm_max_u64_v: element-wise n1k_comm_max_u64 over n <= 64 values). */
n1k_status n1k_exchange_rows_v(n1k_comm *c, n1k_handle *sender, const n1k_batch *batch, n1k_handle *receiver,
                               const uint64_t *capacity_rows);
n1k_status n1k_exchange_sent_rows(n1k_comm *c, n1k_handle *sender, uint64_t *out);
n1k_status n1k_comm_max_u64_v(n1k_comm *c, n1k_handle *h, uint32_t n, const uint64_t *values, uint64_t *out);
n1k_status n1k_gather_groups(n1k_comm *c, n1k_handle *h, const n1k_result *local, n1k_result *out);
/* The same, carrying every rank's verdict on its own step: a rank whose n1k_finish failed (data its shard alone holds, say)
 * still enters the collective with local_status != N1K_OK (local may be NULL) instead of leaving its peers waiting in it;
 * every rank gets the first failing rank's status in *worst_status (N1K_OK: `out` holds the gathered groups). */
n1k_status n1k_gather_groups_status(n1k_comm *c, n1k_handle *h, const n1k_result *local, int local_status, n1k_result *out,
                                    int *worst_status);
/* 1 when the last failure reported on this handle (by an exchange call on a sender, by n1k_finish on a receiver) was told
 * to / learnt from the verdict words of an exchange: every rank's step failed alike and no rank enters the gather. */
int n1k_failure_is_global(const n1k_handle *h);
/* One whole step of the row exchange in one call: n1k_reset(receiver), n1k_reset(sender), n1k_exchange_rows,
 * n1k_finish(receiver), n1k_gather_groups_status(merger) — what a host runs per query.  A failure with
 * n1k_failure_is_global(receiver) or n1k_failure_is_global(sender): the step is void on every rank alike and no gather took
 * place (N1K_REGION_FULL: a region overflowed — enlarge capacity_rows on every rank and repeat); otherwise the rank's own status,
 * with *worst_status as n1k_gather_groups_status reports it. */
n1k_status n1k_rows_step(n1k_comm *c, n1k_handle *sender, const n1k_batch *batch, n1k_handle *receiver, n1k_handle *merger,
                         uint64_t capacity_rows, n1k_result *out, int *worst_status);
n1k_status n1k_rows_step_v(n1k_comm *c, n1k_handle *sender, const n1k_batch *batch, n1k_handle *receiver, n1k_handle *merger,
                           const uint64_t *capacity_rows, n1k_result *out, int *worst_status); /* capacities per destination */
/* The same for partial groups: n1k_reset(receiver), n1k_reset(sender), n1k_push_device_batch(sender, batch) (InitialGroup
 * over the shard), n1k_exchange_partials, n1k_finish(receiver) and — unless `gathered`, where every rank merged every rank's
 * groups and `receiver` (a handle that carries the plan's grouped tail) holds the result — the gather through `merger`. */
n1k_status n1k_partials_step(n1k_comm *c, n1k_handle *sender, const n1k_batch *batch, n1k_handle *receiver, n1k_handle *merger,
                             uint64_t capacity_groups, int gathered, n1k_result *out, int *worst_status);

/* ------------------------------------------------------------- utilities -- */

/* Device-side synthetic column generator of SURVEY.md §8(d) (bench/test input;
 * the same generator exists on the CPU in oracle/ for parity).  All outputs are
 * device pointers with room for nrows elements; any may be NULL. */
typedef struct n1k_synth_spec {
    uint64_t seed;
    uint64_t first_row;  /* global ordinal of row 0 of this call (sharding) */
    uint64_t nrows;
    uint64_t total_rows; /* N of the whole data set (user_id range = N/10) */
    uint32_t k_cat;      /* number of categories */
    uint32_t zipf;       /* 0 = uniform, 1 = Zipf(s=1) via cdf */
    const double *cat_cdf; /* device pointer, k_cat entries, only for zipf */
} n1k_synth_spec;
n1k_status n1k_synth_columns(int device, void *stream, const n1k_synth_spec *spec, uint32_t *cat_codes,
                             uint8_t *price_tags, uint64_t *price_payload, uint8_t *user_tags,
                             uint64_t *user_payload, uint8_t *region_tags, uint64_t *region_payload);

/* The same rows as raw JSON documents (bench / test input for n1k_push_json): from HOST copies of the arrays
 * n1k_synth_columns filled, document i = {"id":"d<i>","cat":"cat_<c>","price":<p>,"user_id":<u>,"region_id":<r>,"pad":"x..."}
 * (a MISSING value leaves its field out, NULL prints null, floats print their shortest round-trip digits; `pad` x's).
 * offsets: nrows + 1 entries.  N1K_OOM: `cap` bytes do not hold them (*used then says how many it takes). */
n1k_status n1k_synth_documents(uint64_t nrows, uint64_t first_id, const uint32_t *cat_codes, const uint8_t *price_tags,
                               const uint64_t *price_payload, const uint64_t *user_payload, const uint64_t *region_payload,
                               uint32_t pad, char *bytes, size_t cap, uint64_t *offsets, size_t *used);

int n1k_abi_version(void);
/* number of visible HIP devices (0 when none); never initialises a context */
int n1k_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* N1K_H */
