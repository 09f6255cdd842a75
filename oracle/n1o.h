/*
 * n1o.h — CPU oracle for the Filter -> Group -> Aggregate path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked, imported or
 * called by the product (query_amd/, include/); only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may use it, and only as the
 * checker / the timed CPU baseline.
 *
 * The oracle is a restatement, in plain C, of the reference's Go algorithm
 * (row at a time, boxed tagged values, canonical-JSON group keys, one map per
 * Parallel copy, serial Intermediate/Final); every function cites the reference
 * file:line it follows.  The reference itself (Go, GOPATH-era, generated
 * parser absent) cannot be built in this image, so the oracle is pinned by the
 * reference's own golden case files (tests/golden/).
 *
 * Column data uses the layout of include/n1k.h (data format only).
 */
#ifndef N1O_H
#define N1O_H
#include <stddef.h>
#include <stdint.h>
#include "../include/n1k.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct n1o_table {
    uint64_t nrows;
    uint32_t ncols;
    uint32_t dict_n;
    const char *const *names; /* leaf path text per column, e.g. "(`default`.`price`)" */
    const n1k_col *cols;
    const uint64_t *dict_offsets; /* dict_n + 1 */
    const char *dict_bytes;
} n1o_table;

typedef struct n1o_result {
    uint64_t ngroups;
    uint32_t nkeys;
    uint32_t naggs;
    n1k_value *keys; /* [ngroups][nkeys] */
    n1k_value *aggs; /* [ngroups][naggs] */
    uint64_t nselected;
    uint64_t *selected;
    uint64_t rows_filtered_in; /* rows that passed the filter */
    double seconds;            /* wall time of the run proper (no parsing) */
    char err[512];
    /* values built by the run (ARRAY_AGG arrays): string i is extra_bytes[extra_offsets[i] .. extra_offsets[i+1]),
     * referred to by the code dict_n + i */
    char *extra_bytes;
    uint64_t *extra_offsets;
    uint32_t extra_n;
} n1o_result;

/*
 * condition: Filter condition text or NULL.  has_group: 0 = Filter-only
 * (selected rows returned), 1 = run Initial/Intermediate/Final group.
 * threads: number of Parallel copies (execution/parallel.go:52-75).
 * Returns 0 on success, non-zero with out->err set otherwise.
 */
int n1o_run(const char *condition, const char *const *keys, uint32_t nkeys, const char *const *aggs,
            uint32_t naggs, int has_group, const n1o_table *t, int threads, n1o_result *out);
void n1o_free_result(n1o_result *r);

/* Evaluate one expression for every row (unit tests of a5-a8): out[nrows]. */
int n1o_eval(const char *expr, const n1o_table *t, n1k_value *out, char *err, size_t errlen);

/* CountScan over the file datastore: number of directory entries
 * (datastore/file/file.go:296-302).  -1 on error. */
int64_t n1o_count_scan(const char *dir);

/* Synthetic columns of SURVEY.md §8(d); any output may be NULL. */
void n1o_synth_columns(uint64_t seed, uint64_t first_row, uint64_t nrows, uint64_t total_rows, uint32_t k_cat,
                       const double *cat_cdf /* NULL = uniform */, uint32_t *cat_codes, uint8_t *price_tags,
                       uint64_t *price_payload, uint8_t *user_tags, uint64_t *user_payload,
                       uint8_t *region_tags, uint64_t *region_payload);
/* Zipf(s=1) cdf over k categories, as used by both generators. */
void n1o_zipf_cdf(uint32_t k, double *cdf);

#ifdef __cplusplus
}
#endif
#endif
