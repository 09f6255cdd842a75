/*
 * cabi_client.c — a plain-C client of libn1k.so that plays the call sequence of the cgo operator shown in
 * INTEGRATION.md (the reference side of the boundary: execution/execution.go:26-64 Operator life cycle,
 * execution/parallel.go:67-83 one Copy() of the operator per goroutine, execution/base.go:313-338 SendStop from
 * another goroutine).  Test infrastructure: built with gcc and run by tests/test_cabi_client.py.
 *
 *   cabi_client <libn1k.so> symbols
 *       dlopen the library, resolve every symbol the client uses, create and destroy an operator (no device needed)
 *   cabi_client <libn1k.so> run <plan.json> <data.bin> <out.txt> <batch_rows>
 *       1. two operator copies (n1k_create x 2, the dictionary interned into both), one OS thread each:
 *          row-at-a-time staging of its half of the rows into C buffers (what processItem does per AnnotatedValue),
 *          n1k_push_batch every <batch_rows> rows;
 *       2. a third copy is pushed to in a loop by one thread while another calls n1k_stop: the loop must end with
 *          N1K_STOPPED (≙ processItem returning false after SendStop), and n1k_reset makes the copy usable again;
 *       3. the second copy's groups are merged into the first (n1k_export_groups / n1k_merge_groups ≙ the fan-in of
 *          the Parallel copies into IntermediateGroup), n1k_finish, and the groups are written to <out.txt>.
 *
 * data.bin: u64 nrows, u32 ncols, u32 ndict; per column u32 kind; per column its arrays (TAGGED64: u8 tags[nrows],
 * u64 payload[nrows]; DICT32: u32 codes[nrows]); the dictionary as ndict x (u32 length, bytes).
 */
#include <dlfcn.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../include/n1k.h"

#define FN(ret, name, ...) static ret (*p_##name)(__VA_ARGS__)
FN(n1k_status, n1k_create, const char *, size_t, n1k_handle **);
FN(void, n1k_destroy, n1k_handle *);
FN(n1k_status, n1k_reset, n1k_handle *);
FN(void, n1k_stop, n1k_handle *);
FN(const char *, n1k_last_error, const n1k_handle *);
FN(const char *, n1k_create_error, void);
FN(uint32_t, n1k_num_columns, const n1k_handle *);
FN(const char *, n1k_column_path, const n1k_handle *, uint32_t);
FN(uint32_t, n1k_num_aggregates, const n1k_handle *);
FN(const char *, n1k_aggregate_name, const n1k_handle *, uint32_t);
FN(n1k_status, n1k_dict_intern, n1k_handle *, uint32_t, const uint64_t *, const char *, uint32_t *);
FN(n1k_status, n1k_dict_get, const n1k_handle *, uint32_t, const char **, size_t *);
FN(n1k_status, n1k_push_batch, n1k_handle *, const n1k_batch *);
FN(n1k_status, n1k_finish, n1k_handle *, n1k_result *);
FN(n1k_status, n1k_export_groups, n1k_handle *, const void **, size_t *);
FN(n1k_status, n1k_merge_groups, n1k_handle *, const void *, size_t);
FN(n1k_status, n1k_get_stats, const n1k_handle *, n1k_stats *);
FN(int, n1k_abi_version, void);

static void *must_sym(void *lib, const char *name) {
    void *p = dlsym(lib, name);
    if (!p) {
        fprintf(stderr, "missing symbol %s\n", name);
        exit(2);
    }
    return p;
}
#define LOAD(name) *(void **)(&p_##name) = must_sym(lib, #name)

static void die(const char *what, n1k_handle *h, n1k_status st) {
    fprintf(stderr, "%s: status %d: %s\n", what, (int)st, h ? p_n1k_last_error(h) : p_n1k_create_error());
    exit(1);
}

/* ---- input ---- */
typedef struct {
    uint64_t nrows;
    uint32_t ncols, ndict;
    uint32_t *kind;
    uint8_t **tags;
    uint64_t **payload;
    uint32_t **codes;
    uint64_t *dict_off;
    char *dict_bytes;
} table;

static void read_all(FILE *f, void *dst, size_t n) {
    if (n && fread(dst, 1, n, f) != n) {
        fprintf(stderr, "short read\n");
        exit(2);
    }
}

static table load_table(const char *path) {
    table t;
    memset(&t, 0, sizeof t);
    FILE *f = fopen(path, "rb");
    if (!f) {
        perror(path);
        exit(2);
    }
    read_all(f, &t.nrows, 8);
    read_all(f, &t.ncols, 4);
    read_all(f, &t.ndict, 4);
    t.kind = calloc(t.ncols, sizeof *t.kind);
    t.tags = calloc(t.ncols, sizeof *t.tags);
    t.payload = calloc(t.ncols, sizeof *t.payload);
    t.codes = calloc(t.ncols, sizeof *t.codes);
    read_all(f, t.kind, 4 * (size_t)t.ncols);
    for (uint32_t c = 0; c < t.ncols; c++) {
        if (t.kind[c] == N1K_COL_DICT32) {
            t.codes[c] = malloc(4 * t.nrows + 4);
            read_all(f, t.codes[c], 4 * t.nrows);
        } else {
            t.tags[c] = malloc(t.nrows + 1);
            t.payload[c] = malloc(8 * t.nrows + 8);
            read_all(f, t.tags[c], t.nrows);
            read_all(f, t.payload[c], 8 * t.nrows);
        }
    }
    t.dict_off = calloc((size_t)t.ndict + 1, 8);
    size_t cap = 1 << 16, used = 0;
    t.dict_bytes = malloc(cap);
    for (uint32_t i = 0; i < t.ndict; i++) {
        uint32_t len;
        read_all(f, &len, 4);
        if (used + len + 1 > cap) {
            while (used + len + 1 > cap) cap *= 2;
            t.dict_bytes = realloc(t.dict_bytes, cap);
        }
        read_all(f, t.dict_bytes + used, len);
        used += len;
        t.dict_off[i + 1] = used;
    }
    fclose(f);
    return t;
}

/* ---- one operator copy fed row by row from its own thread ---- */
typedef struct {
    n1k_handle *h;
    const table *t;
    uint64_t first, last, batch_rows;
    n1k_status status;
    uint64_t batches;
} feeder;

typedef struct {
    uint8_t **tags;
    uint64_t **payload;
    uint32_t **codes;
    n1k_col *cols;
} staging;

static staging make_staging(const table *t, uint64_t rows) {
    staging s;
    s.tags = calloc(t->ncols, sizeof *s.tags);
    s.payload = calloc(t->ncols, sizeof *s.payload);
    s.codes = calloc(t->ncols, sizeof *s.codes);
    s.cols = calloc(t->ncols, sizeof *s.cols);
    for (uint32_t c = 0; c < t->ncols; c++) {
        s.cols[c].kind = t->kind[c];
        if (t->kind[c] == N1K_COL_DICT32) {
            s.codes[c] = malloc(4 * rows + 4);
            s.cols[c].codes = s.codes[c];
        } else {
            s.tags[c] = malloc(rows + 1);
            s.payload[c] = malloc(8 * rows + 8);
            s.cols[c].tags = s.tags[c];
            s.cols[c].payload = s.payload[c];
        }
    }
    return s;
}

static n1k_status flush(n1k_handle *h, const table *t, staging *s, uint64_t rows) {
    n1k_batch b;
    memset(&b, 0, sizeof b);
    b.nrows = rows;
    b.ncols = t->ncols;
    b.cols = s->cols;
    return p_n1k_push_batch(h, &b); /* copies the staging buffers before it returns (cgo rule) */
}

static void *feed(void *arg) {
    feeder *f = arg;
    const table *t = f->t;
    staging s = make_staging(t, f->batch_rows);
    uint64_t staged = 0;
    f->status = N1K_OK;
    for (uint64_t r = f->first; r < f->last; r++) { /* ≙ processItem: one row's leaf values into the staging columns */
        for (uint32_t c = 0; c < t->ncols; c++) {
            if (t->kind[c] == N1K_COL_DICT32) s.codes[c][staged] = t->codes[c][r];
            else {
                s.tags[c][staged] = t->tags[c][r];
                s.payload[c][staged] = t->payload[c][r];
            }
        }
        if (++staged == f->batch_rows) {
            f->status = flush(f->h, t, &s, staged);
            f->batches++;
            staged = 0;
            if (f->status != N1K_OK) return NULL;
        }
    }
    if (staged) { /* ≙ afterItems: the last partial batch */
        f->status = flush(f->h, t, &s, staged);
        f->batches++;
    }
    return NULL;
}

/* pushes the same rows again and again until the operator is stopped */
static void *feed_until_stopped(void *arg) {
    feeder *f = arg;
    for (int i = 0; i < 100000; i++) {
        feeder once = *f;
        feed(&once);
        f->batches += once.batches;
        f->status = once.status;
        if (once.status != N1K_OK) return NULL;
    }
    return NULL;
}

static n1k_handle *make_operator(const char *plan, const table *t) {
    n1k_handle *h = NULL;
    n1k_status st = p_n1k_create(plan, strlen(plan), &h);
    if (st != N1K_OK) die("n1k_create", NULL, st);
    if (p_n1k_num_columns(h) != t->ncols) {
        fprintf(stderr, "plan needs %u columns, data has %u\n", p_n1k_num_columns(h), t->ncols);
        exit(1);
    }
    if (t->ndict) {
        uint32_t *codes = calloc(t->ndict, 4);
        st = p_n1k_dict_intern(h, t->ndict, t->dict_off, t->dict_bytes, codes);
        if (st != N1K_OK) die("n1k_dict_intern", h, st);
        for (uint32_t i = 0; i < t->ndict; i++)
            if (codes[i] != i) {
                fprintf(stderr, "dictionary interned right after create must keep code == index\n");
                exit(1);
            }
        free(codes);
    }
    return h;
}

static void print_value(FILE *o, const n1k_handle *h, const n1k_value *v) {
    switch (v->tag) {
        case N1K_T_MISSING: fprintf(o, "M"); break;
        case N1K_T_NULL: fprintf(o, "N"); break;
        case N1K_T_FALSE: fprintf(o, "F"); break;
        case N1K_T_TRUE: fprintf(o, "T"); break;
        case N1K_T_INT: fprintf(o, "I%lld", (long long)v->v.i); break;
        case N1K_T_FLOAT: fprintf(o, "D%.17g", v->v.f); break;
        default: {
            const char *p = NULL;
            size_t n = 0;
            if (p_n1k_dict_get(h, (uint32_t)v->v.code, &p, &n) != N1K_OK) {
                fprintf(stderr, "n1k_dict_get failed\n");
                exit(1);
            }
            fprintf(o, "%c", v->tag == N1K_T_STRING ? 'S' : (v->tag == N1K_T_ARRAY ? 'A' : 'O'));
            for (size_t i = 0; i < n; i++) fprintf(o, "%02x", (unsigned char)p[i]);
        }
    }
}

int main(int argc, char **argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s <libn1k.so> symbols | run <plan.json> <data.bin> <out.txt> <batch_rows>\n", argv[0]);
        return 2;
    }
    void *lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!lib) {
        fprintf(stderr, "dlopen: %s\n", dlerror());
        return 2;
    }
    LOAD(n1k_create); LOAD(n1k_destroy); LOAD(n1k_reset); LOAD(n1k_stop); LOAD(n1k_last_error); LOAD(n1k_create_error);
    LOAD(n1k_num_columns); LOAD(n1k_column_path); LOAD(n1k_num_aggregates); LOAD(n1k_aggregate_name); LOAD(n1k_dict_intern);
    LOAD(n1k_dict_get); LOAD(n1k_push_batch); LOAD(n1k_finish); LOAD(n1k_export_groups); LOAD(n1k_merge_groups);
    LOAD(n1k_get_stats); LOAD(n1k_abi_version);
    if (p_n1k_abi_version() != N1K_ABI_VERSION) {
        fprintf(stderr, "ABI version %d, header says %d\n", p_n1k_abi_version(), N1K_ABI_VERSION);
        return 1;
    }
    if (!strcmp(argv[2], "symbols")) {
        const char *plan = "{\"#operator\":\"InitialGroup\",\"aggregates\":[\"count(*)\"],\"group_keys\":[\"(`d`.`k`)\"]}";
        n1k_handle *h = NULL;
        n1k_status st = p_n1k_create(plan, strlen(plan), &h);
        if (st != N1K_OK) die("n1k_create", NULL, st);
        if (p_n1k_num_columns(h) != 1 || strcmp(p_n1k_column_path(h, 0), "(`d`.`k`)") || p_n1k_num_aggregates(h) != 1 ||
            strcmp(p_n1k_aggregate_name(h, 0), "count(*)")) {
            fprintf(stderr, "binding calls disagree with the plan\n");
            return 1;
        }
        n1k_handle *bad = NULL;
        st = p_n1k_create("{\"#operator\":\"Fetch\"}", 21, &bad);
        if (st == N1K_OK || bad != NULL) {
            fprintf(stderr, "a plan outside the path must be refused\n");
            return 1;
        }
        p_n1k_stop(h);
        p_n1k_destroy(h);
        printf("symbols ok\n");
        return 0;
    }
    if (strcmp(argv[2], "run") || argc < 7) return 2;
    FILE *pf = fopen(argv[3], "rb");
    if (!pf) {
        perror(argv[3]);
        return 2;
    }
    static char plan[1 << 16];
    size_t pn = fread(plan, 1, sizeof plan - 1, pf);
    plan[pn] = 0;
    fclose(pf);
    table t = load_table(argv[4]);
    const uint64_t batch_rows = strtoull(argv[6], NULL, 10);

    /* 1. two operator copies, two OS threads (≙ Parallel.RunOnce, execution/parallel.go:67-73) */
    n1k_handle *h[3];
    for (int i = 0; i < 3; i++) h[i] = make_operator(plan, &t);
    feeder f[2];
    pthread_t th[2];
    for (int i = 0; i < 2; i++) {
        memset(&f[i], 0, sizeof f[i]);
        f[i].h = h[i];
        f[i].t = &t;
        f[i].first = t.nrows * (uint64_t)i / 2;
        f[i].last = t.nrows * (uint64_t)(i + 1) / 2;
        f[i].batch_rows = batch_rows;
        pthread_create(&th[i], NULL, feed, &f[i]);
    }
    /* 2. meanwhile: a third copy that is stopped from this thread while its own thread is pushing */
    feeder fs;
    memset(&fs, 0, sizeof fs);
    fs.h = h[2];
    fs.t = &t;
    fs.first = 0;
    fs.last = t.nrows < 4 * batch_rows ? t.nrows : 4 * batch_rows;
    fs.batch_rows = batch_rows;
    pthread_t ths;
    pthread_create(&ths, NULL, feed_until_stopped, &fs);
    struct timespec nap = {0, 30 * 1000 * 1000};
    nanosleep(&nap, NULL);
    p_n1k_stop(h[2]); /* ≙ SendStop from another goroutine (execution/base.go:313-338) */
    pthread_join(ths, NULL);
    for (int i = 0; i < 2; i++) pthread_join(th[i], NULL);
    if (fs.status != N1K_STOPPED) {
        fprintf(stderr, "the stopped operator ended with status %d after %llu batches, expected N1K_STOPPED\n", (int)fs.status,
                (unsigned long long)fs.batches);
        return 1;
    }
    n1k_result res;
    if (p_n1k_finish(h[2], &res) != N1K_STOPPED) {
        fprintf(stderr, "n1k_finish of a stopped operator must report N1K_STOPPED\n");
        return 1;
    }
    n1k_status st = p_n1k_reset(h[2]); /* ≙ reopen: usable again */
    if (st != N1K_OK) die("n1k_reset", h[2], st);
    fs.status = N1K_OK;
    fs.batches = 0;
    feed(&fs);
    if (fs.status != N1K_OK) die("push after reset", h[2], fs.status);
    st = p_n1k_finish(h[2], &res);
    if (st != N1K_OK) die("finish after reset", h[2], st);
    for (int i = 0; i < 2; i++)
        if (f[i].status != N1K_OK) die("n1k_push_batch", h[i], f[i].status);

    /* 3. fan-in of the copies (≙ IntermediateGroup over the Parallel copies' partial groups) and FinalGroup */
    const void *blob = NULL;
    size_t blen = 0;
    st = p_n1k_export_groups(h[1], &blob, &blen);
    if (st != N1K_OK) die("n1k_export_groups", h[1], st);
    st = p_n1k_merge_groups(h[0], blob, blen);
    if (st != N1K_OK) die("n1k_merge_groups", h[0], st);
    st = p_n1k_finish(h[0], &res);
    if (st != N1K_OK) die("n1k_finish", h[0], st);
    n1k_stats s0, s1;
    p_n1k_get_stats(h[0], &s0);
    p_n1k_get_stats(h[1], &s1);
    FILE *o = fopen(argv[5], "w");
    if (!o) {
        perror(argv[5]);
        return 2;
    }
    fprintf(o, "# groups %llu keys %u aggs %u rows_in %llu+%llu batches %llu+%llu stopped_after %llu\n", (unsigned long long)res.ngroups,
            res.nkeys, res.naggs, (unsigned long long)s0.rows_in, (unsigned long long)s1.rows_in, (unsigned long long)f[0].batches,
            (unsigned long long)f[1].batches, (unsigned long long)fs.batches);
    for (uint64_t g = 0; g < res.ngroups; g++) {
        for (uint32_t k = 0; k < res.nkeys; k++) {
            print_value(o, h[0], &res.keys[g * res.nkeys + k]);
            fputc(' ', o);
        }
        fputc('|', o);
        for (uint32_t a = 0; a < res.naggs; a++) {
            fputc(' ', o);
            print_value(o, h[0], &res.aggs[g * res.naggs + a]);
        }
        fputc('\n', o);
    }
    fclose(o);
    for (int i = 0; i < 3; i++) p_n1k_destroy(h[i]);
    printf("client ok: %llu groups\n", (unsigned long long)res.ngroups);
    return 0;
}
