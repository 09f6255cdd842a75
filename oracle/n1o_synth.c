/*
 * n1o_synth.c — CPU generator of the synthetic document columns.
 * TEST INFRASTRUCTURE (see n1o.h).  The HIP generator in
 * query_amd/csrc/synth.hip produces bit-identical columns; tests compare them.
 *
 * Document shape: {"id":"d<i>","cat":"cat_<c>","price":<p>,"user_id":<u>,"region_id":<r>}
 *   cat       dictionary code c in [0,k_cat)  (dictionary: "cat_0".."cat_{k-1}", then "n/a" = code k_cat)
 *   price     80 % two-decimal float in [0,100) (integral ones fold to INT as value.NewValue does,
 *             value/value.go:377-382), 18 % integer in [0,100], 1 % NULL, 0.5 % MISSING, 0.5 % string "n/a"
 *   user_id   integer in [0, max(1,total_rows/10))
 *   region_id integer in [0,64)
 * Randomness: counter based, splitmix64(seed + 8*row + field).
 */
#include "n1o.h"
#include <string.h>

static inline uint64_t splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void n1o_zipf_cdf(uint32_t k, double *cdf) {
    double h = 0.0;
    for (uint32_t i = 0; i < k; i++) h += 1.0 / (double)(i + 1);
    double acc = 0.0;
    for (uint32_t i = 0; i < k; i++) {
        acc += (1.0 / (double)(i + 1)) / h;
        cdf[i] = acc;
    }
    if (k) cdf[k - 1] = 1.0;
}

void n1o_synth_columns(uint64_t seed, uint64_t first_row, uint64_t nrows, uint64_t total_rows, uint32_t k_cat,
                       const double *cat_cdf, uint32_t *cat_codes, uint8_t *price_tags, uint64_t *price_payload,
                       uint8_t *user_tags, uint64_t *user_payload, uint8_t *region_tags,
                       uint64_t *region_payload) {
    uint64_t urange = total_rows / 10;
    if (urange == 0) urange = 1;
    for (uint64_t j = 0; j < nrows; j++) {
        uint64_t i = first_row + j;
        uint64_t base = seed + i * 8ull;
        if (cat_codes) {
            uint64_t r = splitmix64(base + 0);
            uint32_t c;
            if (cat_cdf) {
                double u = (double)(r >> 11) * 0x1.0p-53;
                uint32_t lo = 0, hi = k_cat; /* first index with cdf[idx] > u */
                while (lo < hi) {
                    uint32_t mid = lo + ((hi - lo) >> 1);
                    if (cat_cdf[mid] > u) hi = mid; else lo = mid + 1;
                }
                c = lo < k_cat ? lo : k_cat - 1;
            } else {
                c = (uint32_t)(((unsigned __int128)r * (unsigned __int128)k_cat) >> 64);
            }
            cat_codes[j] = c;
        }
        if (price_tags || price_payload) {
            uint64_t sel = splitmix64(base + 1) % 1000ull;
            uint64_t r2 = splitmix64(base + 2);
            uint8_t tag;
            uint64_t pay = 0;
            if (sel < 800) {
                uint64_t cents = r2 % 10000ull;
                if (cents % 100ull == 0) {
                    tag = N1K_T_INT;
                    pay = cents / 100ull;
                } else {
                    double p = (double)cents / 100.0;
                    tag = N1K_T_FLOAT;
                    memcpy(&pay, &p, 8);
                }
            } else if (sel < 980) {
                tag = N1K_T_INT;
                pay = r2 % 101ull;
            } else if (sel < 990) {
                tag = N1K_T_NULL;
            } else if (sel < 995) {
                tag = N1K_T_MISSING;
            } else {
                tag = N1K_T_STRING;
                pay = k_cat; /* code of "n/a" */
            }
            if (price_tags) price_tags[j] = tag;
            if (price_payload) price_payload[j] = pay;
        }
        if (user_tags) user_tags[j] = N1K_T_INT;
        if (user_payload) user_payload[j] = splitmix64(base + 3) % urange;
        if (region_tags) region_tags[j] = N1K_T_INT;
        if (region_payload) region_payload[j] = splitmix64(base + 4) % 64ull;
    }
}
