/*
 * cortex_synth.c — CPU twin of the on-device synthetic embedding generator
 * (cortex_amd/csrc/synth.hip).  TEST INFRASTRUCTURE ONLY.
 *
 * Data model: clustered mixture of unit rows.
 *   C = max(1, N/50) centres ~ N(0,I) normalised;
 *   row = normalise(centre + sigma * g / sqrt(d)), sigma in {0.25, 0.42, 0.60}
 *   with weights {0.2, 0.4, 0.4}; rows r with r%1000==999 are exact duplicates
 *   of an earlier base row, r%1000==998 near-duplicates (sigma' = 0.045,
 *   cos ~ 0.999).  Optional per-row scale in [0.5, 2) for the un-normalised
 *   fixture.  Counter-based Philox4x32-10, key = seed, so any row can be
 *   generated independently and the CPU and GPU produce the same bits:
 *   "gaussians" are Irwin-Hall sums of four 24-bit uniforms (integer exact),
 *   every float op is a single IEEE operation (no FMA contraction; build with
 *   -ffp-contract=off), and the norm uses one fixed summation order: 64
 *   lane-strided sequential partial sums followed by a 32/16/8/4/2/1 fold.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

static void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                          uint32_t k0, uint32_t k1, uint32_t out[4]) {
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
        uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += PHILOX_W0; k1 += PHILOX_W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* approx N(0,1): sum of four 24-bit uniforms, centred, scaled by 1/std */
static float gauss(uint64_t seed, uint64_t idx, uint32_t stream) {
    uint32_t x[4];
    philox4x32_10((uint32_t)idx, (uint32_t)(idx >> 32), stream, 0u,
                  (uint32_t)seed, (uint32_t)(seed >> 32), x);
    int32_t s = (int32_t)((x[0] >> 8) + (x[1] >> 8) + (x[2] >> 8) + (x[3] >> 8)) - 33554430;
    return (float)s * 1.0323829e-07f; /* 1 / (2^24 / sqrt(3)) */
}

/* the fixed-order sum of squares shared with the GPU kernel */
static float sumsq_fixed(const float *v, size_t d) {
    float part[64];
    for (int l = 0; l < 64; l++) {
        float a = 0.0f;
        for (size_t j = l; j < d; j += 64) { float p = v[j] * v[j]; a = a + p; }
        part[l] = a;
    }
    for (int s = 32; s >= 1; s >>= 1)
        for (int l = 0; l < s; l++) part[l] = part[l] + part[l + s];
    return part[0];
}

static void normalise(float *v, size_t d) {
    float n = sqrtf(sumsq_fixed(v, d));
    for (size_t j = 0; j < d; j++) v[j] = v[j] / n;
}

void cxs_centre(uint64_t seed, uint64_t c, size_t d, float *out) {
    for (size_t j = 0; j < d; j++) out[j] = gauss(seed, c * (uint64_t)d + j, 0u);
    normalise(out, d);
}

static void row_hash(uint64_t seed_rows, uint64_t r, uint32_t x[4]) {
    philox4x32_10((uint32_t)r, (uint32_t)(r >> 32), 1u, 0u,
                  (uint32_t)seed_rows, (uint32_t)(seed_rows >> 32), x);
}

static float sigma_of(uint32_t x1) {
    if (x1 < 858993459u) return 0.25f;
    if (x1 < 2576980378u) return 0.42f;
    return 0.60f;
}

/* base row r: normalise(centre[cluster] + sigma/sqrt(d) * g) */
static void base_row(uint64_t seed_centres, uint64_t seed_rows, uint64_t n_centres,
                     uint64_t r, size_t d, float *out, float *scratch) {
    uint32_t x[4];
    row_hash(seed_rows, r, x);
    uint64_t c = x[0] % n_centres;
    float amp = sigma_of(x[1]) * (1.0f / sqrtf((float)d));
    cxs_centre(seed_centres, c, d, scratch);
    for (size_t j = 0; j < d; j++) {
        float g = gauss(seed_rows, r * (uint64_t)d + j, 2u);
        float t = amp * g;
        out[j] = scratch[j] + t;
    }
    normalise(out, d);
}

/* Full generator for row r of a corpus of n rows.
 * flags bit0: with duplicates / near-duplicates; bit1: per-row scale. */
void cxs_row(uint64_t seed_centres, uint64_t seed_rows, uint64_t seed_dup, uint64_t n_centres,
             uint64_t r, size_t d, uint32_t flags, float *out) {
    float *scratch = (float *)malloc(d * sizeof(float));
    uint32_t x[4];
    row_hash(seed_rows, r, x);
    uint32_t m = (uint32_t)(r % 1000u);
    if ((flags & 1u) && m >= 998u && r >= 998u) {
        uint64_t p = x[2] % r;
        if (p % 1000u >= 998u) p -= 2;
        base_row(seed_centres, seed_rows, n_centres, p, d, out, scratch);
        if (m == 998u) {
            float amp = 0.045f * (1.0f / sqrtf((float)d));
            for (size_t j = 0; j < d; j++) {
                float g = gauss(seed_dup, r * (uint64_t)d + j, 3u);
                float t = amp * g;
                out[j] = out[j] + t;
            }
            normalise(out, d);
        }
    } else {
        base_row(seed_centres, seed_rows, n_centres, r, d, out, scratch);
    }
    if (flags & 2u) {
        float s = 0.5f + 1.5f * ((float)(x[3] >> 8) * 5.9604645e-08f);
        for (size_t j = 0; j < d; j++) out[j] = out[j] * s;
    }
    free(scratch);
}

void cxs_fill(uint64_t seed_centres, uint64_t seed_rows, uint64_t seed_dup, uint64_t n_centres,
              uint64_t row_lo, uint64_t n_rows, size_t d, uint32_t flags, float *out) {
    long i;
#pragma omp parallel for schedule(static)
    for (i = 0; i < (long)n_rows; i++)
        cxs_row(seed_centres, seed_rows, seed_dup, n_centres, row_lo + (uint64_t)i, d, flags,
                out + (size_t)i * d);
}
