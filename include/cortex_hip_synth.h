/*
 * cortex_hip_synth.h — benchmark/test support: deterministic synthetic
 * embeddings generated directly in HBM, so multi-GB corpora
 * are never shipped over PCIe.  Not part of the reference's interface; the
 * reference has no data generator.  Bit-identical to its CPU twin in
 * oracle/cortex_synth.c (tests/test_hip_synth.py).
 */
#ifndef CORTEX_HIP_SYNTH_H
#define CORTEX_HIP_SYNTH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CX_SYNTH_DUPLICATES 1u /* rows r%1000==999 exact, ==998 near duplicates of an earlier row */
#define CX_SYNTH_SCALED 2u     /* per-row scale in [0.5, 2): the un-normalised fixture */

/* Writes rows [row_lo, row_lo + n_rows) of the synthetic corpus whose mixture
 * has n_centres centres, row-major f32 [n_rows][dim], into d_out (HBM on
 * `device`).  Synchronous.  0 = ok, else CX_ERR_* with cx_last_error(). */
int cx_synth_fill_dev(int device, float *d_out, uint64_t seed_centres, uint64_t seed_rows,
                      uint64_t seed_dup, uint64_t n_centres, uint64_t row_lo, uint64_t n_rows,
                      uint32_t dim, uint32_t flags);

/* Measured peaks of the device for the rooflines: sustained HBM read bandwidth of a plain
 * streaming-read kernel (GB/s, best of `reps` passes over `bytes`), and the sustained dense bf16 MFMA rate of a
 * register-only v_mfma_f32_32x32x16_bf16 loop running for about `ms_target` ms (TFLOP/s, clock settled). */
int cx_probe_read_bw(int device, uint64_t bytes, uint32_t reps, double *out_gbs);
int cx_probe_mfma_tflops(int device, double ms_target, double *out_tflops);
/* the same with every step's operands read from LDS at the filter GEMM's ratio (12 ds_read_b128 per 16 MFMAs);
 * with_dma != 0 adds that kernel's global traffic (4 LDS-DMAs of 1 KiB per wave and step, L2-resident source) */
int cx_probe_mfma_lds_tflops(int device, double ms_target, int with_dma, double *out_tflops);

#ifdef __cplusplus
}
#endif
#endif
