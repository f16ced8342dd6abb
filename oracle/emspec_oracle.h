/*
 * emspec_oracle.h — CPU oracle for the reassigned-spectrogram hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (em-spec_amd/, libemspec)
 * may include, link or call this.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker.
 *
 * PARITY UNPINNED: the reference implementation of this path is not in
 * /root/reference (source is private, README.md:73) and the reference holds
 * no tests, fixtures or golden vectors.  This oracle
 * therefore restates the published three-window reassignment method
 * (Auger & Flandrin 1995; the stage list of SURVEY.md §8a) and is pinned by
 * analytic known-answer tests and an independent numpy formulation
 * (oracle/ref_numpy.py), not by reference outputs.
 *
 * Three restatements live here (the third, eo_frames_exact / eo_batch_exact in emspec_exact.c, is the binary64
 * bit model of EMSPEC_MODE_EXACT):
 *   eo_frames_f64  float64, the textbook method: three explicitly windowed
 *                  DFTs (h, (n-c)h, dh/dn) -> P, t-hat, f-hat.  "Truth".
 *   eo_frames_f32  float32 BIT MODEL of the arithmetic the HIP kernels are
 *                  specified to perform: one packed complex
 *                  radix-2 DIF FFT of x + j(n-c)x, conjugate split, spectral
 *                  Hann identities, fixed operation order, explicit fmaf.
 *                  Integer (col,row) parity is checked against this one.
 */
#ifndef EMSPEC_ORACLE_H
#define EMSPEC_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct eo_cfg {
    int32_t n;        /* FFT size N (power of two) */
    int32_t hop;      /* H */
    int32_t rows;     /* R */
    int32_t reassign; /* 0/1 */
    float sample_rate, fmin_hz, fmax_hz;
    float gain, db_top, db_range, gate_db, power_floor;
} eo_cfg;

/* twiddle: n floats (n/2 complex, re/im interleaved); ebin: rows+1 floats. */
int eo_tables(const eo_cfg* c, float* twiddle, float* ebin);
int eo_default_lut(uint8_t* rgba256x4);
/* process-wide: use this Hz edge table (rows+1 entries) instead of the log axis; NULL resets */
int eo_set_custom_edges_hz(const float* hz, int count);

/* One stream, frames [frame0, frame0+nframes): per-bin outputs [nframes][n/2+1]. */
int eo_frames_f32(const eo_cfg* c, const float* pcm, int64_t L, int64_t frame0,
                  int64_t nframes, float* power, int32_t* col, int32_t* row);
int eo_frames_f64(const eo_cfg* c, const float* pcm, int64_t L, int64_t frame0,
                  int64_t nframes, double* power, double* that, double* khat,
                  int32_t* col, int32_t* row);

/* Whole pipeline (bit model): pcm[S][L] -> db/rgba/index [S][C][R]; any
 * output may be NULL.  threads: OpenMP threads over streams (<=0: all). */
int eo_batch_f32(const eo_cfg* c, const float* pcm, int32_t S, int64_t L,
                 const uint8_t* lut, float* db, uint8_t* rgba, uint8_t* index,
                 int32_t threads);
/* Energy histogram only (before dB): hist[S][C][R] float. */
int eo_hist_f32(const eo_cfg* c, const float* pcm, int32_t S, int64_t L,
                float* hist, int32_t threads);
int eo_max_threads(void);

/* float64 row edges in DFT-bin units (rows+1 doubles): the table eo_frames_f64 and the exact mode compare against */
int eo_edges64(const eo_cfg* c, double* edges);
/* ratio^x by the specified evaluation the row edges are built with (no libm) */
double eo_spec_pow(double ratio, double x);

/* ---- EXACT mode (emspec_exact.c): the binary64 bit model of EMSPEC_MODE_EXACT ----
 * eo_frames_exact: per-bin power (double), absolute column, row and the bin's fixed-point energy q
 *                  (0 when the bin is dropped), each [nframes][n/2+1]; any output may be NULL.
 * eo_batch_exact:  whole pipeline with the int64 histogram: db (float32 of the binary64 dB) / rgba / index
 *                  [S][C][R], hist (optional) the int64 cell sums.  Order-independent by construction.
 * eo_exact_db:     the specified 10*log10 evaluation (no libm), exposed for the GPU probe test. */
int eo_frames_exact(const eo_cfg* c, const float* pcm, int64_t L, int64_t frame0, int64_t nframes,
                    double* power, int32_t* col, int32_t* row, int64_t* q);
int eo_batch_exact(const eo_cfg* c, const float* pcm, int32_t S, int64_t L, const uint8_t* lut, float* db,
                   uint8_t* rgba, uint8_t* index, int64_t* hist, int32_t threads);
double eo_exact_db(double x);   /* = (double)eo_exact_db32((float)x) */
float eo_exact_db32(float x);
float eo_exact_sum32(int64_t sum);   /* the int64 cell sum -> binary32 (two u32 conversions + one fmaf) */

/* emspec_cpu_fast.c: the same pipeline written for speed on a CPU (Stockham radix-4 FFT, ring histogram, vectorised
 * dB), for bench.py's cpu_baseline leg.  Not bit-identical to the bit model; checked against it at the test tolerance. */
int eo_fast_batch(const eo_cfg* c, const float* pcm, int32_t S, int64_t L, float* db, uint8_t* index, int32_t threads);

#ifdef __cplusplus
}
#endif
#endif
