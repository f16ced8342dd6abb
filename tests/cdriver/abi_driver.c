/*
 * abi_driver.c — exercises the C ABI of libemspec from plain C (no Python, no node):
 * the boundary must be usable by any FFI.  Built by tests/test_gpu_parity.py with
 *   gcc -std=c11 abi_driver.c -I include -L em-spec_amd -lemspec -lm
 * Exit code 0 = every check passed.  Needs a gfx950 device.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "emspec.h"

#define CHECK(cond, msg) do { if (!(cond)) { fprintf(stderr, "FAIL %s:%d: %s\n", __FILE__, __LINE__, msg); return 1; } } while (0)

int main(void) {
    emspec_config cfg;
    CHECK(emspec_default_config(&cfg) == EMSPEC_OK, "default_config");
    emspec_engine* e = NULL;
    int rc = emspec_create(&cfg, &e);
    if (rc != EMSPEC_OK) { fprintf(stderr, "create: %s\n", emspec_last_error(NULL)); return 2; }
    CHECK(strncmp(emspec_device_arch(e), "gfx950", 6) == 0, "arch");

    const int n = 4096, hop = 256, frames = 40, R = cfg.rows;
    const long L = n + (long)hop * (frames - 1);
    float* pcm = (float*)malloc(sizeof(float) * L);
    for (long i = 0; i < L; ++i) pcm[i] = 0.5f * sinf(2.0f * 3.14159265f * 1000.0f * (float)i / 48000.0f) + (i == 5000 ? 0.8f : 0.0f);
    CHECK(emspec_num_columns(L, n, hop) == frames, "num_columns");
    const int D = emspec_latency_columns(n, hop, 1);
    CHECK(D == 8, "latency");

    /* batched, host buffers */
    float* db = (float*)malloc(sizeof(float) * frames * R);
    unsigned char* rgba = (unsigned char*)malloc((size_t)4 * frames * R);
    emspec_out out = {db, rgba, NULL};
    CHECK(emspec_batch(e, pcm, 1, L, n, hop, 1, &out) == EMSPEC_OK, emspec_last_error(e));
    /* the 1 kHz sine must be the brightest row of a middle column and sit near 0 dB - 6 dB (amplitude 0.5) */
    int best = 0;
    for (int r = 1; r < R; ++r) if (db[20 * R + r] > db[20 * R + best]) best = r;
    const double fr = 20.0 * pow(24000.0 / 20.0, (best + 0.5) / R);
    CHECK(fabs(fr - 1000.0) < 10.0, "peak row is not at 1 kHz");
    CHECK(fabs(db[20 * R + best] - (-6.02)) < 0.2, "peak level is not -6 dB");
    CHECK(rgba[4 * (20 * R + best) + 3] == 255, "alpha");

    /* streaming: the renderer call, frame by frame, must reproduce the batched columns */
    float* col = (float*)malloc(sizeof(float) * R);
    double worst = 0;
    int64_t ci = -2;
    for (int j = 0; j < frames; ++j) {
        CHECK(emspec_column(e, pcm + (long)j * hop, n, hop, 1, col, NULL, R, &ci) == EMSPEC_OK, emspec_last_error(e));
        if (j < D) { CHECK(ci == -1, "priming index"); continue; }
        CHECK(ci == j - D, "column index");
        for (int r = 0; r < R; ++r) { double d = fabs(col[r] - db[(j - D) * R + r]); if (d > worst) worst = d; }
    }
    for (int k = 0; k < D; ++k) {
        CHECK(emspec_column_flush(e, col, NULL, R, &ci) == EMSPEC_OK, "flush");
        for (int r = 0; r < R; ++r) { double d = fabs(col[r] - db[ci * R + r]); if (d > worst) worst = d; }
    }
    CHECK(worst < 8.7e-4, "streaming != batch");
    CHECK(emspec_column_flush(e, col, NULL, R, &ci) == EMSPEC_ERR_STATE, "flush past the end");

    /* error paths keep the process alive and explain themselves */
    CHECK(emspec_batch(e, pcm, 1, L, 3000, hop, 1, &out) == EMSPEC_ERR_INVALID_ARG, "bad fft size accepted");
    CHECK(strlen(emspec_last_error(e)) > 0, "empty error message");
    CHECK(emspec_column(e, pcm, 1024, hop, 1, col, NULL, R, &ci) == EMSPEC_ERR_STATE, "shape change mid-stream accepted");
    CHECK(emspec_reset(e) == EMSPEC_OK, "reset");
    CHECK(emspec_column(e, pcm, 1024, hop, 0, col, NULL, R, &ci) == EMSPEC_OK && ci == 0, "reassign-off column has no latency");
    emspec_destroy(e);

    /* EXACT mode through the same ABI (cfg.mode): the streaming call must give the batch call's BITS (64-bit fixed-point
     * histogram: order-independent), two batch runs the same bytes, and the per-bin dump the fixed-point energies */
    cfg.mode = EMSPEC_MODE_EXACT;
    emspec_engine* x = NULL;
    CHECK(emspec_create(&cfg, &x) == EMSPEC_OK, emspec_last_error(NULL));
    float* xdb = (float*)malloc(sizeof(float) * frames * R);
    float* xdb2 = (float*)malloc(sizeof(float) * frames * R);
    emspec_out xo = {xdb, NULL, NULL}, xo2 = {xdb2, NULL, NULL};
    CHECK(emspec_batch(x, pcm, 1, L, n, hop, 1, &xo) == EMSPEC_OK, emspec_last_error(x));
    CHECK(emspec_batch(x, pcm, 1, L, n, hop, 1, &xo2) == EMSPEC_OK, emspec_last_error(x));
    CHECK(memcmp(xdb, xdb2, sizeof(float) * frames * R) == 0, "exact mode: two runs differ");
    double xworst = 0;
    for (int i = 0; i < frames * R; ++i) if (db[i] > -60.0f) { double d = fabs(xdb[i] - db[i]); if (d > xworst) xworst = d; }
    CHECK(xworst < 0.05, "exact and fast mode disagree on a strong cell");
    for (int j = 0; j < frames; ++j) {
        CHECK(emspec_column(x, pcm + (long)j * hop, n, hop, 1, col, NULL, R, &ci) == EMSPEC_OK, emspec_last_error(x));
        if (j >= D) CHECK(memcmp(col, xdb + (size_t)(j - D) * R, sizeof(float) * R) == 0, "exact mode: streaming column differs from the batch bits");
    }
    for (int k = 0; k < D; ++k) {
        CHECK(emspec_column_flush(x, col, NULL, R, &ci) == EMSPEC_OK, "flush");
        CHECK(memcmp(col, xdb + (size_t)ci * R, sizeof(float) * R) == 0, "exact mode: flushed column differs from the batch bits");
    }
    {
        const int K = n / 2 + 1;
        double* pw = (double*)malloc(sizeof(double) * 2 * K);
        int32_t* cc = (int32_t*)malloc(sizeof(int32_t) * 2 * K);
        int32_t* rr = (int32_t*)malloc(sizeof(int32_t) * 2 * K);
        int64_t* qq = (int64_t*)malloc(sizeof(int64_t) * 2 * K);
        CHECK(emspec_parity_dump_exact(x, pcm, 1, L, n, hop, 1, 20, 2, pw, cc, rr, qq) == EMSPEC_OK, emspec_last_error(x));
        int peak = 0;
        for (int k = 1; k < K; ++k) if (pw[k] > pw[peak]) peak = k;
        CHECK(abs(peak - 85) <= 1, "exact dump: the 1 kHz line is not at bin 85");        /* 1000 Hz * 4096 / 48000 = 85.3 */
        CHECK(rr[peak] >= 0 && qq[peak] > 0, "exact dump: the peak bin was dropped");
        CHECK(fabs((double)qq[peak] - pw[peak] * 4294967296.0) <= 0.5 + 1e-6 * (double)qq[peak], "exact dump: q is not round(power * 2^52 / (N/4)^2) = power * 2^32");
        float fpw[8]; int32_t fc[8], frw[8];
        CHECK(emspec_parity_dump(x, pcm, 1, L, n, hop, 1, 0, 1, fpw, fc, frw) == EMSPEC_ERR_STATE, "float32 dump accepted by an exact-mode engine");
        free(pw); free(cc); free(rr); free(qq);
    }
    emspec_destroy(x);
    free(xdb); free(xdb2);
    free(pcm); free(db); free(rgba); free(col);
    printf("abi_driver ok: streaming vs batch max |dB| diff %.2e; exact mode: streaming == batch bits, exact vs fast max %.2e dB on strong cells\n", worst, xworst);
    return 0;
}
