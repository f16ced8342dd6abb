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
    free(pcm); free(db); free(rgba); free(col);
    printf("abi_driver ok: streaming vs batch max |dB| diff %.2e\n", worst);
    return 0;
}
