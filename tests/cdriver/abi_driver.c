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
    /* several streams from ordinary malloc'ed memory: the pipelined path (units through three staging sets, the copies out on
     * the library's second host thread, its page-touching threads ahead of it) - stream s is the signal scaled by (1 - s/8):
     * stream 0 must give the single-stream call's bits, every stream its own single-stream call's */
    {
        const int S5 = 5;
        float* p5 = (float*)malloc(sizeof(float) * S5 * L);
        float* d5 = (float*)malloc(sizeof(float) * S5 * frames * R);
        unsigned char* i5 = (unsigned char*)malloc((size_t)S5 * frames * R);
        unsigned char* i1 = (unsigned char*)malloc((size_t)frames * R);
        for (int s5 = 0; s5 < S5; ++s5) for (long i = 0; i < L; ++i) p5[(size_t)s5 * L + i] = pcm[i] * (1.0f - (float)s5 / 8.0f);
        emspec_out o5 = {d5, NULL, i5};
        for (int rep = 0; rep < 2; ++rep) {
            memset(i5, 0xAB, (size_t)S5 * frames * R);
            CHECK(emspec_batch(x, p5, S5, L, n, hop, 1, &o5) == EMSPEC_OK, emspec_last_error(x));
            CHECK(memcmp(d5, xdb, sizeof(float) * frames * R) == 0, "multi-stream pageable batch: stream 0 differs from the single-stream bits");
            for (int s5 = 1; s5 < S5; ++s5) {
                emspec_out o1 = {xdb2, NULL, i1};
                CHECK(emspec_batch(x, p5 + (size_t)s5 * L, 1, L, n, hop, 1, &o1) == EMSPEC_OK, emspec_last_error(x));
                CHECK(memcmp(d5 + (size_t)s5 * frames * R, xdb2, sizeof(float) * frames * R) == 0, "multi-stream pageable batch: a stream's dB differs from its own call");
                CHECK(memcmp(i5 + (size_t)s5 * frames * R, i1, (size_t)frames * R) == 0, "multi-stream pageable batch: a stream's palette index differs from its own call");
            }
        }
        free(p5); free(d5); free(i5); free(i1);
    }
    /* the live multi-stream session from plain C: three streams (the signal, half of it, silence) fed hop by hop through
     * emspec_push_samples_multi from PAGE-LOCKED blocks (emspec_host_alloc: the kernel reads and writes them in place), one
     * restarted half way; stream 0 must give the batch call's bits, the silent stream the floor everywhere */
    {
        const int S = 3;
        float* blk = NULL; float* odb = NULL;
        CHECK(emspec_host_alloc(sizeof(float) * S * hop, (void**)&blk) == EMSPEC_OK, "host_alloc");
        CHECK(emspec_host_alloc(sizeof(float) * S * 1 * R, (void**)&odb) == EMSPEC_OK, "host_alloc");
        int64_t counts[3], firsts[3], cols[3];
        CHECK(emspec_live_streams(x) == 0, "a live session before the first call");
        CHECK(emspec_reset_stream(x, 0) == EMSPEC_ERR_STATE, "reset_stream without a session");
        /* prime with n - hop samples: no frame is complete, nothing is launched */
        float* prime = (float*)malloc(sizeof(float) * S * (n - hop));
        for (int s = 0; s < S; ++s) for (int i = 0; i < n - hop; ++i) prime[(size_t)s * (n - hop) + i] = s == 0 ? pcm[i] : (s == 1 ? 0.5f * pcm[i] : 0.0f);
        CHECK(emspec_push_columns_multi(x, n - hop, n, hop, 1) == 0, "push_columns_multi");
        CHECK(emspec_push_samples_multi(x, prime, S, n - hop, n - hop, n, hop, 1, NULL, NULL, R, 0, counts, firsts) == EMSPEC_OK, emspec_last_error(x));
        CHECK(emspec_live_streams(x) == S && counts[0] == 0 && firsts[2] == -1, "priming block");
        int restarted_at = -1;
        for (int j = 0; j < frames; ++j) {
            if (j == 25) { CHECK(emspec_reset_stream(x, 1) == EMSPEC_OK, emspec_last_error(x)); restarted_at = j; }
            for (int i = 0; i < hop; ++i) {
                const float v = pcm[(long)(n - hop) + (long)j * hop + i];
                blk[i] = v; blk[hop + i] = 0.5f * v; blk[2 * hop + i] = 0.0f;
            }
            CHECK(emspec_push_samples_multi(x, blk, S, hop, hop, n, hop, 1, odb, NULL, R, 1, counts, firsts) == EMSPEC_OK, emspec_last_error(x));
            CHECK(counts[0] == (j >= D ? 1 : 0) && counts[2] == counts[0], "live: columns per call");
            if (j >= D) {
                CHECK(firsts[0] == j - D, "live: column index");
                CHECK(memcmp(odb, xdb + (size_t)(j - D) * R, sizeof(float) * R) == 0, "live: stream 0 differs from the batch bits");
                for (int r = 0; r < R; ++r) CHECK(odb[(size_t)2 * R + r] == odb[(size_t)2 * R], "live: the silent stream is not flat");
            }
            if (restarted_at >= 0) CHECK(counts[1] == 0, "live: the restarted stream has no complete frame yet (it needs n samples)");
        }
        CHECK(emspec_columns(x, pcm, S, n, hop, 1, odb, NULL, R, cols) == EMSPEC_ERR_STATE, "feeding form changed mid-session");
        for (int k = 0; k < D; ++k) {
            CHECK(emspec_columns_flush(x, odb, NULL, R, cols) == EMSPEC_OK, emspec_last_error(x));
            CHECK(cols[0] == frames - D + k && cols[1] == -1, "live flush: column indices");
            CHECK(memcmp(odb, xdb + (size_t)cols[0] * R, sizeof(float) * R) == 0, "live flush: stream 0 differs from the batch bits");
        }
        CHECK(emspec_columns_flush(x, odb, NULL, R, cols) == EMSPEC_ERR_STATE, "live flush past the end");
        CHECK(emspec_reset(x) == EMSPEC_OK && emspec_live_streams(x) == 0, "reset ends the live session");
        emspec_host_free(blk); emspec_host_free(odb); free(prime);
        /* the same session from ORDINARY (pageable) memory - the library stages blocks and columns on the host thread - with
         * blocks of five hops (several columns per call, outputs [S][max_columns][rows]) and dB + RGBA out; then the per-frame
         * form (emspec_columns) from pageable frames; both against the batch call's bits */
        {
            const int B = 5, maxc = 6;
            float* pb = (float*)malloc(sizeof(float) * S * B * hop);
            float* pdb = (float*)malloc(sizeof(float) * S * maxc * R);
            unsigned char* prg = (unsigned char*)malloc((size_t)4 * S * maxc * R);
            long fed = 0;
            int next = 0;
            while (fed + (long)B * hop <= L) {
                for (int s2 = 0; s2 < S; ++s2) for (int i = 0; i < B * hop; ++i) pb[(size_t)s2 * B * hop + i] = s2 == 0 ? pcm[fed + i] : (s2 == 1 ? 0.5f * pcm[fed + i] : 0.0f);
                CHECK(emspec_push_columns_multi(x, B * hop, n, hop, 1) <= maxc, "push_columns_multi bound");
                CHECK(emspec_push_samples_multi(x, pb, S, B * hop, B * hop, n, hop, 1, pdb, prg, R, maxc, counts, firsts) == EMSPEC_OK, emspec_last_error(x));
                for (int i = 0; i < counts[0]; ++i) {
                    CHECK(firsts[0] + i == next, "pageable live: column order");
                    CHECK(memcmp(pdb + (size_t)i * R, xdb + (size_t)next * R, sizeof(float) * R) == 0, "pageable live: stream 0 differs from the batch bits");
                    CHECK(prg[4 * ((size_t)i * R) + 3] == 255, "pageable live: alpha");
                    ++next;
                }
                fed += (long)B * hop;
            }
            CHECK(next >= 10, "pageable live: too few columns");
            CHECK(emspec_reset(x) == EMSPEC_OK, "reset");
            float* fr = (float*)malloc(sizeof(float) * S * n);
            for (int j = 0; j < 12; ++j) {
                for (int s2 = 0; s2 < S; ++s2) for (int i = 0; i < n; ++i) fr[(size_t)s2 * n + i] = s2 == 0 ? pcm[(long)j * hop + i] : 0.0f;
                CHECK(emspec_columns(x, fr, S, n, hop, 1, pdb, prg, R, cols) == EMSPEC_OK, emspec_last_error(x));
                if (j >= D) CHECK(cols[0] == j - D && memcmp(pdb, xdb + (size_t)(j - D) * R, sizeof(float) * R) == 0, "pageable emspec_columns differs from the batch bits");
            }
            CHECK(emspec_reset(x) == EMSPEC_OK, "reset");
            free(pb); free(pdb); free(prg); free(fr);
        }
    }
    emspec_destroy(x);
    free(xdb); free(xdb2);
    free(pcm); free(db); free(rgba); free(col);
    printf("abi_driver ok: streaming vs batch max |dB| diff %.2e; exact mode: streaming == batch bits, exact vs fast max %.2e dB on strong cells; 5-stream pageable batch == per-stream bits; live multi-stream session (3 streams, page-locked and pageable blocks, sample- and frame-fed) == batch bits\n", worst, xworst);
    return 0;
}
