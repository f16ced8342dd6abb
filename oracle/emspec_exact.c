/*
 * emspec_exact.c — CPU oracle of the EXACT mode (EMSPEC_MODE_EXACT).
 *
 * TEST INFRASTRUCTURE ONLY (see emspec_oracle.h); PARITY UNPINNED — the reference holds no source, tests or
 * fixtures (/root/reference/README.md:73).  BASELINE.json:north_star asks for results that match "the reference
 * JS/WebAudio CPU path ... exactly on reassigned integer (time,freq) bin indices"; JavaScript arithmetic is IEEE
 * binary64, so the exact mode computes the whole path in binary64 and accumulates the histogram in 64-bit fixed point
 * (integer adds commute: the finished bytes do not depend on the order in which bins arrive).
 *
 * This file is the BIT MODEL of that mode: the HIP kernels (em-spec_amd/csrc/exact.hip.inc) perform the same IEEE
 * binary64 operations in the same order, so every output below is compared with array_equal, not a tolerance.
 * Stage names are SURVEY.md §8(a)'s.  Independent check: eo_frames_f64 (emspec_oracle.c: three explicitly windowed
 * FFTs, shares no code with this file) — tests/test_oracle.py requires the (col,row) of the two to agree on every bin.
 *
 * Build: gcc -O2 -ffp-contract=off -mfma (oracle/Makefile): the only fused operations are the explicit fma() calls.
 */
#include "emspec_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define EX_PI 3.14159265358979323846

static int ex_ilog2(int n) { int l = 0; while ((1 << l) < n) ++l; return l; }

/* cos and sin of a in [0, pi/4] by their Taylor series in Horner form, plain binary64 operations in this order (no
 * libm call: glibc's sincos(), which gcc substitutes for a cos()/sin() pair, and its separate cos()/sin() differ in the
 * last bit for some arguments, so a table built from libm depends on the compiler).  Truncation < 3e-18; result
 * within about one ulp. */
static void ex_cos_sin(double a, double* c, double* s) {
    const double z = a * a;
    double ps = -1.0 / 121645100408832000.0;       /* -1/19! */
    ps = ps * z + 1.0 / 355687428096000.0;         /* +1/17! */
    ps = ps * z - 1.0 / 1307674368000.0;           /* -1/15! */
    ps = ps * z + 1.0 / 6227020800.0;              /* +1/13! */
    ps = ps * z - 1.0 / 39916800.0;                /* -1/11! */
    ps = ps * z + 1.0 / 362880.0;                  /* +1/9! */
    ps = ps * z - 1.0 / 5040.0;                    /* -1/7! */
    ps = ps * z + 1.0 / 120.0;                     /* +1/5! */
    ps = ps * z - 1.0 / 6.0;                       /* -1/3! */
    *s = a + a * (ps * z);
    double pc = 1.0 / 6402373705728000.0;          /* +1/18! */
    pc = pc * z - 1.0 / 20922789888000.0;          /* -1/16! */
    pc = pc * z + 1.0 / 87178291200.0;             /* +1/14! */
    pc = pc * z - 1.0 / 479001600.0;               /* -1/12! */
    pc = pc * z + 1.0 / 3628800.0;                 /* +1/10! */
    pc = pc * z - 1.0 / 40320.0;                   /* -1/8! */
    pc = pc * z + 1.0 / 720.0;                     /* +1/6! */
    pc = pc * z - 1.0 / 24.0;                      /* -1/4! */
    pc = pc * z + 0.5;                             /* +1/2! */
    *c = 1.0 - pc * z;
}

/* stage "Windows" is implicit (spectral Hann identities); tables: binary64 twiddles, binary64 row edges.
 * tw[q] = (cos, -sin)(2 pi q / N): the first octant from ex_cos_sin, the second by cos(pi/2 - x) = sin x, the second
 * quarter by the quarter turn, tw[N/4] = (0,-1). */
static void ex_twiddle(int n, double* tw) {
    for (int q = 0; q <= n / 8; ++q) {
        double c, s;
        ex_cos_sin(2.0 * EX_PI * (double)q / (double)n, &c, &s);
        tw[2 * q] = c;
        tw[2 * q + 1] = -s;
        if (q > 0) { /* mirror about pi/4: angle index n/4 - q */
            tw[2 * (n / 4 - q)] = s;
            tw[2 * (n / 4 - q) + 1] = -c;
        }
    }
    for (int q = 0; q < n / 4; ++q) { /* second quarter by the quarter-turn symmetry, as in the float32 table */
        tw[2 * (q + n / 4)] = tw[2 * q + 1];
        tw[2 * (q + n / 4) + 1] = -tw[2 * q];
    }
    tw[2 * (n / 4)] = 0.0;
    tw[2 * (n / 4) + 1] = -1.0;
}

/* stage "STFT": the canonical radix-2 DIF network of the float32 bit model (emspec_oracle.c:fft_dif_f32), in binary64 */
static void ex_fft_dif(int n, int log2n, double* re, double* im, const double* tw) {
    for (int s = 0; s < log2n; ++s) {
        int m = n >> (s + 1);
        for (int blk = 0; blk < n; blk += 2 * m) {
            for (int j = 0; j < m; ++j) {
                int p = blk + j, q = j << s;
                double ar = re[p], ai = im[p], br = re[p + m], bi = im[p + m];
                re[p] = ar + br;
                im[p] = ai + bi;
                double dr = ar - br, di = ai - bi;
                if (q == 0) {
                    re[p + m] = dr; im[p + m] = di;
                } else if (q == n / 4) {
                    re[p + m] = di; im[p + m] = -dr;
                } else {
                    double wr = tw[2 * q], wi = tw[2 * q + 1];
                    double t = di * wi;
                    double u = di * wr;
                    re[p + m] = fma(dr, wr, -t);
                    im[p + m] = fma(dr, wi, u);
                }
            }
        }
    }
}
static unsigned ex_bitrev(unsigned v, int bits) {
    unsigned r = 0;
    for (int i = 0; i < bits; ++i) { r = (r << 1) | (v & 1); v >>= 1; }
    return r;
}

/* row = largest r with e[r] <= kh, or -1 unless e[0] <= kh < e[R] */
static int ex_row(const double* e, int R, double kh) {
    if (!(kh >= e[0]) || !(kh < e[R])) return -1;
    int lo = 0, hi = R;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (e[mid] <= kh) lo = mid; else hi = mid;
    }
    return lo;
}

typedef struct {
    int n, log2n, K, R, D;
    double* tw;
    double* e;
    double tscale, pfloor, pmax, qscale;
} explan;

/* fixed point of the histogram: one unit = 2^-52 of a full-scale sine's |X_h|^2 = (N/4)^2 (a power of two, so the
 * scaling is exact); a bin is accumulated iff pfloor <= P <= pmax = 2^61 / qscale (2^9 full-scale sines) */
static double ex_qscale(int log2n) { return ldexp(1.0, 52 - (2 * log2n - 4)); }

static int explan_init(explan* p, const eo_cfg* c) {
    p->n = c->n; p->log2n = ex_ilog2(c->n); p->K = c->n / 2 + 1; p->R = c->rows;
    p->D = c->reassign ? (c->n + 2 * c->hop - 1) / (2 * c->hop) : 0;
    p->tw = (double*)malloc(sizeof(double) * c->n);
    p->e = (double*)malloc(sizeof(double) * (c->rows + 1));
    if (!p->tw || !p->e) return -1;
    ex_twiddle(c->n, p->tw);
    if (eo_edges64(c, p->e)) return -1;
    p->tscale = (double)c->n / 2.0 / (double)c->hop;
    double fs_peak = (double)c->n / 4.0;
    p->pfloor = (double)c->power_floor * fs_peak * fs_peak;
    p->qscale = ex_qscale(p->log2n);
    p->pmax = ldexp(1.0, 61) / p->qscale;
    return 0;
}
static void explan_free(explan* p) { free(p->tw); free(p->e); }

/* stages "Frame gather" .. "Index quantise" for one frame; q = the bin's fixed-point energy (0 when dropped) */
static void ex_frame(const explan* p, const eo_cfg* c, const float* x, int64_t j, double* scratch /* 8n */,
                     double* power, int32_t* col, int32_t* row, int64_t* q) {
    const int n = p->n, K = p->K, half = n / 2;
    double *zr = scratch, *zi = scratch + n, *sr = scratch + 2 * n, *si = scratch + 3 * n;
    const double rs = 2.0 / (double)n;
    for (int i = 0; i < n; ++i) {
        zr[i] = (double)x[i];
        zi[i] = (double)x[i] * ((double)(i - half) * rs); /* exact: 24 x 15 significant bits */
    }
    ex_fft_dif(n, p->log2n, zr, zi, p->tw);
    for (int k = 0; k < n; ++k) {
        unsigned b = ex_bitrev((unsigned)k, p->log2n);
        sr[k] = zr[b]; si[k] = zi[b];
    }
    double *Yr = scratch + 4 * n, *Yi = scratch + 5 * n, *Tr = scratch + 6 * n, *Ti = scratch + 7 * n;
    for (int kk = -1; kk <= half + 1; ++kk) {
        int a = (kk + n) % n, b = (n - a) % n;
        double ar = sr[a], ai = si[a], br = sr[b], bi = si[b];
        Yr[kk + 1] = ar + br; Yi[kk + 1] = ai - bi;
        Tr[kk + 1] = ai + bi; Ti[kk + 1] = br - ar;
    }
    for (int k = 0; k < K; ++k) {
        double y0r = Yr[k + 1], y0i = Yi[k + 1], ymr = Yr[k], ymi = Yi[k], ypr = Yr[k + 2], ypi = Yi[k + 2];
        double t0r = Tr[k + 1], t0i = Ti[k + 1], tmr = Tr[k], tmi = Ti[k], tpr = Tr[k + 2], tpi = Ti[k + 2];
        double Ar = (y0r + y0r) - (ymr + ypr), Ai = (y0i + y0i) - (ymi + ypi);
        double Br = (t0r + t0r) - (tmr + tpr), Bi = (t0i + t0i) - (tmi + tpi);
        double Dr = ymr - ypr, Di = ymi - ypi;
        double den = fma(Ar, Ar, Ai * Ai);
        double P = den * 0.015625;
        int32_t cj = (int32_t)j, rw = -1;
        if (P >= p->pfloor && P <= p->pmax) {
            if (c->reassign) {
                double numT = fma(Br, Ar, Bi * Ai);
                double numF = fma(Dr, Ar, Di * Ai);
                double inv = 1.0 / den;
                double ts = numT * inv;
                double ks = numF * inv;
                double cf = floor(fma(ts, p->tscale, 0.5));
                if (fabs(cf) <= (double)p->D) {
                    cj = (int32_t)j + (int32_t)cf;
                    rw = ex_row(p->e, p->R, (double)k + ks);
                }
            } else {
                rw = ex_row(p->e, p->R, (double)k);
            }
        }
        if (power) power[k] = P;
        if (col) col[k] = cj;
        if (row) row[k] = rw;
        if (q) q[k] = rw >= 0 ? (int64_t)rint(P * p->qscale) : 0; /* round to nearest even; < 2^61 by the gate */
    }
}

int eo_frames_exact(const eo_cfg* c, const float* pcm, int64_t L, int64_t frame0, int64_t nframes,
                    double* power, int32_t* col, int32_t* row, int64_t* q) {
    if (!c || !pcm || c->n < 64 || (c->n & (c->n - 1))) return -1;
    if (frame0 < 0 || (frame0 + nframes - 1) * c->hop + c->n > L) return -2;
    explan p; if (explan_init(&p, c)) return -3;
    double* buf = (double*)malloc(sizeof(double) * 8 * c->n);
    for (int64_t f = 0; f < nframes; ++f) {
        int64_t j = frame0 + f;
        ex_frame(&p, c, pcm + j * c->hop, j, buf, power ? power + f * p.K : NULL, col ? col + f * p.K : NULL,
                 row ? row + f * p.K : NULL, q ? q + f * p.K : NULL);
    }
    free(buf); explan_free(&p);
    return 0;
}

/* stage "dB + colour", exact mode: 10 log10 through a SPECIFIED binary32 evaluation (no libm call, no division; IEEE operations
 * in this order, fmaf where written, so the GPU and the CPU produce the same bits): x = m 2^e with m in (sqrt(1/2), sqrt 2],
 * f = m - 1 (exact), log2 m = f P(f) with P the degree-8 interpolant of log2(1+f)/f at the Chebyshev nodes of
 * [sqrt(1/2)-1, sqrt(2)-1] (truncation 1.5e-8; with binary32 rounding the result is within 6.1e-8 of log2 m, tests/test_oracle.py).
 * Binary32, not binary64: the outputs are a float32 dB and a palette byte, and on MI355X a binary64 vector instruction costs
 * four binary32 ones; the indices, the energies and their sums stay binary64 / int64.  No division (until round 5's last form:
 * the atanh series in s = (m-1)/(m+1)): a correctly rounded binary32 division is ten vector instructions on that hardware,
 * and this stage's instruction count is part of what bounds the fused kernel.
 * The cell's energy enters through eo_exact_sum32: the two 32-bit halves of the non-negative int64 sum, each converted
 * exactly-or-rounded-once (u32 -> binary32), joined by one fmaf. */
float eo_exact_sum32(int64_t sum) {
    const uint64_t u = (uint64_t)sum;
    return fmaf((float)(uint32_t)(u >> 32), 4294967296.0f, (float)(uint32_t)(u & 0xffffffffu));
}
float eo_exact_db32(float x) {
    union { float f; uint32_t u; } v;
    v.f = x;
    int e = (int)((v.u >> 23) & 0xff) - 127;
    v.u = (v.u & 0x007fffffu) | 0x3f800000u;
    float m = v.f;
    if (m > 1.41421354f) { m = m * 0.5f; e += 1; }
    const float f = m - 1.0f;
    float p = 0.123109683f;
    p = fmaf(p, f, -0.205861881f);
    p = fmaf(p, f, 0.216078222f);
    p = fmaf(p, f, -0.239169881f);
    p = fmaf(p, f, 0.287903249f);
    p = fmaf(p, f, -0.360693276f);
    p = fmaf(p, f, 0.480910748f);
    p = fmaf(p, f, -0.721347451f);
    p = fmaf(p, f, 1.44269502f);
    const float l2 = f * p;
    return ((float)e + l2) * 3.01029992f; /* 10 log10(2) */
}
double eo_exact_db(double x) { return (double)eo_exact_db32((float)x); }

/* stage "Scatter" in fixed point + "dB + colour".  hist (optional): the int64 sums [S][C][R]. */
int eo_batch_exact(const eo_cfg* c, const float* pcm, int32_t S, int64_t L, const uint8_t* lut, float* db,
                   uint8_t* rgba, uint8_t* index, int64_t* hist_out, int32_t threads) {
    if (!c || !pcm || c->n < 64 || (c->n & (c->n - 1))) return -1;
    explan p; if (explan_init(&p, c)) return -3;
    uint8_t deflut[1024];
    if (!lut) { eo_default_lut(deflut); lut = deflut; }
    const int64_t C = (L >= c->n) ? (L - c->n) / c->hop + 1 : 0;
    const int R = c->rows, K = p.K;
    const double nn = (double)c->n;
    const double scale = 32.0 / (3.0 * nn * nn) * (double)c->gain * (double)c->gain;
    /* the stage's constants, each rounded once to binary32 (emspec_api.cpp: exact_db_map states the same operations) */
    const float sc = (float)(scale * (1.0 / p.qscale));
    const float lo = (float)((double)c->db_top - (double)c->db_range), inv_range = (float)(1.0 / (double)c->db_range);
    const float gate = c->gate_db;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
#endif
    for (int s = 0; s < S; ++s) {
        int64_t* hist = (int64_t*)calloc((size_t)C * R, sizeof(int64_t));
        double* buf = (double*)malloc(sizeof(double) * 8 * c->n);
        int32_t* cl = (int32_t*)malloc(sizeof(int32_t) * K);
        int32_t* rw = (int32_t*)malloc(sizeof(int32_t) * K);
        int64_t* qq = (int64_t*)malloc(sizeof(int64_t) * K);
        const float* x = pcm + (size_t)s * L;
        for (int64_t j = 0; j < C; ++j) {
            ex_frame(&p, c, x + j * c->hop, j, buf, NULL, cl, rw, qq);
            for (int k = 0; k < K; ++k)
                if (rw[k] >= 0 && cl[k] >= 0 && cl[k] < C) hist[(size_t)cl[k] * R + rw[k]] += qq[k];
        }
        size_t base = (size_t)s * C * R;
        for (size_t i = 0; i < (size_t)C * R; ++i) {
            float d = eo_exact_db32(fmaf(eo_exact_sum32(hist[i]), sc, 1e-20f));
            float v = (d - lo) * inv_range;
            v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
            if (d < gate) v = 0.0f;
            int ix = (int)(v * 255.0f + 0.5f);
            if (db) db[base + i] = d;
            if (index) index[base + i] = (uint8_t)ix;
            if (rgba) memcpy(rgba + 4 * (base + i), lut + 4 * ix, 4);
            if (hist_out) hist_out[base + i] = hist[i];
        }
        free(hist); free(buf); free(cl); free(rw); free(qq);
    }
    explan_free(&p);
    return 0;
}
