/*
 * emspec_oracle.c — CPU oracle (see emspec_oracle.h: TEST INFRASTRUCTURE ONLY,
 * PARITY UNPINNED — no reference source, tests or fixtures exist to cite;
 * /root/reference/README.md:73).
 *
 * Follows SURVEY.md §8(a) stage by stage; stage names below are that table's.
 *
 * Build: gcc -O2 -ffp-contract=off -mfma -fopenmp -shared -fPIC   (oracle/Makefile)
 *   -ffp-contract=off : the bit model's operation order is the specification;
 *                       the only fused operations are the explicit fmaf() calls.
 */
#include "emspec_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define EO_PI 3.14159265358979323846

static int ilog2(int n) { int l = 0; while ((1 << l) < n) ++l; return l; }
static int is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

static int check_cfg(const eo_cfg* c) {
    if (!c || !is_pow2(c->n) || c->n < 64 || c->n > 65536) return -1;
    if (c->hop < 1 || c->hop > c->n) return -1;
    if (c->rows < 1 || c->rows > 8192) return -1;
    if (!(c->fmin_hz > 0) || !(c->fmax_hz > c->fmin_hz)) return -1;
    return 0;
}

/* ---- tables (stage "Windows" is implicit: Hann enters through the spectral
 *      identities; the only tables are twiddles and row edges) ------------- */
static void make_twiddle(int n, float* tw) {
    for (int q = 0; q < n / 2; ++q) {
        double a = 2.0 * EO_PI * (double)q / (double)n;
        tw[2 * q] = (float)cos(a);
        tw[2 * q + 1] = (float)(-sin(a));
    }
    /* second quarter by symmetry, tw[q + N/4] = -j tw[q] = (tw[q].im, -tw[q].re): the values cos/sin give anyway
     * (an identity with a correctly rounded libm, checked by tests/test_oracle.py), stated so that it is a property
     * of the table */
    for (int q = 0; q < n / 4; ++q) {
        tw[2 * (q + n / 4)] = tw[2 * q + 1];
        tw[2 * (q + n / 4) + 1] = -tw[2 * q];
    }
    /* the quarter-turn entry is exact, so "multiply by tw[N/4]" == "(di,-dr)" */
    tw[2 * (n / 4)] = 0.0f;
    tw[2 * (n / 4) + 1] = -1.0f;
}
/* Optional arbitrary monotone axis (mirrors emspec_set_row_edges_hz): rows+1 edges in Hz. */
static float* g_custom_edges = NULL;
static int g_custom_count = 0;
int eo_set_custom_edges_hz(const float* hz, int count) {
    free(g_custom_edges); g_custom_edges = NULL; g_custom_count = 0;
    if (hz && count > 0) {
        g_custom_edges = (float*)malloc(sizeof(float) * count);
        memcpy(g_custom_edges, hz, sizeof(float) * count);
        g_custom_count = count;
    }
    return 0;
}


/* ratio^x by a SPECIFIED evaluation: exp2(x * log2(ratio)) from plain IEEE binary64 operations in this
 * order - no libm, whose pow() is not correctly rounded and differs between C libraries, so a table built from it would
 * depend on the host.  log2 by the atanh series on the mantissa folded into [1/sqrt2, sqrt2]; 2^f, |f| <= 1/2,
 * by the Taylor series of e^(f ln 2) in Horner form (truncation < 4e-18); scaling by 2^i is exact.  Within ~3 ulp of the
 * real value; what matters is that every build produces the same bits. */
static double spec_log2(double x) {
    union { double d; uint64_t u; } v;
    v.d = x;
    int e = (int)((v.u >> 52) & 0x7ff) - 1023;
    v.u = (v.u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m = v.d;
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    const double s = (m - 1.0) / (m + 1.0);
    const double z = s * s;
    double pz = 1.0 / 21.0;
    pz = pz * z + 1.0 / 19.0;
    pz = pz * z + 1.0 / 17.0;
    pz = pz * z + 1.0 / 15.0;
    pz = pz * z + 1.0 / 13.0;
    pz = pz * z + 1.0 / 11.0;
    pz = pz * z + 1.0 / 9.0;
    pz = pz * z + 1.0 / 7.0;
    pz = pz * z + 1.0 / 5.0;
    pz = pz * z + 1.0 / 3.0;
    pz = pz * z + 1.0;
    return (double)e + (s * pz) * 2.8853900817779268; /* 2 / ln 2 */
}
static double spec_exp2(double x) {
    const double i = (double)(long long)(x < 0.0 ? x - 0.5 : x + 0.5); /* nearest integer (halves away from zero) */
    const double t = (x - i) * 0.6931471805599453;                    /* x - i is exact; |t| <= 0.3466 */
    double p = 1.0 / 87178291200.0;      /* 1/14! */
    p = p * t + 1.0 / 6227020800.0;      /* 1/13! */
    p = p * t + 1.0 / 479001600.0;       /* 1/12! */
    p = p * t + 1.0 / 39916800.0;        /* 1/11! */
    p = p * t + 1.0 / 3628800.0;         /* 1/10! */
    p = p * t + 1.0 / 362880.0;          /* 1/9! */
    p = p * t + 1.0 / 40320.0;           /* 1/8! */
    p = p * t + 1.0 / 5040.0;            /* 1/7! */
    p = p * t + 1.0 / 720.0;             /* 1/6! */
    p = p * t + 1.0 / 120.0;             /* 1/5! */
    p = p * t + 1.0 / 24.0;              /* 1/4! */
    p = p * t + 1.0 / 6.0;               /* 1/3! */
    p = p * t + 0.5;                     /* 1/2! */
    p = p * t + 1.0;
    p = p * t + 1.0;
    union { double d; uint64_t u; } s;
    s.u = (uint64_t)(1023 + (long long)i) << 52;                      /* 2^i, |i| < 1000 */
    return p * s.d;
}
double eo_spec_pow(double ratio, double x) {
    if (x == 1.0) return ratio; /* the axis ends exactly at fmax (as pow(ratio, 1) would) */
    return spec_exp2(x * spec_log2(ratio));
}

static void make_edges64(const eo_cfg* c, double* e) {
    /* row edges expressed in DFT-bin units (Hz * N / fs): custom table, or log-spaced */
    if (g_custom_edges && g_custom_count == c->rows + 1) {
        for (int r = 0; r <= c->rows; ++r) e[r] = (double)g_custom_edges[r] * (double)c->n / (double)c->sample_rate;
        return;
    }
    double ratio = (double)c->fmax_hz / (double)c->fmin_hz;
    for (int r = 0; r <= c->rows; ++r)
        e[r] = (double)c->fmin_hz * eo_spec_pow(ratio, (double)r / (double)c->rows) *
               (double)c->n / (double)c->sample_rate;
}
/* float64 row edges in DFT-bin units (rows+1): what eo_frames_f64 and the exact mode (emspec_exact.c) compare against */
int eo_edges64(const eo_cfg* c, double* e) {
    if (check_cfg(c) || !e) return -1;
    make_edges64(c, e);
    return 0;
}
int eo_tables(const eo_cfg* c, float* twiddle, float* ebin) {
    if (check_cfg(c)) return -1;
    if (twiddle) make_twiddle(c->n, twiddle);
    if (ebin) {
        double* e = (double*)malloc(sizeof(double) * (c->rows + 1));
        make_edges64(c, e);
        for (int r = 0; r <= c->rows; ++r) ebin[r] = (float)e[r];
        free(e);
    }
    return 0;
}

/* 5-stop gradient measured from assets/settings.png, integer
 * interpolation so every implementation produces identical bytes. */
int eo_default_lut(uint8_t* lut) {
    static const int stops[5][3] = {{0, 0, 0}, {80, 0, 80}, {200, 50, 50}, {255, 150, 0}, {255, 255, 200}};
    for (int i = 0; i < 256; ++i) {
        int pos = i * 4, seg = pos / 255;
        if (seg > 3) seg = 3;
        int fr = pos - seg * 255;
        for (int ch = 0; ch < 3; ++ch)
            lut[4 * i + ch] = (uint8_t)((stops[seg][ch] * (255 - fr) + stops[seg + 1][ch] * fr + 127) / 255);
        lut[4 * i + 3] = 255;
    }
    return 0;
}

/* row = largest r with ebin[r] <= kh, valid only for ebin[0] <= kh < ebin[R] */
static int row_lookup_f32(const float* ebin, int R, float kh) {
    if (!(kh >= ebin[0]) || !(kh < ebin[R])) return -1;
    int lo = 0, hi = R; /* invariant: ebin[lo] <= kh < ebin[hi] */
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (ebin[mid] <= kh) lo = mid; else hi = mid;
    }
    return lo;
}
static int row_lookup_f64(const double* e, int R, double kh) {
    if (!(kh >= e[0]) || !(kh < e[R])) return -1;
    int lo = 0, hi = R;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (e[mid] <= kh) lo = mid; else hi = mid;
    }
    return lo;
}

static int latency_cols(const eo_cfg* c) {
    return c->reassign ? (c->n + 2 * c->hop - 1) / (2 * c->hop) : 0;
}

/* ---- float32 bit model ------------------------------------------------- */
/* Stage "STFT": canonical radix-2 decimation-in-frequency FFT, in place,
 * natural-order input, bit-reversed output.  Stage s pairs (p, p+m),
 * m = N >> (s+1), twiddle index q = (p mod m) << s:
 *     x[p]   = a + b
 *     x[p+m] = (a - b) * tw[q]
 * with the complex product defined as
 *     re = fmaf(dr, wr, -(di*wi));  im = fmaf(dr, wi, di*wr)
 * and the trivial twiddles applied exactly (q==0: d;  q==N/4: (di,-dr)).
 * Any kernel that performs these same butterflies (in any grouping into
 * radix-4/8/16 register passes) produces bit-identical output. */
static void fft_dif_f32(int n, int log2n, float* re, float* im, const float* tw) {
    for (int s = 0; s < log2n; ++s) {
        int m = n >> (s + 1);
        for (int blk = 0; blk < n; blk += 2 * m) {
            for (int j = 0; j < m; ++j) {
                int p = blk + j, q = j << s;
                float ar = re[p], ai = im[p], br = re[p + m], bi = im[p + m];
                re[p] = ar + br;
                im[p] = ai + bi;
                float dr = ar - br, di = ai - bi;
                if (q == 0) {
                    re[p + m] = dr; im[p + m] = di;
                } else if (q == n / 4) {
                    re[p + m] = di; im[p + m] = -dr;
                } else {
                    float wr = tw[2 * q], wi = tw[2 * q + 1];
                    float t = di * wi;
                    float u = di * wr;
                    re[p + m] = fmaf(dr, wr, -t);
                    im[p + m] = fmaf(dr, wi, u);
                }
            }
        }
    }
}
static unsigned bitrev(unsigned v, int bits) {
    unsigned r = 0;
    for (int i = 0; i < bits; ++i) { r = (r << 1) | (v & 1); v >>= 1; }
    return r;
}

typedef struct {
    int n, log2n, K, R, D;
    float* tw;
    float* ebin;
    float tscale, pfloor_abs; /* tscale = (N/2)/H: ramp units -> columns */
} plan32;

static int plan32_init(plan32* p, const eo_cfg* c) {
    p->n = c->n; p->log2n = ilog2(c->n); p->K = c->n / 2 + 1; p->R = c->rows;
    p->D = latency_cols(c);
    p->tw = (float*)malloc(sizeof(float) * c->n);
    p->ebin = (float*)malloc(sizeof(float) * (c->rows + 1));
    if (!p->tw || !p->ebin) return -1;
    eo_tables(c, p->tw, p->ebin);
    p->tscale = (float)((double)c->n / 2.0 / (double)c->hop);
    double fs_peak = (double)c->n / 4.0; /* |X_h| of a full-scale sine */
    p->pfloor_abs = (float)((double)c->power_floor * fs_peak * fs_peak);
    return 0;
}
static void plan32_free(plan32* p) { free(p->tw); free(p->ebin); }

/* One frame: stages "Frame gather" .. "Index quantise" of SURVEY.md §8(a).
 */
static void frame_f32(const plan32* p, const eo_cfg* c, const float* x, int64_t j,
                      float* scratch /* 8n floats */,
                      float* power, int32_t* col, int32_t* row) {
    const int n = p->n, K = p->K, half = n / 2;
    float *zr = scratch, *zi = scratch + n, *sr = scratch + 2 * n, *si = scratch + 3 * n;
    /* pack z[n] = x[n] + j * r[n] * x[n],  r[n] = (n - c) / (N/2) in [-1,1): exact in f32,
     * one f32 multiply.  The ramp is normalised so both packed signals have
     * comparable norm — otherwise the FFT's rounding error (relative to |z|)
     * would swamp the x part. */
    const float rs = 2.0f / (float)n;
    for (int i = 0; i < n; ++i) {
        zr[i] = x[i];
        zi[i] = x[i] * ((float)(i - half) * rs);
    }
    fft_dif_f32(n, p->log2n, zr, zi, p->tw);
    /* undo the bit reversal: Z[k] sits at position bitrev(k) */
    for (int k = 0; k < n; ++k) {
        unsigned b = bitrev((unsigned)k, p->log2n);
        sr[k] = zr[b]; si[k] = zi[b];
    }
    /* conjugate split (scaled by 2):  Yh = Z[k] + conj Z[N-k],  Th = -j (Z[k] - conj Z[N-k])
     * evaluated for k = -1 .. N/2+1 ; stored at index k+1 */
    float *Yr = scratch + 4 * n, *Yi = scratch + 5 * n, *Tr = scratch + 6 * n, *Ti = scratch + 7 * n;
    for (int kk = -1; kk <= half + 1; ++kk) {
        int a = (kk + n) % n, b = (n - a) % n;
        float ar = sr[a], ai = si[a], br = sr[b], bi = si[b];
        Yr[kk + 1] = ar + br; Yi[kk + 1] = ai - bi;
        Tr[kk + 1] = ai + bi; Ti[kk + 1] = br - ar;
    }
    for (int k = 0; k < K; ++k) {
        /* spectral Hann identities (all scaled by 8):
         *   A = 8 X_h  = 2Y[k] - (Y[k-1] + Y[k+1])
         *   B = 8 X_th = 2T[k] - (T[k-1] + T[k+1])
         *   Dd: Y[k-1] - Y[k+1]  (X_dh = -j (2pi/N) Dd / 8) */
        float y0r = Yr[k + 1], y0i = Yi[k + 1], ymr = Yr[k], ymi = Yi[k], ypr = Yr[k + 2], ypi = Yi[k + 2];
        float t0r = Tr[k + 1], t0i = Ti[k + 1], tmr = Tr[k], tmi = Ti[k], tpr = Tr[k + 2], tpi = Ti[k + 2];
        float Ar = (y0r + y0r) - (ymr + ypr), Ai = (y0i + y0i) - (ymi + ypi);
        float Br = (t0r + t0r) - (tmr + tpr), Bi = (t0i + t0i) - (tmi + tpi);
        float Dr = ymr - ypr, Di = ymi - ypi;
        /* stage "Power + gate" */
        float den = fmaf(Ar, Ar, Ai * Ai);
        float P = den * 0.015625f; /* |X_h|^2 = |A|^2 / 64, exact scaling */
        power[k] = P;
        int32_t cj = (int32_t)j, rw = -1;
        if (P >= p->pfloor_abs && P <= 1.0e36f) {
            if (c->reassign) {
                /* stage "Reassign": t-shift = Re(B conj A)/|A|^2 [units of N/2 samples],
                 *                   k-shift = Re(Dd conj A)/|A|^2 [bins]   */
                float numT = fmaf(Br, Ar, Bi * Ai);
                float numF = fmaf(Dr, Ar, Di * Ai);
                float inv = 1.0f / den;
                float ts = numT * inv;
                float ks = numF * inv;
                /* stage "Index quantise" */
                float cf = floorf(fmaf(ts, p->tscale, 0.5f)); /* ts is in units of N/2 samples */
                if (fabsf(cf) <= (float)p->D) {
                    cj = (int32_t)j + (int32_t)cf;
                    float kh = (float)k + ks;
                    rw = row_lookup_f32(p->ebin, p->R, kh);
                }
            } else {
                rw = row_lookup_f32(p->ebin, p->R, (float)k);
            }
        }
        col[k] = cj; row[k] = rw;
    }
}

int eo_frames_f32(const eo_cfg* c, const float* pcm, int64_t L, int64_t frame0,
                  int64_t nframes, float* power, int32_t* col, int32_t* row) {
    if (check_cfg(c) || !pcm) return -1;
    if (frame0 < 0 || (frame0 + nframes - 1) * c->hop + c->n > L) return -2;
    plan32 p; if (plan32_init(&p, c)) return -3;
    int n = c->n;
    float* buf = (float*)malloc(sizeof(float) * 8 * n);
    for (int64_t f = 0; f < nframes; ++f) {
        int64_t j = frame0 + f;
        frame_f32(&p, c, pcm + j * c->hop, j, buf,
                  power + f * p.K, col + f * p.K, row + f * p.K);
    }
    free(buf); plan32_free(&p);
    return 0;
}

/* stage "Scatter": hist[col][row] += P, frames in order, bins in order (f32) */
static void stream_hist_f32(const plan32* p, const eo_cfg* c, const float* pcm, int64_t L, float* hist) {
    int n = c->n; int64_t C = (L >= n) ? (L - n) / c->hop + 1 : 0;
    float* buf = (float*)malloc(sizeof(float) * 8 * n);
    float* pw = (float*)malloc(sizeof(float) * p->K);
    int32_t* cl = (int32_t*)malloc(sizeof(int32_t) * p->K);
    int32_t* rw = (int32_t*)malloc(sizeof(int32_t) * p->K);
    memset(hist, 0, sizeof(float) * (size_t)C * p->R);
    for (int64_t j = 0; j < C; ++j) {
        frame_f32(p, c, pcm + j * c->hop, j, buf, pw, cl, rw);
        for (int k = 0; k < p->K; ++k)
            if (rw[k] >= 0 && cl[k] >= 0 && cl[k] < C) hist[(size_t)cl[k] * p->R + rw[k]] += pw[k];
    }
    free(buf); free(pw); free(cl); free(rw);
}

int eo_hist_f32(const eo_cfg* c, const float* pcm, int32_t S, int64_t L, float* hist, int32_t threads) {
    if (check_cfg(c) || !pcm || !hist) return -1;
    plan32 p; if (plan32_init(&p, c)) return -3;
    int64_t C = (L >= c->n) ? (L - c->n) / c->hop + 1 : 0;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
#endif
    for (int s = 0; s < S; ++s)
        stream_hist_f32(&p, c, pcm + (size_t)s * L, L, hist + (size_t)s * C * c->rows);
    plan32_free(&p);
    return 0;
}

/* stage "dB + colour" */
typedef struct { float scale, lo, inv_range, gate; } dbmap;
static dbmap make_dbmap(const eo_cfg* c) {
    dbmap m;
    double nn = (double)c->n;
    m.scale = (float)(32.0 / (3.0 * nn * nn) * (double)c->gain * (double)c->gain);
    m.lo = c->db_top - c->db_range;
    m.inv_range = (float)(1.0 / (double)c->db_range);
    m.gate = c->gate_db;
    return m;
}
static inline float cell_db(const dbmap* m, float e) { return 10.0f * log10f(e * m->scale + 1e-20f); }
static inline int cell_index(const dbmap* m, float db) {
    float v = (db - m->lo) * m->inv_range;
    v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
    if (db < m->gate) v = 0.0f;
    return (int)(v * 255.0f + 0.5f);
}

int eo_batch_f32(const eo_cfg* c, const float* pcm, int32_t S, int64_t L, const uint8_t* lut,
                 float* db, uint8_t* rgba, uint8_t* index, int32_t threads) {
    if (check_cfg(c) || !pcm) return -1;
    plan32 p; if (plan32_init(&p, c)) return -3;
    uint8_t deflut[1024];
    if (!lut) { eo_default_lut(deflut); lut = deflut; }
    int64_t C = (L >= c->n) ? (L - c->n) / c->hop + 1 : 0;
    const int R = c->rows;
    dbmap m = make_dbmap(c);
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
#endif
    for (int s = 0; s < S; ++s) {
        float* hist = (float*)malloc(sizeof(float) * (size_t)C * R);
        stream_hist_f32(&p, c, pcm + (size_t)s * L, L, hist);
        size_t base = (size_t)s * C * R;
        for (size_t i = 0; i < (size_t)C * R; ++i) {
            float d = cell_db(&m, hist[i]);
            int ix = cell_index(&m, d);
            if (db) db[base + i] = d;
            if (index) index[base + i] = (uint8_t)ix;
            if (rgba) memcpy(rgba + 4 * (base + i), lut + 4 * ix, 4);
        }
        free(hist);
    }
    plan32_free(&p);
    return 0;
}

int eo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---- float64 textbook method ------------------------------------------- */
/* Three explicitly windowed DFTs (SURVEY.md §8a rows "Windows", "STFT x3",
 * "Reassign"):  h = periodic Hann, th = (n-c) h, dh = (pi/N) sin(2 pi n/N).
 *   t-hat = j H + c + Re(X_th conj X_h)/P          [samples]
 *   k-hat = k - (N/2pi) Im(X_dh conj X_h)/P        [bins]  (f-hat = k-hat fs/N)
 * Independent of the bit model on purpose: separate FFT, separate windows. */
static void fft64(int n, int log2n, double* re, double* im) {
    /* iterative radix-2 decimation-in-time with explicit bit reversal */
    for (unsigned i = 0; i < (unsigned)n; ++i) {
        unsigned r = bitrev(i, log2n);
        if (r > i) { double t = re[i]; re[i] = re[r]; re[r] = t; t = im[i]; im[i] = im[r]; im[r] = t; }
    }
    for (int len = 2; len <= n; len <<= 1) {
        int h = len / 2;
        for (int blk = 0; blk < n; blk += len)
            for (int j = 0; j < h; ++j) {
                double a = -2.0 * EO_PI * (double)j / (double)len;
                double wr = cos(a), wi = sin(a);
                double xr = re[blk + j + h], xi = im[blk + j + h];
                double tr = xr * wr - xi * wi, ti = xr * wi + xi * wr;
                re[blk + j + h] = re[blk + j] - tr; im[blk + j + h] = im[blk + j] - ti;
                re[blk + j] += tr; im[blk + j] += ti;
            }
    }
}

int eo_frames_f64(const eo_cfg* c, const float* pcm, int64_t L, int64_t frame0, int64_t nframes,
                  double* power, double* that, double* khat, int32_t* col, int32_t* row) {
    if (check_cfg(c) || !pcm) return -1;
    if (frame0 < 0 || (frame0 + nframes - 1) * c->hop + c->n > L) return -2;
    const int n = c->n, K = n / 2 + 1, l2 = ilog2(n), half = n / 2, R = c->rows;
    const int D = latency_cols(c);
    double* e = (double*)malloc(sizeof(double) * (R + 1));
    make_edges64(c, e);
    double* w = (double*)malloc(sizeof(double) * 6 * n);
    double *hr = w, *hi = w + n, *tr = w + 2 * n, *ti = w + 3 * n, *dr = w + 4 * n, *di = w + 5 * n;
    double fs_peak = (double)n / 4.0, pfloor = (double)c->power_floor * fs_peak * fs_peak;
    for (int64_t f = 0; f < nframes; ++f) {
        int64_t j = frame0 + f;
        const float* x = pcm + j * c->hop;
        for (int i = 0; i < n; ++i) {
            double a = 2.0 * EO_PI * (double)i / (double)n;
            double hw = 0.5 - 0.5 * cos(a), dw = (EO_PI / (double)n) * sin(a);
            hr[i] = x[i] * hw; hi[i] = 0; tr[i] = x[i] * hw * (double)(i - half); ti[i] = 0;
            dr[i] = x[i] * dw; di[i] = 0;
        }
        fft64(n, l2, hr, hi); fft64(n, l2, tr, ti); fft64(n, l2, dr, di);
        for (int k = 0; k < K; ++k) {
            double P = hr[k] * hr[k] + hi[k] * hi[k];
            size_t o = (size_t)f * K + k;
            power[o] = P;
            int32_t cj = (int32_t)j, rw = -1;
            double th_ = (double)j * c->hop + half, kh_ = (double)k;
            if (P >= pfloor && P > 0) {
                if (c->reassign) {
                    double ts = (tr[k] * hr[k] + ti[k] * hi[k]) / P;               /* Re(Xth conj Xh)/P */
                    double ks = -((double)n / (2.0 * EO_PI)) * (di[k] * hr[k] - dr[k] * hi[k]) / P; /* -(N/2pi) Im(Xdh conj Xh)/P */
                    th_ += ts; kh_ += ks;
                    double cf = floor(ts / (double)c->hop + 0.5);
                    if (fabs(cf) <= (double)D) { cj = (int32_t)j + (int32_t)cf; rw = row_lookup_f64(e, R, kh_); }
                } else {
                    rw = row_lookup_f64(e, R, kh_);
                }
            }
            if (that) that[o] = th_;
            if (khat) khat[o] = kh_;
            if (col) col[o] = cj;
            if (row) row[o] = rw;
        }
    }
    free(w); free(e);
    return 0;
}
