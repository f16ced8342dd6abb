// emspec_api.cpp — the C ABI of libemspec (include/emspec.h): engine, plan
// cache, streaming state, host<->device staging.  Compiled with hipcc.
//
// No reference FFI exists to mirror (reference source is private,
// /root/reference/README.md:73); the contract is SURVEY.md §8(b).
// There is deliberately NO CPU path here: without a gfx950 device
// emspec_create fails.
#include "emspec_engine.h"
#ifdef EMSPEC_DIAG
#pragma GCC visibility push(default)
#include "../../include/emspec_debug.h"
#pragma GCC visibility pop
#endif
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <map>
#include <new>
#include <string>
#include <vector>

using namespace emspec;

namespace {
thread_local std::string g_create_error = "";
}  // namespace

namespace emspec {
int fail(const emspec_engine* e, int code, const std::string& msg) {
    if (e) e->err = msg; else g_create_error = msg;
    return code;
}
int grow(emspec_engine* e, void** ptr, size_t* have, size_t want) {
    if (*have >= want) return EMSPEC_OK;
    if (*ptr) { HIPCHK(e, hipFree(*ptr)); *ptr = nullptr; *have = 0; }
    HIPCHK(e, hipMalloc(ptr, want));   // (hipErrorOutOfMemory -> EMSPEC_ERR_OUT_OF_MEMORY; *ptr stays null, *have 0)
    *have = want;
    return EMSPEC_OK;
}
}  // namespace emspec

namespace emspec {   // (internal linkage is not needed: the library exports only what emspec.map lists; emspec_live.cpp uses these)

void default_lut(uint8_t* lut) {
    // 5-stop gradient measured from the reference's settings screenshot
    // (assets/settings.png): 0/25/50/75/100 %.
    static const int stops[5][3] = {{0, 0, 0}, {80, 0, 80}, {200, 50, 50}, {255, 150, 0}, {255, 255, 200}};
    for (int i = 0; i < 256; ++i) {
        const int pos = 4 * i;
        const int seg = pos >= 3 * 255 ? 3 : pos / 255;
        const int w1 = pos - seg * 255, w0 = 255 - w1;
        for (int c = 0; c < 3; ++c) lut[4 * i + c] = (uint8_t)((stops[seg][c] * w0 + stops[seg + 1][c] * w1 + 127) / 255);
        lut[4 * i + 3] = 255;
    }
}

// (oracle/emspec_exact.c:ex_cos_sin states the same operations)
/* cos and sin of a in [0, pi/4] by their Taylor series in Horner form, plain binary64 operations in this order (no
 * libm call: glibc's sincos(), which gcc substitutes for a cos()/sin() pair, and its separate cos()/sin() differ in the
 * last bit for some arguments, so a table built from libm depends on the compiler).  Truncation < 3e-18; result
 * within about one ulp. */
void cos_sin_octant(double a, double* c, double* s) {
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

// (oracle/emspec_oracle.c: eo_spec_pow states the same operations)
/* ratio^x by a SPECIFIED evaluation: exp2(x * log2(ratio)) from plain IEEE binary64 operations in this
 * order - no libm, whose pow() is not correctly rounded and differs between C libraries, so a table built from it would
 * depend on the host.  log2 by the atanh series on the mantissa folded into [1/sqrt2, sqrt2]; 2^f, |f| <= 1/2,
 * by the Taylor series of e^(f ln 2) in Horner form (truncation < 4e-18); scaling by 2^i is exact.  Within ~3 ulp of the
 * real value; what matters is that every build produces the same bits. */
static double spec_log2(double x) {
    uint64_t u;
    memcpy(&u, &x, 8);
    int e = (int)((u >> 52) & 0x7ff) - 1023;
    u = (u & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double m;
    memcpy(&m, &u, 8);
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
    const uint64_t su = (uint64_t)(1023 + (long long)i) << 52;        /* 2^i, |i| < 1000 */
    double sd;
    memcpy(&sd, &su, 8);
    return p * sd;
}
static double spec_pow(double ratio, double x) {
    if (x == 1.0) return ratio; /* the axis ends exactly at fmax (as pow(ratio, 1) would) */
    return spec_exp2(x * spec_log2(ratio));
}

int latency(int n, int hop, int reassign) { return reassign ? (n + 2 * hop - 1) / (2 * hop) : 0; }

int check_shape(const emspec_engine* e, int n, int hop) {
    if (!supported_fft(n)) return fail(e, EMSPEC_ERR_INVALID_ARG, "fft size must be a power of two in [256,16384]");
    if (hop < 1 || hop > n) return fail(e, EMSPEC_ERR_INVALID_ARG, "hop must be in [1, fft size]");
    return EMSPEC_OK;
}

// DESIGN.md §3 "Tables": evaluated in double, rounded once to float.
int get_plan(emspec_engine* e, int n, Plan** out) {
    auto it = e->plans.find(n);
    if (it != e->plans.end()) { *out = &it->second; return EMSPEC_OK; }
    Plan p;
    p.n = n;
    p.h_tw.resize(n);
    const double pi = 3.14159265358979323846;
    for (int q = 0; q < n / 2; ++q) {
        const double a = 2.0 * pi * (double)q / (double)n;
        p.h_tw[2 * q] = (float)std::cos(a);
        p.h_tw[2 * q + 1] = (float)(-std::sin(a));
    }
    // Second quarter by symmetry: tw[q + N/4] = -j tw[q] = (tw[q].im, -tw[q].re).  With a correctly rounded libm this is
    // what cos/sin give anyway (checked for every N here); writing it down makes it a property of the table that the
    // kernels may rely on (fused_n16384.hip.inc loads 8 pass-1 twiddles instead of 15).  oracle/emspec_oracle.c does the same.
    for (int q = 0; q < n / 4; ++q) {
        p.h_tw[2 * (q + n / 4)] = p.h_tw[2 * q + 1];
        p.h_tw[2 * (q + n / 4) + 1] = -p.h_tw[2 * q];
    }
    p.h_tw[2 * (n / 4)] = 0.0f;       // quarter turn is exact: (0,-1)
    p.h_tw[2 * (n / 4) + 1] = -1.0f;
    const int R = e->cfg.rows;
    p.h_ebin.resize(R + 1);
    const double ratio = (double)e->cfg.fmax_hz / (double)e->cfg.fmin_hz;
    for (int r = 0; r <= R; ++r)
        p.h_ebin[r] = e->custom_edges_hz.empty()
                          ? (float)((double)e->cfg.fmin_hz * spec_pow(ratio, (double)r / (double)R) * (double)n /
                                    (double)e->cfg.sample_rate)
                          : (float)((double)e->custom_edges_hz[r] * (double)n / (double)e->cfg.sample_rate);
    for (int r = 0; r < R; ++r)
        if (!(p.h_ebin[r] < p.h_ebin[r + 1])) return fail(e, EMSPEC_ERR_INVALID_ARG, "row edges are not strictly increasing in float32 (too many rows for this range)");
    HIPCHK(e, hipMalloc(&p.d_tw, sizeof(float) * n));
    HIPCHK(e, hipMalloc(&p.d_ebin, sizeof(float) * (R + 1)));
    HIPCHK(e, hipMemcpy(p.d_tw, p.h_tw.data(), sizeof(float) * n, hipMemcpyHostToDevice));
    HIPCHK(e, hipMemcpy(p.d_ebin, p.h_ebin.data(), sizeof(float) * (R + 1), hipMemcpyHostToDevice));
    if (e->exact()) {
        // the same tables in binary64 (oracle/emspec_exact.c: ex_twiddle, eo_edges64)
        std::vector<double> tw((size_t)n), eb((size_t)R + 1);
        for (int q = 0; q <= n / 8; ++q) {   // first octant by the specified series, second by cos(pi/2 - x) = sin x
            double c, sn;
            cos_sin_octant(2.0 * pi * (double)q / (double)n, &c, &sn);
            tw[2 * q] = c;
            tw[2 * q + 1] = -sn;
            if (q > 0) {
                tw[2 * (n / 4 - q)] = sn;
                tw[2 * (n / 4 - q) + 1] = -c;
            }
        }
        for (int q = 0; q < n / 4; ++q) {
            tw[2 * (q + n / 4)] = tw[2 * q + 1];
            tw[2 * (q + n / 4) + 1] = -tw[2 * q];
        }
        tw[2 * (n / 4)] = 0.0;
        tw[2 * (n / 4) + 1] = -1.0;
        for (int r = 0; r <= R; ++r)
            eb[r] = e->custom_edges_hz.empty()
                        ? (double)e->cfg.fmin_hz * spec_pow(ratio, (double)r / (double)R) * (double)n / (double)e->cfg.sample_rate
                        : (double)e->custom_edges_hz[r] * (double)n / (double)e->cfg.sample_rate;
        for (int r = 0; r < R; ++r)
            if (!(eb[r] < eb[r + 1])) return fail(e, EMSPEC_ERR_INVALID_ARG, "row edges are not strictly increasing");
        p.h_e0 = eb[0];
        p.h_eR = eb[R];
        HIPCHK(e, hipMalloc(&p.d_tw64, sizeof(double) * n));
        HIPCHK(e, hipMalloc(&p.d_ebin64, sizeof(double) * (R + 1)));
        HIPCHK(e, hipMemcpy(p.d_tw64, tw.data(), sizeof(double) * n, hipMemcpyHostToDevice));
        HIPCHK(e, hipMemcpy(p.d_ebin64, eb.data(), sizeof(double) * (R + 1), hipMemcpyHostToDevice));
    }
    auto ins = e->plans.emplace(n, std::move(p));
    *out = &ins.first->second;
    return EMSPEC_OK;
}

PlanDev plan_dev(const emspec_engine* e, const Plan& p, int hop, int reassign) {
    PlanDev d;
    d.tw = p.d_tw;
    d.ebin = p.d_ebin;
    d.rows = e->cfg.rows;
    d.log_rows = e->custom_edges_hz.empty() ? 1 : 0;
    d.D = latency(p.n, hop, reassign);
    d.reassign = reassign ? 1 : 0;
    d.hop = hop;
    d.tscale = (float)((double)p.n / 2.0 / (double)hop);
    const double pk = (double)p.n / 4.0;   // |X_h| of a full-scale sine
    d.pfloor_abs = (float)((double)e->cfg.power_floor * pk * pk);
    // "shared" = 1: other kernels take CUs while a fused launch runs (a communicator with other ranks: RCCL transfers, the
    // gather's pack / expand) -> the shared-device segment plan.  2: this engine's own two-lane host pipeline (emspec_batch
    // runs neighbouring stream-chunks on two HIP streams): two fused launches share the chip, so segments are capped at
    // 1,024 columns (a one-round plan of very long workgroups would degenerate into two rounds); the chunks of that pipeline
    // are small (96 MB of staging), so their segments are short anyway and keep the exclusive plan's low halo.
    d.shared = comm_shares_device(e) ? 1 : 0;
    return d;
}

// EXACT mode: the plan and the dB map in binary64 (oracle/emspec_exact.c: explan_init, eo_batch_exact)
ExactPlanDev exact_plan_dev(const emspec_engine* e, const Plan& p, int hop, int reassign) {
    ExactPlanDev d;
    d.tw = p.d_tw64;
    d.ebin = p.d_ebin64;
    d.rows = e->cfg.rows;
    d.log_rows = e->custom_edges_hz.empty() ? 1 : 0;
    d.D = latency(p.n, hop, reassign);
    d.reassign = reassign ? 1 : 0;
    d.hop = hop;
    d.tscale = (double)p.n / 2.0 / (double)hop;
    const double pk = (double)p.n / 4.0;
    d.pfloor = (double)e->cfg.power_floor * pk * pk;
    int log2n = 0;
    while ((1 << log2n) < p.n) ++log2n;
    d.qscale = std::ldexp(1.0, 52 - (2 * log2n - 4));
    d.pmax = std::ldexp(1.0, 61) / d.qscale;
    d.pfloor64 = 64.0 * d.pfloor;
    d.pmax64 = 64.0 * d.pmax;
    d.qscale64 = d.qscale / 64.0;
    d.e0 = p.h_e0;
    d.eR = p.h_eR;
    d.l2e0 = std::log2((float)p.h_e0);
    d.rscale = (float)d.rows / (std::log2((float)p.h_eR) - d.l2e0);
    return d;
}
ExactDbMap exact_db_map(const emspec_engine* e, int n, const ExactPlanDev& pd) {
    // (oracle/emspec_exact.c: eo_batch_exact states the same operations; every constant is rounded once to binary32)
    ExactDbMap m;
    const double nn = (double)n;
    const double scale = 32.0 / (3.0 * nn * nn) * (double)e->cfg.gain * (double)e->cfg.gain;
    m.sc = (float)(scale * (1.0 / pd.qscale));
    m.lo = (float)((double)e->cfg.db_top - (double)e->cfg.db_range);
    m.inv_range = (float)(1.0 / (double)e->cfg.db_range);
    m.gate = e->cfg.gate_db;
    return m;
}

DbMap db_map(const emspec_engine* e, int n) {
    DbMap m;
    const double nn = (double)n;
    m.scale = (float)(32.0 / (3.0 * nn * nn) * (double)e->cfg.gain * (double)e->cfg.gain);
    m.lo = e->cfg.db_top - e->cfg.db_range;
    m.inv_range = (float)(1.0 / (double)e->cfg.db_range);
    m.gate = e->cfg.gate_db;
    return m;
}

}  // namespace emspec

static void drop_plans(emspec_engine* e);

extern "C" {

int emspec_default_config(emspec_config* c) {
    if (!c) return EMSPEC_ERR_INVALID_ARG;
    std::memset(c, 0, sizeof(*c));
    c->abi_version = EMSPEC_ABI_VERSION;
    c->device = 0;
    c->rows = 1024;
    c->sample_rate = 48000.0f;
    c->fmin_hz = 20.0f;
    c->fmax_hz = 24000.0f;
    c->gain = 1.0f;
    c->db_top = 0.0f;
    c->db_range = 80.0f;
    c->gate_db = -80.0f;
    c->power_floor = 1e-14f;
    return EMSPEC_OK;
}

int emspec_create(const emspec_config* cfg, emspec_engine** out) {
    if (!cfg || !out) return fail(nullptr, EMSPEC_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    if (cfg->abi_version != EMSPEC_ABI_VERSION) return fail(nullptr, EMSPEC_ERR_INVALID_ARG, "abi_version mismatch");
    if (cfg->rows < 64 || cfg->rows > 4096 || cfg->rows % 4) return fail(nullptr, EMSPEC_ERR_INVALID_ARG, "rows must be a multiple of 4 in [64,4096]");
    if (cfg->mode != EMSPEC_MODE_FAST && cfg->mode != EMSPEC_MODE_EXACT) return fail(nullptr, EMSPEC_ERR_INVALID_ARG, "mode must be EMSPEC_MODE_FAST or EMSPEC_MODE_EXACT");
    if (!(cfg->sample_rate > 0) || !(cfg->fmin_hz > 0) || !(cfg->fmax_hz > cfg->fmin_hz) || !(cfg->db_range > 0) ||
        !(cfg->gain > 0) || !(cfg->power_floor >= 0))
        return fail(nullptr, EMSPEC_ERR_INVALID_ARG, "bad sample_rate/fmin/fmax/db_range/gain/power_floor");
    if (cfg->fmax_hz > 0.5f * cfg->sample_rate)
        return fail(nullptr, EMSPEC_ERR_INVALID_ARG, "fmax_hz must not exceed sample_rate/2");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, EMSPEC_ERR_NO_DEVICE, "no HIP device available (libemspec has no CPU path)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, EMSPEC_ERR_NO_DEVICE, "device ordinal out of range");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) return fail(nullptr, EMSPEC_ERR_HIP, "hipGetDeviceProperties failed");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, EMSPEC_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
    emspec_engine* e = new (std::nothrow) emspec_engine();
    if (!e) return fail(nullptr, EMSPEC_ERR_OUT_OF_MEMORY, "out of host memory");
    e->cfg = *cfg;
    e->device = cfg->device;
    e->arch = prop.gcnArchName;
    int rc = EMSPEC_OK;
    do {
        if (hipSetDevice(e->device) != hipSuccess) { rc = fail(nullptr, EMSPEC_ERR_HIP, "hipSetDevice failed"); break; }
        if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) { rc = fail(nullptr, EMSPEC_ERR_HIP, "hipStreamCreate failed"); break; }
        // the copy streams of the host-buffer pipeline, created with the engine: which copy engine a HIP stream's transfers run on
        // follows from the order streams are made in, and two that are made late in a process full of other streams can land on
        // ONE engine - H2D and D2H then take turns (measured: index out 3.4e7 instead of 4.4e7 columns/s)
        if (hipStreamCreateWithFlags(&e->stream_in, hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithFlags(&e->stream_out, hipStreamNonBlocking) != hipSuccess) { rc = fail(nullptr, EMSPEC_ERR_HIP, "hipStreamCreate failed"); break; }
        if (hipMalloc(&e->d_lut, 1024 + 64) != hipSuccess) { rc = fail(nullptr, EMSPEC_ERR_OUT_OF_MEMORY, "hipMalloc(lut) failed"); break; }
        uint8_t lut[1024];
        default_lut(lut);
        if (hipMemcpy(e->d_lut, lut, 1024, hipMemcpyHostToDevice) != hipSuccess) { rc = fail(nullptr, EMSPEC_ERR_HIP, "lut upload failed"); break; }
    } while (0);
    if (rc != EMSPEC_OK) { emspec_destroy(e); return rc; }
    *out = e;
    return EMSPEC_OK;
}

void emspec_destroy(emspec_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (e->stream_in) (void)hipStreamSynchronize(e->stream_in);
    if (e->stream_out) (void)hipStreamSynchronize(e->stream_out);
    comm_destroy(e);
    live_destroy(e);
    drop_plans(e);
    (void)hipFree(e->d_xlow);
    if (e->xlow_event) (void)hipEventDestroy(e->xlow_event);
    (void)hipFree(e->d_lut); (void)hipFree(e->d_hist); (void)hipFree(e->d_stage);
    (void)hipFree(e->d_raw); (void)hipFree(e->d_post); (void)hipFree(e->d_peak);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    if (e->stream_in) (void)hipStreamDestroy(e->stream_in);
    if (e->stream_out) (void)hipStreamDestroy(e->stream_out);
    for (auto& ev : e->pipe_ev) if (ev) (void)hipEventDestroy(ev);
    if (e->h_hdr) (void)hipHostFree(e->h_hdr);
    (void)hipFree(e->d_packscratch);
    delete e;
}

const char* emspec_last_error(const emspec_engine* e) { return e ? e->err.c_str() : g_create_error.c_str(); }
int32_t emspec_mode(const emspec_engine* e) { return e ? e->cfg.mode : -1; }
#ifndef EMSPEC_SOURCES_SHA
#define EMSPEC_SOURCES_SHA "unknown"
#endif
#define EMSPEC_STR2(x) #x
#define EMSPEC_STR(x) EMSPEC_STR2(x)
const char* emspec_build_info(void) { return "emspec abi=" EMSPEC_STR(EMSPEC_ABI_VERSION) " sources=" EMSPEC_SOURCES_SHA " arch=gfx950"; }
int emspec_device_status(emspec_engine* e) {
    if (!e) return EMSPEC_ERR_INVALID_ARG;
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipDeviceSynchronize());
    const int w = read_kernel_error(true);
    if (w != 0) return fail(e, EMSPEC_ERR_HIP, w < 0 ? "the device's kernel error word could not be read"
                                                  : "a kernel's bounded wait timed out (protocol error): the results of the launches since the last check are invalid");
    return EMSPEC_OK;
}
const char* emspec_device_arch(const emspec_engine* e) { return e ? e->arch.c_str() : ""; }
}  // extern "C"
// EXACT mode, N = 4096 / 2048 / 1024: rows of the ring that the no-parking kernel (exact_fused_lr.hip.inc) keeps in global memory, or -1
// when it does not serve this engine's shape or axis.  The axis is served when at most 6 % of a frame's bins lie below row
// rl (on the default log axis at hop 256: rl = 448 of 1024 rows, 38 of 2,049 bins); each of those costs a device-scope
// atomic, so a linear axis (44 % of the bins there) stays on round 4's kernel, which parks instead.
static int exact_lr_rows(const emspec_engine* e, int n, const ExactPlanDev& pd) {
#ifdef EMSPEC_DIAG
    if (const char* ev = getenv("EMSPEC_EXACT_PARKED")) { if (ev[0] == '1') return -1; }   // A/B aid: round 4's kernel
#endif
    const int rl = exact_fused_lr_low_rows(n, pd);
    if (rl <= 0) return rl;
    const double hz = e->custom_edges_hz.empty()
                          ? (double)e->cfg.fmin_hz * spec_pow((double)e->cfg.fmax_hz / (double)e->cfg.fmin_hz, (double)rl / (double)e->cfg.rows)
                          : (double)e->custom_edges_hz[rl];
    const double share = hz / ((double)e->cfg.sample_rate * 0.5);
    return share <= 0.06 ? rl : -1;
}
extern "C" {
int emspec_uses_fused(const emspec_engine* e, int32_t n, int32_t hop, int32_t reassign) {
    if (!e || n < 1 || hop < 1) return 0;
    if (e->exact()) {   // only the shape decides (exact_fused_supported reads rows and D)
        ExactPlanDev pd{};
        pd.rows = e->cfg.rows;
        pd.D = latency(n, hop, reassign);
        return (exact_lr_rows(e, n, pd) >= 0 || exact_fused_supported(n, pd)) ? 1 : 0;
    }
    return fused_supported(n, hop, e->cfg.rows, reassign) ? 1 : 0;
}

static void drop_plans(emspec_engine* e) {
    for (auto& kv : e->plans) {
        (void)hipFree(kv.second.d_tw); (void)hipFree(kv.second.d_ebin);
        (void)hipFree(kv.second.d_tw64); (void)hipFree(kv.second.d_ebin64);
    }
    e->plans.clear();
}

int emspec_set_row_edges_hz(emspec_engine* e, const float* edges_hz, int32_t count) {
    if (!e) return EMSPEC_ERR_INVALID_ARG;
    if (live_pending(e)) return fail(e, EMSPEC_ERR_STATE, "columns are pending; flush or reset before changing the row edges");
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    if (!edges_hz) {   // back to the configured log axis
        e->custom_edges_hz.clear();
        drop_plans(e);
        return EMSPEC_OK;
    }
    if (count != e->cfg.rows + 1) return fail(e, EMSPEC_ERR_INVALID_ARG, "need rows+1 edges");
    for (int r = 0; r <= e->cfg.rows; ++r) {
        const float f = edges_hz[r];
        if (!(f > 0.0f) || !(f <= 0.5f * e->cfg.sample_rate) || (r > 0 && !(f > edges_hz[r - 1])))
            return fail(e, EMSPEC_ERR_INVALID_ARG, "row edges must be strictly increasing in (0, sample_rate/2]");
    }
    e->custom_edges_hz.assign(edges_hz, edges_hz + count);
    drop_plans(e);
    return EMSPEC_OK;
}

int emspec_warped_edges_hz(int32_t rows, float fmin_hz, float fmax_hz, float low_end_boost, float freq_scale,
                           float* out) {
    if (!out || rows < 1 || !(fmin_hz > 0.0f) || !(fmax_hz > fmin_hz) || !(low_end_boost > 0.0f) || !(freq_scale > 0.0f))
        return EMSPEC_ERR_INVALID_ARG;
    const double span = std::log((double)fmax_hz / (double)fmin_hz) / (double)freq_scale;
    for (int r = 0; r <= rows; ++r)
        out[r] = (float)((double)fmin_hz * std::exp(span * std::pow((double)r / (double)rows, (double)low_end_boost)));
    return EMSPEC_OK;
}

int emspec_make_colormap(float brightness, uint8_t* out) {
    if (!out || !(brightness >= 0.0f)) return EMSPEC_ERR_INVALID_ARG;
    static const double stops[5][3] = {{0, 0, 0}, {80, 0, 80}, {200, 50, 50}, {255, 150, 0}, {255, 255, 200}};
    for (int i = 0; i < 256; ++i) {
        const double v = std::min(1.0, ((double)i / 255.0) * ((double)brightness / 0.5));
        const double t = v * 4.0;
        const int s = std::min(3, (int)std::floor(t));
        const double f = t - (double)s;
        for (int c = 0; c < 3; ++c) out[4 * i + c] = (uint8_t)std::floor(stops[s][c] + f * (stops[s + 1][c] - stops[s][c]) + 0.5);
        out[4 * i + 3] = 255;
    }
    return EMSPEC_OK;
}

int emspec_get_row_edges_hz(emspec_engine* e, float* edges_hz, int32_t count) {
    if (!e || !edges_hz) return EMSPEC_ERR_INVALID_ARG;
    if (count != e->cfg.rows + 1) return fail(e, EMSPEC_ERR_INVALID_ARG, "need room for rows+1 edges");
    const int R = e->cfg.rows;
    const double ratio = (double)e->cfg.fmax_hz / (double)e->cfg.fmin_hz;
    for (int r = 0; r <= R; ++r)
        edges_hz[r] = e->custom_edges_hz.empty() ? (float)((double)e->cfg.fmin_hz * spec_pow(ratio, (double)r / (double)R))
                                                 : e->custom_edges_hz[r];
    return EMSPEC_OK;
}

int emspec_host_alloc(size_t bytes, void** out) {
    if (!out || bytes == 0) return EMSPEC_ERR_INVALID_ARG;
    *out = nullptr;
    if (hipHostMalloc(out, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return EMSPEC_ERR_OUT_OF_MEMORY; }
    return EMSPEC_OK;
}
void emspec_host_free(void* p) { if (p) (void)hipHostFree(p); }

int emspec_set_colormap(emspec_engine* e, const uint8_t* rgba) {
    if (!e || !rgba) return fail(e, EMSPEC_ERR_INVALID_ARG, "null argument");
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipMemcpyAsync(e->d_lut, rgba, 1024, hipMemcpyHostToDevice, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return EMSPEC_OK;
}

int64_t emspec_num_columns(int64_t L, int32_t n, int32_t hop) {
    if (n <= 0 || hop <= 0 || L < n) return 0;
    return (L - n) / hop + 1;
}

int32_t emspec_latency_columns(int32_t n, int32_t hop, int32_t reassign) {
    if (n <= 0 || hop <= 0) return 0;
    return latency(n, hop, reassign);
}

int emspec_get_tables(emspec_engine* e, int32_t n, float* edges, float* tw) {
    if (!e) return EMSPEC_ERR_INVALID_ARG;
    int rc = check_shape(e, n, 1);
    if (rc) return rc;
    HIPCHK(e, hipSetDevice(e->device));
    Plan* p;
    if ((rc = get_plan(e, n, &p))) return rc;
    // read back what the kernels actually see
    if (edges) HIPCHK(e, hipMemcpy(edges, p->d_ebin, sizeof(float) * (e->cfg.rows + 1), hipMemcpyDeviceToHost));
    if (tw) HIPCHK(e, hipMemcpy(tw, p->d_tw, sizeof(float) * n, hipMemcpyDeviceToHost));
    return EMSPEC_OK;
}

}  // extern "C"

// Per-bin record workspaces (the shapes without a fused kernel): streams are processed in chunks so that the workspace
// stays bounded.  The budget follows the device: a quarter of what is free (counting what this engine already holds),
// at least 256 MiB, at most `cap` - a chunk only has to cover enough streams to fill the CUs, so a few GiB cost nothing
// measurable, and a fixed 12 GiB (round 3) pinned that much HBM per EXACT engine for its lifetime.  When the allocation
// fails all the same, the chunk is halved and tried again; one stream that does not fit is an out-of-memory error.
static int grow_record_workspace(emspec_engine* e, size_t per_stream, size_t extra, size_t cap, int S, int* chunk_out) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); free_b = cap * 4; }
    size_t budget = (free_b + e->hist_bytes) / 4;
    budget = budget < ((size_t)256 << 20) ? ((size_t)256 << 20) : (budget > cap ? cap : budget);
#ifdef EMSPEC_DIAG
    if (const char* ev = getenv("EMSPEC_RECORD_BUDGET_MB")) budget = (size_t)atol(ev) << 20;   // test hook: force several stream-chunks
#endif
    int chunk = (int)(budget / per_stream);
    chunk = chunk < 1 ? 1 : (chunk > S ? S : chunk);
    for (;;) {
        const int rc = grow(e, (void**)&e->d_hist, &e->hist_bytes, per_stream * (size_t)chunk + extra);
        if (rc == EMSPEC_OK) { *chunk_out = chunk; return EMSPEC_OK; }
        (void)hipGetLastError();
        if (rc != EMSPEC_ERR_OUT_OF_MEMORY || chunk == 1) return rc;
        chunk = (chunk + 1) / 2;
    }
}

// columns of S device-resident streams -> dB / RGBA / index, no display post-process
static int run_columns(emspec_engine* e, const PlanDev& pd, const DbMap& m, const float* pcm, int32_t S, int64_t L,
                       int32_t n, int32_t hop, int32_t reassign, int64_t C, float* db, uint8_t* rgba, uint8_t* index,
                       hipStream_t st) {
    int rc;
    if (fused_supported(n, hop, e->cfg.rows, reassign)) {
        HIPCHK(e, launch_fused(n, pd, m, e->d_lut, pcm, L, S, C, db, rgba, index, st));
        return EMSPEC_OK;
    }
    // generic path: per-bin records (frames_kernel) -> 32-column LDS tiles (tile_scatter_kernel),
    // in chunks of streams so the record workspace stays bounded
    const size_t rec_per_stream = (size_t)C * (n / 2 + 2) * sizeof(uint2);   // frame stride K+1 (even)
    int chunk = 1;
    if ((rc = grow_record_workspace(e, rec_per_stream, 0, (size_t)4 << 30, S, &chunk))) return rc;
    const size_t col_cells = (size_t)C * e->cfg.rows;
    for (int s0 = 0; s0 < S; s0 += chunk) {
        const int sc = (S - s0 < chunk) ? S - s0 : chunk;
        FrameSinks sk;
        sk.records = reinterpret_cast<uint2*>(e->d_hist);
        HIPCHK(e, launch_frames(n, pd, pcm + (size_t)s0 * L, L, sc, 0, C, sk, st));
        HIPCHK(e, launch_tile_scatter(sk.records, n, pd, m, e->d_lut, sc, C, db ? db + s0 * col_cells : nullptr,
                                      rgba ? rgba + 4 * s0 * col_cells : nullptr,
                                      index ? index + s0 * col_cells : nullptr, st));
    }
    return EMSPEC_OK;
}

// The low-row scratch of the no-parking EXACT kernel: grown on demand; a launch on another HIP stream than the previous one
// waits for it (one scratch per engine: two launches must not run side by side on it)
static int exact_lr_prepare(emspec_engine* e, int n, const ExactPlanDev& pd, int rl, int S, int64_t C, hipStream_t st) {
    if (rl <= 0) return EMSPEC_OK;
    int rc;
    const size_t need = exact_fused_lr_scratch_bytes(n, pd, rl, S, C);
    if (need > e->xlow_bytes) {
        if (e->xlow_used) HIPCHK(e, hipEventSynchronize(e->xlow_event));      // the old buffer may still be in use
        if ((rc = grow(e, (void**)&e->d_xlow, &e->xlow_bytes, need))) return rc;
    }
    if (!e->xlow_event) HIPCHK(e, hipEventCreateWithFlags(&e->xlow_event, hipEventDisableTiming));
    if (e->xlow_used) HIPCHK(e, hipStreamWaitEvent(st, e->xlow_event, 0));
    e->xlow_used = true;
    return EMSPEC_OK;
}

// EXACT mode: per-bin (q, key) records (exact_frames_kernel) -> u64 LDS tiles (exact_tile_scatter_kernel), in chunks of
// streams so the record workspace stays bounded
static int run_columns_exact(emspec_engine* e, const Plan& p, const float* pcm, int32_t S, int64_t L, int32_t n, int32_t hop,
                             int32_t reassign, int64_t C, float* db, uint8_t* rgba, uint8_t* index, hipStream_t st) {
    int rc;
    const ExactPlanDev pd = exact_plan_dev(e, p, hop, reassign);
    const ExactDbMap m = exact_db_map(e, n, pd);
    const int rl = exact_lr_rows(e, n, pd);
    if (rl >= 0) {   // one kernel, no records, no parking (exact_fused_lr.hip.inc)
        if ((rc = exact_lr_prepare(e, n, pd, rl, S, C, st))) return rc;
        HIPCHK(e, launch_exact_fused_lr(n, pd, m, e->d_lut, pcm, L, S, C, rl, e->d_xlow, e->xlow_bytes, db, rgba, index, st));
        if (rl > 0) HIPCHK(e, hipEventRecord(e->xlow_event, st));
        return EMSPEC_OK;
    }
    if (exact_fused_supported(n, pd)) {   // one kernel with the ring parked under the planes (exact_fused.hip.inc): any axis
        HIPCHK(e, launch_exact_fused(n, pd, m, e->d_lut, pcm, L, S, C, db, rgba, index, st));
        return EMSPEC_OK;
    }
    const size_t Kp = (size_t)exact_record_stride(n);
    const size_t q_per_stream = (size_t)C * Kp * sizeof(long long), key_per_stream = (size_t)C * Kp * sizeof(uint32_t);
    int chunk = 1;
    // (8 GiB here, 4 GiB for the float32 records: the walking scatter reads a 2D-frame halo per segment, and a stream-chunk's
    // segments get longer with the streams it holds - five streams of configs[4] per chunk: 23 % halo, ten: 12 %; 67.5 -> 66.1 ms per step, and 16 GiB measured 67.0)
    if ((rc = grow_record_workspace(e, q_per_stream + key_per_stream, 256, (size_t)8 << 30, S, &chunk))) return rc;
    const size_t col_cells = (size_t)C * e->cfg.rows;
    // the scatter's low-row scratch (exact.hip.inc: a ring too large for LDS is walked with its sparse low rows in global
    // memory); cleared once per batch - the kernel leaves it zero, this only guards against a launch that was cut short
    const float* ebin_host = p.h_ebin.data();
    // (the last chunk may hold fewer streams, and fewer streams are cut into more segments: size for both)
    const size_t low_need = std::max(exact_scatter_scratch_bytes(n, pd, chunk, C, ebin_host),
                                     S % chunk ? exact_scatter_scratch_bytes(n, pd, S % chunk, C, ebin_host) : (size_t)0);
    if (low_need) {
        if (low_need > e->xlow_bytes) {
            if (e->xlow_used) HIPCHK(e, hipEventSynchronize(e->xlow_event));
            if ((rc = grow(e, (void**)&e->d_xlow, &e->xlow_bytes, low_need))) return rc;
        }
        if (!e->xlow_event) HIPCHK(e, hipEventCreateWithFlags(&e->xlow_event, hipEventDisableTiming));
        if (e->xlow_used) HIPCHK(e, hipStreamWaitEvent(st, e->xlow_event, 0));
        e->xlow_used = true;
        HIPCHK(e, hipMemsetAsync(e->d_xlow, 0, low_need, st));
    }
    for (int s0 = 0; s0 < S; s0 += chunk) {
        const int sc = (S - s0 < chunk) ? S - s0 : chunk;
        ExactSinks sk;
        sk.rec_q = reinterpret_cast<long long*>(e->d_hist);
        sk.rec_key = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(e->d_hist) + ((q_per_stream * chunk + 255) & ~(size_t)255));
        HIPCHK(e, launch_exact_frames(n, pd, pcm + (size_t)s0 * L, L, sc, 0, C, sk, st));
        HIPCHK(e, launch_exact_tile_scatter(sk.rec_q, sk.rec_key, n, pd, m, e->d_lut, sc, C, db ? db + s0 * col_cells : nullptr,
                                            rgba ? rgba + 4 * s0 * col_cells : nullptr,
                                            index ? index + s0 * col_cells : nullptr, st, low_need ? ebin_host : nullptr,
                                            low_need ? e->d_xlow : nullptr, low_need));   // (only what was cleared above: the buffer may be larger)
    }
    if (low_need) HIPCHK(e, hipEventRecord(e->xlow_event, st));
    return EMSPEC_OK;
}

extern "C" {

int emspec_batch_device(emspec_engine* e, const float* pcm, int32_t S, int64_t L, int32_t n, int32_t hop,
                        int32_t reassign, float* db, uint8_t* rgba, uint8_t* index, void* hip_stream) {
    if (!e || !pcm) return fail(e, EMSPEC_ERR_INVALID_ARG, "null argument");
    int rc = check_shape(e, n, hop);
    if (rc) return rc;
    if (S < 1 || S > 65535 || L < n) return fail(e, EMSPEC_ERR_INVALID_ARG, "need 1..65535 streams of at least fft-size samples");
    HIPCHK(e, hipSetDevice(e->device));
    hipStream_t st = (hipStream_t)hip_stream;   // NULL = the HIP default stream
    Plan* p;
    if ((rc = get_plan(e, n, &p))) return rc;
    const PlanDev pd = plan_dev(e, *p, hop, reassign);
    const DbMap m = db_map(e, n);
    const int64_t C = emspec_num_columns(L, n, hop);
    if (!db && !rgba && !index) return EMSPEC_OK;
    if (e->smoothing > 0.0f || e->agc > 0.0f) {
        // raw dB columns into a workspace, then AGC + temporal smoothing into the caller's buffers
        const size_t cells = (size_t)S * C * e->cfg.rows;
        if ((rc = grow(e, (void**)&e->d_raw, &e->raw_bytes, cells * 4))) return rc;
        if ((rc = grow(e, (void**)&e->d_peak, &e->peak_bytes, (size_t)S * C * 8 + 16))) return rc;
        float* outdb = db;
        if (!outdb) { if ((rc = grow(e, (void**)&e->d_post, &e->post_bytes, cells * 4))) return rc; outdb = e->d_post; }
        if ((rc = e->exact() ? run_columns_exact(e, *p, pcm, S, L, n, hop, reassign, C, e->d_raw, nullptr, nullptr, st)
                             : run_columns(e, pd, m, pcm, S, L, n, hop, reassign, C, e->d_raw, nullptr, nullptr, st))) return rc;
        HIPCHK(e, launch_postprocess(e->d_raw, outdb, rgba, index, S, C, e->cfg.rows, e->smoothing, e->agc,
                                     e->cfg.db_top, m, e->d_lut, e->d_peak, e->d_peak + (size_t)S * C, st));
        return EMSPEC_OK;
    }
    if (e->exact()) return run_columns_exact(e, *p, pcm, S, L, n, hop, reassign, C, db, rgba, index, st);
    return run_columns(e, pd, m, pcm, S, L, n, hop, reassign, C, db, rgba, index, st);
}

#ifdef EMSPEC_DIAG   // diagnostic entry points (include/emspec_debug.h): libemspec_diag.so only
// Diagnostic: non-zero if a bounded spin of the decoupled-team fused kernel ever timed out on this device.
int emspec_debug_fused_error(emspec_engine* e) {
    if (!e) return EMSPEC_ERR_INVALID_ARG;
    if (hipSetDevice(e->device) != hipSuccess) return EMSPEC_ERR_HIP;
    (void)hipDeviceSynchronize();
    return fused_read_errflag();
}


// Diagnostic: enqueue a kernel that keeps `groups` workgroups resident for ~usec microseconds on hip_stream.
int emspec_debug_occupy(emspec_engine* e, int32_t groups, int32_t usec, void* hip_stream) {
    if (!e || groups < 1 || groups > 4096 || usec < 1 || usec > 2000000) return fail(e, EMSPEC_ERR_INVALID_ARG, "bad argument");
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, launch_occupy(groups, usec, reinterpret_cast<unsigned*>(e->d_lut) + 256, (hipStream_t)hip_stream));   // (a spare word behind the palette)
    return EMSPEC_OK;
}

// Diagnostic (tests only): evaluate the fused kernels' hinted row lookup and the generic
// binary search on `count` host values of k-hat for fft size n.
int emspec_debug_row_lookup(emspec_engine* e, int32_t n, const float* kh, int64_t count, int32_t* out_hint,
                            int32_t* out_exact) {
    if (!e || !kh || !out_hint || !out_exact || count < 0) return fail(e, EMSPEC_ERR_INVALID_ARG, "null argument");
    int rc = check_shape(e, n, 1);
    if (rc) return rc;
    HIPCHK(e, hipSetDevice(e->device));
    Plan* p;
    if ((rc = get_plan(e, n, &p))) return rc;
    float* d_kh = nullptr; int32_t *d_a = nullptr, *d_b = nullptr;
    hipError_t r = hipMalloc(&d_kh, count * 4 + 4);
    if (r == hipSuccess) r = hipMalloc(&d_a, count * 4 + 4);
    if (r == hipSuccess) r = hipMalloc(&d_b, count * 4 + 4);
    if (r == hipSuccess) r = hipMemcpyAsync(d_kh, kh, count * 4, hipMemcpyHostToDevice, e->stream);
    if (r == hipSuccess) r = launch_row_lookup_probe(p->d_ebin, e->cfg.rows, d_kh, count, d_a, d_b, e->stream);
    if (r == hipSuccess) r = hipMemcpyAsync(out_hint, d_a, count * 4, hipMemcpyDeviceToHost, e->stream);
    if (r == hipSuccess) r = hipMemcpyAsync(out_exact, d_b, count * 4, hipMemcpyDeviceToHost, e->stream);
    if (r == hipSuccess) r = hipStreamSynchronize(e->stream);
    (void)hipFree(d_kh); (void)hipFree(d_a); (void)hipFree(d_b);
    HIPCHK(e, r);
    return EMSPEC_OK;
}

int emspec_debug_recip(emspec_engine* e, const float* d, int64_t count, float* out_short, float* out_ieee) {
    if (!e || !d || !out_short || !out_ieee || count < 0) return fail(e, EMSPEC_ERR_INVALID_ARG, "null argument");
    HIPCHK(e, hipSetDevice(e->device));
    float *d_in = nullptr, *d_a = nullptr, *d_b = nullptr;
    hipError_t r = hipMalloc(&d_in, count * 4 + 4);
    if (r == hipSuccess) r = hipMalloc(&d_a, count * 4 + 4);
    if (r == hipSuccess) r = hipMalloc(&d_b, count * 4 + 4);
    if (r == hipSuccess) r = hipMemcpyAsync(d_in, d, count * 4, hipMemcpyHostToDevice, e->stream);
    if (r == hipSuccess) r = launch_recip_probe(d_in, count, d_a, d_b, e->stream);
    if (r == hipSuccess) r = hipMemcpyAsync(out_short, d_a, count * 4, hipMemcpyDeviceToHost, e->stream);
    if (r == hipSuccess) r = hipMemcpyAsync(out_ieee, d_b, count * 4, hipMemcpyDeviceToHost, e->stream);
    if (r == hipSuccess) r = hipStreamSynchronize(e->stream);
    (void)hipFree(d_in); (void)hipFree(d_a); (void)hipFree(d_b);
    HIPCHK(e, r);
    return EMSPEC_OK;
}

int emspec_debug_recip64(emspec_engine* e, const double* d, int64_t count, double* out_short, double* out_ieee) {
    if (!e || !d || !out_short || !out_ieee || count < 0) return fail(e, EMSPEC_ERR_INVALID_ARG, "null argument");
    HIPCHK(e, hipSetDevice(e->device));
    double *d_in = nullptr, *d_a = nullptr, *d_b = nullptr;
    hipError_t r = hipMalloc(&d_in, count * 8 + 8);
    if (r == hipSuccess) r = hipMalloc(&d_a, count * 8 + 8);
    if (r == hipSuccess) r = hipMalloc(&d_b, count * 8 + 8);
    if (r == hipSuccess) r = hipMemcpyAsync(d_in, d, count * 8, hipMemcpyHostToDevice, e->stream);
    if (r == hipSuccess) r = launch_recip64_probe(d_in, count, d_a, d_b, e->stream);
    if (r == hipSuccess) r = hipMemcpyAsync(out_short, d_a, count * 8, hipMemcpyDeviceToHost, e->stream);
    if (r == hipSuccess) r = hipMemcpyAsync(out_ieee, d_b, count * 8, hipMemcpyDeviceToHost, e->stream);
    if (r == hipSuccess) r = hipStreamSynchronize(e->stream);
    (void)hipFree(d_in); (void)hipFree(d_a); (void)hipFree(d_b);
    HIPCHK(e, r);
    return EMSPEC_OK;
}

// Diagnostic (not part of the product path): run the stamped build of the fused kernel and
// return, per workgroup and wave, the cycles spent in each barrier-delimited phase.
// cycles: [groups][waves][8 slots] uint64 on the HOST; *groups receives the workgroup count and
// *waves the waves per workgroup (call with cycles == NULL first to size the buffer).
int emspec_debug_phase_cycles(emspec_engine* e, const float* pcm_dev, int32_t S, int64_t L, int32_t n, int32_t hop,
                              int32_t reassign, float* db_dev, uint8_t* index_dev, uint64_t* cycles, int64_t* groups,
                              int32_t* waves) {
    if (!e || !pcm_dev || !groups) return fail(e, EMSPEC_ERR_INVALID_ARG, "null argument");
    HIPCHK(e, hipSetDevice(e->device));
    Plan* p;
    int rc;
    if ((rc = get_plan(e, n, &p))) return rc;
    if (e->exact()) {   // the stamped build of exact_fused4096_kernel (waves: 16; slots: exact_fused.hip.inc)
        const ExactPlanDev xpd = exact_plan_dev(e, *p, hop, reassign);
        const ExactDbMap xm = exact_db_map(e, n, xpd);
        const int64_t Cx = emspec_num_columns(L, n, hop);
        const int xrl = exact_lr_rows(e, n, xpd);
        if (xrl >= 0) {   // the stamped build of exact_fused4096_lr_kernel (slots: exact_fused_lr.hip.inc)
            if ((rc = exact_lr_prepare(e, n, xpd, xrl, S, Cx, e->stream))) return rc;
            HIPCHK(e, launch_exact_fused_lr(n, xpd, xm, e->d_lut, pcm_dev, L, S, Cx, xrl, e->d_xlow, e->xlow_bytes, db_dev, nullptr, index_dev, e->stream, nullptr, groups));
            if (xrl > 0) HIPCHK(e, hipEventRecord(e->xlow_event, e->stream));   // (every launch on the scratch: later ones on other streams wait for it)
            if (waves) *waves = 16;
            if (!cycles) return EMSPEC_OK;
            unsigned long long* dl = nullptr;
            const size_t lbytes = (size_t)(*groups) * 16 * 8 * sizeof(unsigned long long);
            HIPCHK(e, hipMalloc(&dl, lbytes));
            hipError_t lr = launch_exact_fused_lr(n, xpd, xm, e->d_lut, pcm_dev, L, S, Cx, xrl, e->d_xlow, e->xlow_bytes, db_dev, nullptr, index_dev, e->stream, dl, groups);
            if (lr == hipSuccess && xrl > 0) lr = hipEventRecord(e->xlow_event, e->stream);
            if (lr == hipSuccess) lr = hipMemcpyAsync(cycles, dl, lbytes, hipMemcpyDeviceToHost, e->stream);
            if (lr == hipSuccess) lr = hipStreamSynchronize(e->stream);
            (void)hipFree(dl);
            HIPCHK(e, lr);
            return EMSPEC_OK;
        }
        if (!exact_fused_supported(n, xpd)) return fail(e, EMSPEC_ERR_INVALID_ARG, "no fused exact kernel for this shape");
        HIPCHK(e, launch_exact_fused(n, xpd, xm, e->d_lut, pcm_dev, L, S, Cx, db_dev, nullptr, index_dev, e->stream, nullptr, groups));
        if (waves) *waves = 16;
        if (!cycles) return EMSPEC_OK;
        unsigned long long* dx = nullptr;
        const size_t xbytes = (size_t)(*groups) * 16 * 8 * sizeof(unsigned long long);
        HIPCHK(e, hipMalloc(&dx, xbytes));
        hipError_t xr = launch_exact_fused(n, xpd, xm, e->d_lut, pcm_dev, L, S, Cx, db_dev, nullptr, index_dev, e->stream, dx, groups);
        if (xr == hipSuccess) xr = hipMemcpyAsync(cycles, dx, xbytes, hipMemcpyDeviceToHost, e->stream);
        if (xr == hipSuccess) xr = hipStreamSynchronize(e->stream);
        (void)hipFree(dx);
        HIPCHK(e, xr);
        return EMSPEC_OK;
    }
    if (!fused_supported(n, hop, e->cfg.rows, reassign)) return fail(e, EMSPEC_ERR_INVALID_ARG, "no fused kernel for this shape");
    const PlanDev pd = plan_dev(e, *p, hop, reassign);
    const DbMap m = db_map(e, n);
    const int64_t C = emspec_num_columns(L, n, hop);
    HIPCHK(e, launch_fused(n, pd, m, e->d_lut, pcm_dev, L, S, C, db_dev, nullptr, index_dev, e->stream, nullptr, groups));
    if (waves) *waves = fused_waves_per_group();
    if (!cycles) return EMSPEC_OK;
    unsigned long long* d = nullptr;
    const size_t bytes = (size_t)(*groups) * fused_waves_per_group() * 8 * sizeof(unsigned long long);
    HIPCHK(e, hipMalloc(&d, bytes));
    hipError_t r = launch_fused(n, pd, m, e->d_lut, pcm_dev, L, S, C, db_dev, nullptr, index_dev, e->stream, d, groups);
    if (r == hipSuccess) r = hipMemcpyAsync(cycles, d, bytes, hipMemcpyDeviceToHost, e->stream);
    if (r == hipSuccess) r = hipStreamSynchronize(e->stream);
    (void)hipFree(d);
    HIPCHK(e, r);
    return EMSPEC_OK;
}
#endif  // EMSPEC_DIAG

}  // extern "C"

// ---- host-buffer batch: a three-stage pipeline over chunks of streams.  H2D copies on their own HIP stream, every kernel on
// the engine's compute stream (so the per-engine workspaces - EXACT records / low-row scratch, display post-process - are
// used by one launch at a time, and no two fused launches share the chip), D2H copies on a third stream; kPipeSets staging
// sets, events between the stages.  PCIe is full duplex: the H2D of chunk k+1, the kernels of chunk k and the D2H of chunk
// k-1 are in flight together.  With `pk` the palette-index columns leave the device as the gather's lossless wire image
// (pack.hip.inc: ~186 B instead of 1,024 B per column on the bench input), one image per stream, tightly packed into the
// caller's buffer in stream order; the host learns each image's size from its 32-byte header, which is copied out behind the
// pack, so the D2H stage of a chunk is enqueued kPipeLag chunks after its kernels.
struct PackedOut { uint8_t* wire; int64_t capacity; int64_t* offsets; };
static constexpr int kPipeSets = 3, kPipeLag = 2;

static int pipe_setup(emspec_engine* e) {
    if (!e->stream_in) HIPCHK(e, hipStreamCreateWithFlags(&e->stream_in, hipStreamNonBlocking));
    if (!e->stream_out) HIPCHK(e, hipStreamCreateWithFlags(&e->stream_out, hipStreamNonBlocking));
    for (int i = 0; i < 3 * kPipeSets; ++i)
        if (!e->pipe_ev[i]) HIPCHK(e, hipEventCreateWithFlags(&e->pipe_ev[i], hipEventDisableTiming));
    return EMSPEC_OK;
}

// One unit of the host pipelines: `sc` whole streams from stream s0 on - or, when the batch has fewer streams than the pipeline
// needs units (BASELINE configs[1] is ONE stream), a run of columns [c0, c0 + cn) of one stream, computed as a batch of its own
// from the frames that reach those columns: D more on either side (a bin moves at most D columns), whose own columns - `skip` in
// front, the rest behind - are computed and left on the device.  Every frame that adds to a kept column is in the run and no
// other frame can reach it, so the kept columns are the whole batch's (EXACT mode: the same bytes; float32: the same sums in
// another order, as between any two launches).
struct PipeItem { int s0, sc; int64_t c0, cn, first_sample, samples, skip, cols; };
static constexpr int kNoThread = -1000, kNoPipeline = -1001;   // pipeline not taken (not EMSPEC_ERR_* values): no helper thread / one unit only

// How many units a batch is cut into.  A unit costs ~0.2 ms (EXACT: 0.4) on the compute stream whatever its size: a launch of
// the fused kernel takes 0.11-0.18 ms however few columns it has - a workgroup WALKS its segment, 2 D halo frames and the ring's
// start-up before the first column leaves (emspec_batch_device on 49 columns: 113 us on the GPU, 5 us to enqueue) - the units'
// kernels run one after the other, and each unit adds ~40 us of event waits and copy start-up.  Behind that, three stages
// overlap: with u units a call takes about
//     max(u x 0.2 ms,  M + (sum - M) / u),   M = the longest of [bytes in / 45 GB/s, kernel time, bytes out / 45 GB/s].
// Until late round 6 the count was fixed (sixteen, or one per stream below that): 8 streams x 2^18 samples took 1.65 ms - eight
// units - for 0.5 ms of copies and kernels.  The kernel rates are the bench line's, rounded; at most sixteen units.
static int pipe_units(bool exact, int n, int64_t columns, size_t bytes_in, size_t bytes_out) {
    const double rate = (n <= 1024 ? 3.4e8 : n <= 2048 ? 2.2e8 : n <= 4096 ? 1.15e8 : n <= 8192 ? 5e7 : 2.2e7) / (exact ? (n > 4096 ? 2.8 : 2.1) : 1.0);
    const double t_in = (double)bytes_in / 45e9, t_out = (double)bytes_out / 45e9, t_k = (double)columns / rate;
    const double longest = std::max(t_in, std::max(t_k, t_out)), sum = t_in + t_k + t_out, per_unit = exact ? 0.4e-3 : 0.2e-3;
    int best = 1;
    double best_t = sum + per_unit;
    for (int u = 2; u <= 16; ++u) {
        const double t = std::max(u * per_unit, longest + (sum - longest) / u);
        if (t < best_t * 0.995) { best = u; best_t = t; }   // (not one unit more for nothing)
    }
#ifdef EMSPEC_DIAG
    if (const char* ev = getenv("EMSPEC_PIPE_CHUNKS")) { const int v = atoi(ev); if (v >= 1) best = v; }   // A/B aid
#endif
    return best;
}

static std::vector<PipeItem> pipe_items(int S, int64_t L, int64_t C, int n, int hop, int D, size_t per_stream_bytes, bool by_time, int target) {
    std::vector<PipeItem> items;
    // runs of columns: when there are fewer than `target` streams; at least 16,384 columns per run - a unit costs ~0.2 ms
    // (pipe_units) whatever its size, and 16 MB each way over PCIe take 0.35 ms (measured with 2,048-column runs: one stream
    // of 2^22 samples 1.49 ms instead of 0.84 in one piece)
    const int64_t pieces = by_time && S < target ? std::min<int64_t>((target + S - 1) / S, C / 16384) : 1;
    if (pieces > 1) {
        for (int s = 0; s < S; ++s)
            for (int64_t t = 0; t < pieces; ++t) {
                PipeItem it;
                it.s0 = s; it.sc = 1;
                it.c0 = C * t / pieces;
                it.cn = C * (t + 1) / pieces - it.c0;
                const int64_t f0 = std::max<int64_t>(it.c0 - D, 0), f1 = std::min<int64_t>(it.c0 + it.cn + D, C);   // frames [f0, f1)
                it.first_sample = f0 * hop;
                it.samples = (f1 - f0 - 1) * hop + n;
                it.skip = it.c0 - f0;
                it.cols = f1 - f0;
                items.push_back(it);
            }
        return items;
    }
    // chunks of streams: about `target` per batch (pipe_units), bounded by 1 GiB of staging per set; a chunk of a few streams
    // still fills the chip (segments are cut per launch)
    int chunk = (S + target - 1) / target;
    const int fit = (int)(((size_t)1 << 30) / per_stream_bytes);
    chunk = chunk > fit ? fit : chunk;
    chunk = chunk < 1 ? 1 : chunk;
    for (int s0 = 0; s0 < S; s0 += chunk) items.push_back(PipeItem{s0, std::min(chunk, S - s0), 0, C, 0, L, 0, C});
    return items;
}

static int batch_pipeline(emspec_engine* e, const float* pcm, int32_t S, int64_t L, int32_t n, int32_t hop, int32_t reassign,
                          const emspec_out* out, const PackedOut* pk) {
    int rc;
    if ((rc = pipe_setup(e))) return rc;
    const int64_t C = emspec_num_columns(L, n, hop);
    const int R = e->cfg.rows;
    const size_t col_cells = (size_t)C * R;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t in_s = (size_t)L * sizeof(float);
    const bool want_db = out && out->db, want_rgba = out && out->rgba, want_idx = (out && out->index) || pk;
    const size_t wire_s = pk ? (size_t)wire_bound_bytes(C, R) : 0;
    const size_t per_stream = al(in_s) + al(want_db ? col_cells * 4 : 0) + al(want_rgba ? col_cells * 4 : 0) + al(want_idx ? col_cells : 0) + al(wire_s);
    // (an image is one stream's whole run of columns, and the display post-process walks a stream in time order: whole streams there)
    const bool post = e->smoothing > 0.0f || e->agc > 0.0f;
    const int units = pipe_units(e->exact(), n, (int64_t)S * C, (size_t)S * in_s,
                                 pk ? (size_t)S * col_cells / 5 : (size_t)S * col_cells * ((want_db ? 4 : 0) + (want_rgba ? 4 : 0) + (out && out->index ? 1 : 0)));
    const std::vector<PipeItem> items = pipe_items(S, L, C, n, hop, latency(n, hop, reassign), per_stream, !pk && !post, units);
    const int nchunks = (int)items.size();
    if (nchunks < 2 && !pk) return kNoPipeline;   // one unit: nothing to overlap (the caller's plain path)
    // the staging set: every array at the size its largest unit needs
    size_t cap_in = 0, cap_cells = 0;
    int chunk = 1;
    for (const PipeItem& it : items) {
        cap_in = std::max(cap_in, al((size_t)it.samples * 4 * it.sc));
        cap_cells = std::max(cap_cells, (size_t)it.cols * R * it.sc);
        chunk = std::max(chunk, it.sc);
    }
    const size_t cap_db = want_db ? al(cap_cells * 4) : 0, cap_rgba = want_rgba ? al(cap_cells * 4) : 0, cap_idx = want_idx ? al(cap_cells) : 0;
    const size_t set_bytes = cap_in + cap_db + cap_rgba + cap_idx + al(wire_s) * chunk;
    if ((rc = grow(e, (void**)&e->d_stage, &e->stage_bytes, kPipeSets * set_bytes + 1024))) return rc;
    if (pk) {
        if ((rc = grow(e, (void**)&e->d_packscratch, &e->packscratch_bytes, wire_scratch_bytes(C)))) return rc;
        const size_t hb = (size_t)kPipeSets * chunk * 32;
        if (hb > e->h_hdr_bytes) {
            if (e->h_hdr) (void)hipHostFree(e->h_hdr);
            e->h_hdr = nullptr; e->h_hdr_bytes = 0;
            HIPCHK(e, hipHostMalloc((void**)&e->h_hdr, hb, hipHostMallocDefault));
            e->h_hdr_bytes = hb;
        }
        pk->offsets[0] = 0;
    }
    hipEvent_t* ev_in = e->pipe_ev;
    hipEvent_t* ev_comp = e->pipe_ev + kPipeSets;
    hipEvent_t* ev_out = e->pipe_ev + 2 * kPipeSets;
    struct Set { float* pcm; float* db; uint8_t* rgba; uint8_t* idx; uint8_t* wire; };
    auto set_of = [&](int b) {
        char* base = e->d_stage + (size_t)b * set_bytes;
        Set q;
        q.pcm = (float*)base; base += cap_in;
        q.db = cap_db ? (float*)base : nullptr; base += cap_db;
        q.rgba = cap_rgba ? (uint8_t*)base : nullptr; base += cap_rgba;
        q.idx = cap_idx ? (uint8_t*)base : nullptr; base += cap_idx;
        q.wire = wire_s ? (uint8_t*)base : nullptr;
        return q;
    };
    hipError_t herr = hipSuccess;
    rc = EMSPEC_OK;
    std::string why;
    // the D2H stage of unit f (its kernels are enqueued; with pk: wait for them, then the images' sizes are known)
    auto drain = [&](int f) {
        const PipeItem& it = items[f];
        const int b = f % kPipeSets, s0 = it.s0, sc = it.sc;
        const Set q = set_of(b);
        if (!pk) {
            const size_t from = (size_t)it.skip * R, to = ((size_t)s0 * C + (size_t)it.c0) * R, cells = (size_t)it.cn * R * sc;
            herr = hipStreamWaitEvent(e->stream_out, ev_comp[b], 0);
            if (herr == hipSuccess && want_db) herr = hipMemcpyAsync(out->db + to, q.db + from, cells * 4, hipMemcpyDeviceToHost, e->stream_out);
            if (herr == hipSuccess && want_rgba) herr = hipMemcpyAsync(out->rgba + 4 * to, q.rgba + 4 * from, cells * 4, hipMemcpyDeviceToHost, e->stream_out);
            if (herr == hipSuccess && out->index) herr = hipMemcpyAsync(out->index + to, q.idx + from, cells, hipMemcpyDeviceToHost, e->stream_out);
        } else {
            herr = hipEventSynchronize(ev_comp[b]);
            for (int i = 0; i < sc && herr == hipSuccess && rc == EMSPEC_OK; ++i) {
                const uint32_t* h = reinterpret_cast<const uint32_t*>(e->h_hdr + ((size_t)b * chunk + i) * 32);
                const uint64_t hcols = (uint64_t)h[2] | ((uint64_t)h[3] << 32), hpay = (uint64_t)h[4] | ((uint64_t)h[5] << 32);
                if (h[0] != 0x32574D45u || (int32_t)h[1] != R || hcols != (uint64_t)C || hpay > (uint64_t)col_cells) {
                    rc = EMSPEC_ERR_HIP; why = "the packed image of a stream carries a bad header"; break;
                }
                const int64_t bytes = wire_fixed_bytes(C, R) + (int64_t)((hpay + 15) & ~(uint64_t)15);
                const int64_t at = pk->offsets[s0 + i];
                if (at + bytes > pk->capacity) {
                    rc = EMSPEC_ERR_INVALID_ARG;
                    why = "wire buffer too small (emspec_wire_bound(columns, rows) per stream always suffices)";
                    break;
                }
                herr = hipMemcpyAsync(pk->wire + at, q.wire + (size_t)i * al(wire_s), (size_t)bytes, hipMemcpyDeviceToHost, e->stream_out);
                // every image STARTS on a 16-byte boundary: the fixed part (32 + 4 C (1 + R/32) bytes) is a multiple of 4 only, so
                // up to 12 bytes of slack follow an image (stream s occupies [offsets[s], offsets[s+1]), slack included; the
                // unpackers take the image's real size from its header)
                pk->offsets[s0 + i + 1] = (at + bytes + 15) & ~(int64_t)15;
            }
        }
        if (herr == hipSuccess && rc == EMSPEC_OK) herr = hipEventRecord(ev_out[b], e->stream_out);
    };
    int drained = 0;
    for (int ci = 0; ci < nchunks && rc == EMSPEC_OK && herr == hipSuccess; ++ci) {
        const PipeItem& it = items[ci];
        const int b = ci % kPipeSets, sc = it.sc;
        const Set q = set_of(b);
        // stage 1: samples in (the set's input is free once the kernels of unit ci - kPipeSets are done)
        if (ci >= kPipeSets) herr = hipStreamWaitEvent(e->stream_in, ev_comp[b], 0);
        if (herr == hipSuccess)
            herr = hipMemcpyAsync(q.pcm, pcm + (size_t)it.s0 * L + (size_t)it.first_sample, (size_t)it.samples * 4 * sc, hipMemcpyHostToDevice, e->stream_in);
        if (herr == hipSuccess) herr = hipEventRecord(ev_in[b], e->stream_in);
        // stage 2: kernels (the set's outputs are free once unit ci - kPipeSets has been copied out)
        if (herr == hipSuccess) herr = hipStreamWaitEvent(e->stream, ev_in[b], 0);
        if (herr == hipSuccess && ci >= kPipeSets) herr = hipStreamWaitEvent(e->stream, ev_out[b], 0);
        if (herr != hipSuccess) break;
        rc = emspec_batch_device(e, q.pcm, sc, it.samples, n, hop, reassign, q.db, q.rgba, q.idx, e->stream);
        if (rc != EMSPEC_OK) break;
        if (pk) {
            for (int i = 0; i < sc && herr == hipSuccess; ++i) {
                uint8_t* w = q.wire + (size_t)i * al(wire_s);
                herr = launch_wire_pack(q.idx + (size_t)i * col_cells, C, R, w, e->d_packscratch, e->stream);
                if (herr == hipSuccess) herr = hipMemcpyAsync(e->h_hdr + ((size_t)b * chunk + i) * 32, w, 32, hipMemcpyDeviceToHost, e->stream);
            }
        }
        if (herr == hipSuccess) herr = hipEventRecord(ev_comp[b], e->stream);
        // stage 3, kPipeLag units behind when the host has to read the sizes first
        const int lag = pk ? kPipeLag : 0;
        while (herr == hipSuccess && rc == EMSPEC_OK && drained <= ci - lag) drain(drained++);
    }
    while (herr == hipSuccess && rc == EMSPEC_OK && drained < nchunks) drain(drained++);
    const hipError_t s1 = hipStreamSynchronize(e->stream_in), s2 = hipStreamSynchronize(e->stream), s3 = hipStreamSynchronize(e->stream_out);
    if (rc != EMSPEC_OK) return why.empty() ? rc : fail(e, rc, why);
    HIPCHK(e, herr);
    HIPCHK(e, s1);
    HIPCHK(e, s2);
    HIPCHK(e, s3);
    if (read_kernel_error(true) > 0) return fail(e, EMSPEC_ERR_HIP, "a kernel's bounded wait timed out (protocol error): results invalid");
    return EMSPEC_OK;
}

// Host buffers in ORDINARY (pageable) memory.  The runtime serves such copies itself (it pins the pages and streams them: ~45 GB/s
// one way on this box) but the call returns only when the copy is done, so from one host thread the samples in, the kernels and the
// columns out run one after the other.  Here the same chunks and staging sets as batch_pipeline, with a second host thread that does
// nothing but the copies out: in, compute and out overlap as they do from page-locked memory.  Returns kNoThread (not an
// EMSPEC_ERR_* value) when the thread cannot be started, before anything was done.
static int batch_pipeline_pageable(emspec_engine* e, const float* pcm, int32_t S, int64_t L, int32_t n, int32_t hop, int32_t reassign,
                                   const emspec_out* out) {
    int rc;
    if ((rc = pipe_setup(e))) return rc;
    const int64_t C = emspec_num_columns(L, n, hop);
    const int R = e->cfg.rows;
    const size_t col_cells = (size_t)C * R;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t in_s = (size_t)L * sizeof(float);
    const bool want_db = out->db != nullptr, want_rgba = out->rgba != nullptr, want_idx = out->index != nullptr;
    const size_t per_stream = al(in_s) + al(want_db ? col_cells * 4 : 0) + al(want_rgba ? col_cells * 4 : 0) + al(want_idx ? col_cells : 0);
    const bool post = e->smoothing > 0.0f || e->agc > 0.0f;
    const int units = pipe_units(e->exact(), n, (int64_t)S * C, (size_t)S * in_s,
                                 (size_t)S * col_cells * ((want_db ? 4 : 0) + (want_rgba ? 4 : 0) + (want_idx ? 1 : 0)));
    const std::vector<PipeItem> items = pipe_items(S, L, C, n, hop, latency(n, hop, reassign), per_stream, !post, units);
    const int nchunks = (int)items.size();
    if (nchunks < 2) return kNoPipeline;   // one unit: nothing to overlap
    size_t cap_in = 0, cap_cells = 0;
    for (const PipeItem& it : items) {
        cap_in = std::max(cap_in, al((size_t)it.samples * 4 * it.sc));
        cap_cells = std::max(cap_cells, (size_t)it.cols * R * it.sc);
    }
    const size_t cap_db = want_db ? al(cap_cells * 4) : 0, cap_rgba = want_rgba ? al(cap_cells * 4) : 0, cap_idx = want_idx ? al(cap_cells) : 0;
    const size_t set_bytes = cap_in + cap_db + cap_rgba + cap_idx;
    if ((rc = grow(e, (void**)&e->d_stage, &e->stage_bytes, kPipeSets * set_bytes + 1024))) return rc;
    hipEvent_t* ev_in = e->pipe_ev;
    hipEvent_t* ev_comp = e->pipe_ev + kPipeSets;
    struct Set { float* pcm; float* db; uint8_t* rgba; uint8_t* idx; };
    auto set_of = [&](int b) {
        char* base = e->d_stage + (size_t)b * set_bytes;
        Set q;
        q.pcm = (float*)base; base += cap_in;
        q.db = cap_db ? (float*)base : nullptr; base += cap_db;
        q.rgba = cap_rgba ? (uint8_t*)base : nullptr; base += cap_rgba;
        q.idx = cap_idx ? (uint8_t*)base : nullptr;
        return q;
    };
    // where unit f's kept columns go in the caller's arrays / come from in the set: cell offsets and count
    auto span_of = [&](const PipeItem& it, size_t& from, size_t& to, size_t& cells) {
        from = (size_t)it.skip * R;
        to = ((size_t)it.s0 * C + (size_t)it.c0) * R;
        cells = (size_t)it.cn * R * it.sc;
    };
    std::mutex mu;
    std::condition_variable cv;
    int launched = 0, drained = 0;      // chunks whose kernels are enqueued / whose columns are in the caller's memory
    bool stop = false;
    hipError_t herr_out = hipSuccess;
    // A caller that allocates its result per call (np.empty, new Uint8Array) hands over pages that were never touched: the runtime's
    // copy then takes a page fault per 4 KB on its one thread (1 GB of palette indices: 90 ms of a 114 ms call).  kTouchers threads
    // write one byte into every page of a chunk's destination before the drainer copies the chunk there (every byte of the outputs
    // is overwritten by the call anyway); on resident pages that costs nothing measurable.
    constexpr int kTouchers = 3;
    int touched[kTouchers] = {};
    auto touch_all = [&](int t) {
        auto touch = [&](void* base, size_t bytes) {
            if (!base || !bytes) return;
            volatile char* p = reinterpret_cast<volatile char*>(base);
            const size_t lo = bytes * (size_t)t / kTouchers, hi = bytes * (size_t)(t + 1) / kTouchers;
            for (size_t a = lo; a < hi; a += 4096) p[a] = 0;
            if (t == kTouchers - 1) p[bytes - 1] = 0;
        };
        for (int f = 0; f < nchunks; ++f) {
            size_t from, to, cells;
            span_of(items[f], from, to, cells);
            if (want_db) touch(out->db + to, cells * 4);
            if (want_rgba) touch(out->rgba + 4 * to, cells * 4);
            if (want_idx) touch(out->index + to, cells);
            std::lock_guard<std::mutex> lk(mu);
            touched[t] = f + 1;
            cv.notify_all();
            if (stop) break;
        }
    };
    auto drain_all = [&] {
        hipError_t r = hipSetDevice(e->device);
        for (int f = 0; f < nchunks && r == hipSuccess; ++f) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] {
                    bool ready = launched > f;
                    for (int t = 0; t < kTouchers; ++t) ready = ready && touched[t] > f;
                    return ready || stop;
                });
                if (launched <= f) break;
                bool ready = true;
                for (int t = 0; t < kTouchers; ++t) ready = ready && touched[t] > f;
                if (!ready) break;
            }
            const int b = f % kPipeSets;
            const Set q = set_of(b);
            size_t from, to, cells;
            span_of(items[f], from, to, cells);
            r = hipStreamWaitEvent(e->stream_out, ev_comp[b], 0);
            if (r == hipSuccess && want_db) r = hipMemcpyAsync(out->db + to, q.db + from, cells * 4, hipMemcpyDeviceToHost, e->stream_out);
            if (r == hipSuccess && want_rgba) r = hipMemcpyAsync(out->rgba + 4 * to, q.rgba + 4 * from, cells * 4, hipMemcpyDeviceToHost, e->stream_out);
            if (r == hipSuccess && want_idx) r = hipMemcpyAsync(out->index + to, q.idx + from, cells, hipMemcpyDeviceToHost, e->stream_out);
            if (r == hipSuccess) r = hipStreamSynchronize(e->stream_out);
            std::lock_guard<std::mutex> lk(mu);
            drained = f + 1;
            cv.notify_all();
        }
        std::lock_guard<std::mutex> lk(mu);
        herr_out = r;
        stop = true;
        cv.notify_all();
    };
    std::thread drainer, touchers[kTouchers];
    try {
        drainer = std::thread(drain_all);
    } catch (const std::exception&) {   // no thread to be had: the caller falls back to one chunk on its own thread
        return kNoThread;
    }
    for (int t = 0; t < kTouchers; ++t) {
        try {
            touchers[t] = std::thread(touch_all, t);
        } catch (const std::exception&) {   // (its share counts as touched: the runtime takes those faults itself)
            std::lock_guard<std::mutex> lk(mu);
            touched[t] = nchunks;
            cv.notify_all();
        }
    }
    hipError_t herr = hipSuccess;
    rc = EMSPEC_OK;
    for (int ci = 0; ci < nchunks && rc == EMSPEC_OK && herr == hipSuccess; ++ci) {
        const PipeItem& it = items[ci];
        const int b = ci % kPipeSets, sc = it.sc;
        const Set q = set_of(b);
        if (ci >= kPipeSets) {   // the set is free once unit ci - kPipeSets has left it
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return drained > ci - kPipeSets || stop; });
            if (drained <= ci - kPipeSets) break;
        }
        herr = hipMemcpyAsync(q.pcm, pcm + (size_t)it.s0 * L + (size_t)it.first_sample, (size_t)it.samples * 4 * sc, hipMemcpyHostToDevice, e->stream_in);
        if (herr == hipSuccess) herr = hipEventRecord(ev_in[b], e->stream_in);
        if (herr == hipSuccess) herr = hipStreamWaitEvent(e->stream, ev_in[b], 0);
        if (herr != hipSuccess) break;
        rc = emspec_batch_device(e, q.pcm, sc, it.samples, n, hop, reassign, q.db, q.rgba, q.idx, e->stream);
        if (rc != EMSPEC_OK) break;
        herr = hipEventRecord(ev_comp[b], e->stream);
        if (herr != hipSuccess) break;
        std::lock_guard<std::mutex> lk(mu);
        launched = ci + 1;
        cv.notify_all();
    }
    {
        std::lock_guard<std::mutex> lk(mu);
        if (launched < nchunks) stop = true;   // an error: the drainer finishes what was launched and leaves
        cv.notify_all();
    }
    drainer.join();
    for (auto& th : touchers)
        if (th.joinable()) th.join();
    const hipError_t s1 = hipStreamSynchronize(e->stream_in), s2 = hipStreamSynchronize(e->stream), s3 = hipStreamSynchronize(e->stream_out);
    if (rc != EMSPEC_OK) return rc;
    HIPCHK(e, herr);
    HIPCHK(e, herr_out);
    HIPCHK(e, s1);
    HIPCHK(e, s2);
    HIPCHK(e, s3);
    if (read_kernel_error(true) > 0) return fail(e, EMSPEC_ERR_HIP, "a kernel's bounded wait timed out (protocol error): results invalid");
    return EMSPEC_OK;
}

bool emspec::host_pinned(const void* p) {
    if (!p) return true;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeHost;
}

extern "C" {

int emspec_batch(emspec_engine* e, const float* pcm, int32_t S, int64_t L, int32_t n, int32_t hop, int32_t reassign,
                 const emspec_out* out) {
    if (!e || !pcm || !out) return fail(e, EMSPEC_ERR_INVALID_ARG, "null argument");
    int rc = check_shape(e, n, hop);
    if (rc) return rc;
    if (S < 1 || L < n) return fail(e, EMSPEC_ERR_INVALID_ARG, "need at least one stream of at least fft-size samples");
    HIPCHK(e, hipSetDevice(e->device));
    if (!out->db && !out->rgba && !out->index) return EMSPEC_OK;
    // Page-locked buffers (emspec_host_alloc): every copy is asynchronous, one host thread drives the three stages.  Pageable
    // buffers: the runtime's copies block the calling thread, so a second thread takes the copies out (round 6: 2.15e7 -> 4.3e7
    // columns/s on the bench shape, the page-locked rate; pinning the caller's buffers per call instead costs more than it
    // saves: 49 ms vs 27 ms for 670 MB).  With fewer than sixteen streams the units are runs of a stream's columns (pipe_items);
    // a batch too short for two units (one stream of < 32,768 columns) goes the plain way: one copy in, the kernels, one copy out.
    if (host_pinned(pcm) && host_pinned(out->db) && host_pinned(out->rgba) && host_pinned(out->index))
        rc = batch_pipeline(e, pcm, S, L, n, hop, reassign, out, nullptr);
    else
        rc = batch_pipeline_pageable(e, pcm, S, L, n, hop, reassign, out);
    if (rc != kNoThread && rc != kNoPipeline) return rc;
    const int64_t C = emspec_num_columns(L, n, hop);
    const size_t col_cells = (size_t)C * e->cfg.rows;
    const size_t in_s = (size_t)L * sizeof(float);
    const size_t db_s = out->db ? col_cells * 4 : 0, rgba_s = out->rgba ? col_cells * 4 : 0, idx_s = out->index ? col_cells : 0;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    // one stream, chunks of streams one after the other so that the staging stays bounded (4 GiB) whatever S is
    const size_t per_stream = al(in_s) + al(db_s) + al(rgba_s) + al(idx_s);
    int chunk = (int)(((size_t)4 << 30) / per_stream);
    chunk = chunk < 1 ? 1 : (chunk > S ? S : chunk);
    if ((rc = grow(e, (void**)&e->d_stage, &e->stage_bytes, (size_t)chunk * per_stream + 1024))) return rc;
    hipStream_t st = e->stream;
    hipError_t herr = hipSuccess;
    for (int s0 = 0; s0 < S && rc == EMSPEC_OK && herr == hipSuccess; s0 += chunk) {
        const int sc = (S - s0 < chunk) ? S - s0 : chunk;
        char* base = e->d_stage;
        float* d_pcm = (float*)base; base += al(in_s) * chunk;
        float* d_db = db_s ? (float*)base : nullptr; base += al(db_s) * chunk;
        uint8_t* d_rgba = rgba_s ? (uint8_t*)base : nullptr; base += al(rgba_s) * chunk;
        uint8_t* d_idx = idx_s ? (uint8_t*)base : nullptr;
        herr = hipMemcpyAsync(d_pcm, pcm + (size_t)s0 * L, in_s * sc, hipMemcpyHostToDevice, st);
        if (herr != hipSuccess) break;
        rc = emspec_batch_device(e, d_pcm, sc, L, n, hop, reassign, d_db, d_rgba, d_idx, st);
        if (rc != EMSPEC_OK) break;
        if (db_s) herr = hipMemcpyAsync(out->db + (size_t)s0 * col_cells, d_db, db_s * sc, hipMemcpyDeviceToHost, st);
        if (herr == hipSuccess && rgba_s) herr = hipMemcpyAsync(out->rgba + 4 * (size_t)s0 * col_cells, d_rgba, rgba_s * sc, hipMemcpyDeviceToHost, st);
        if (herr == hipSuccess && idx_s) herr = hipMemcpyAsync(out->index + (size_t)s0 * col_cells, d_idx, idx_s * sc, hipMemcpyDeviceToHost, st);
        if (herr == hipSuccess && s0 + chunk < S) herr = hipStreamSynchronize(st);   // the next chunk reuses the staging
    }
    const hipError_t s1 = hipStreamSynchronize(st);
    if (rc != EMSPEC_OK) return rc;
    HIPCHK(e, herr);
    HIPCHK(e, s1);
    if (read_kernel_error(true) > 0) return fail(e, EMSPEC_ERR_HIP, "a kernel's bounded wait timed out (protocol error): results invalid");
    return EMSPEC_OK;
}

int emspec_batch_packed(emspec_engine* e, const float* pcm, int32_t S, int64_t L, int32_t n, int32_t hop, int32_t reassign,
                        uint8_t* wire, int64_t wire_capacity, int64_t* offsets) {
    if (!e || !pcm || !wire || !offsets || wire_capacity < 0) return fail(e, EMSPEC_ERR_INVALID_ARG, "null argument");
    int rc = check_shape(e, n, hop);
    if (rc) return rc;
    if (S < 1 || L < n) return fail(e, EMSPEC_ERR_INVALID_ARG, "need at least one stream of at least fft-size samples");
    if ((uint64_t)emspec_num_columns(L, n, hop) * (uint64_t)e->cfg.rows >= (1ull << 32))
        return fail(e, EMSPEC_ERR_INVALID_ARG, "at most 2^32 cells per stream");
    if (e->cfg.rows % 4) return fail(e, EMSPEC_ERR_INVALID_ARG, "the wire image needs rows % 4 == 0");
    HIPCHK(e, hipSetDevice(e->device));
    const PackedOut pk{wire, wire_capacity, offsets};
    return batch_pipeline(e, pcm, S, L, n, hop, reassign, nullptr, &pk);
}

int emspec_wire_unpack_host(const uint8_t* wire, int64_t wire_bytes, int64_t columns, int32_t rows, uint8_t* index_out) {
    if (!wire || !index_out || columns < 1 || rows < 1 || wire_bytes < 32) return EMSPEC_ERR_INVALID_ARG;
    uint32_t h[8];
    memcpy(h, wire, 32);
    const uint64_t hcols = (uint64_t)h[2] | ((uint64_t)h[3] << 32), hpay = (uint64_t)h[4] | ((uint64_t)h[5] << 32);
    const int mw = (rows + 31) >> 5;
    const int64_t fixed = 32 + columns * 4 + columns * (int64_t)mw * 4;
    if (h[0] != 0x32574D45u || (int32_t)h[1] != rows || hcols != (uint64_t)columns || hpay > (uint64_t)columns * (uint64_t)rows ||
        wire_bytes < fixed + (int64_t)hpay)
        return EMSPEC_ERR_INVALID_ARG;
    const uint8_t* offp = wire + 32;
    const uint8_t* maskp = offp + columns * 4;
    const uint8_t* pay = wire + fixed;
    for (int64_t c = 0; c < columns; ++c) {
        uint32_t off;
        memcpy(&off, offp + c * 4, 4);
        uint8_t* dst = index_out + c * (int64_t)rows;
        memset(dst, 0, (size_t)rows);
        uint64_t at = off;
        for (int w = 0; w < mw; ++w) {
            uint32_t m;
            memcpy(&m, maskp + (c * mw + w) * 4, 4);
            while (m) {
                const int bit = __builtin_ctz(m);
                m &= m - 1;
                const int r = w * 32 + bit;
                if (r >= rows || at >= hpay) return EMSPEC_ERR_INVALID_ARG;   // a damaged image must not write or read out of range
                dst[r] = pay[at++];
            }
        }
    }
    return EMSPEC_OK;
}

int emspec_parity_dump_device(emspec_engine* e, const float* pcm, int32_t S, int64_t L, int32_t n, int32_t hop,
                              int32_t reassign, int64_t frame0, int64_t nframes, float* power, int32_t* col,
                              int32_t* row, void* hip_stream) {
    if (!e || !pcm || !power || !col || !row) return fail(e, EMSPEC_ERR_INVALID_ARG, "null argument");
    if (e->exact()) return fail(e, EMSPEC_ERR_STATE, "engine is in EXACT mode: use emspec_parity_dump_exact");
    int rc = check_shape(e, n, hop);
    if (rc) return rc;
    const int64_t C = emspec_num_columns(L, n, hop);
    if (S < 1 || S > 65535 || frame0 < 0 || nframes < 0 || frame0 + nframes > C)
        return fail(e, EMSPEC_ERR_INVALID_ARG, "frame range outside the stream");
    HIPCHK(e, hipSetDevice(e->device));
    hipStream_t st = (hipStream_t)hip_stream;   // NULL = the HIP default stream
    Plan* p;
    if ((rc = get_plan(e, n, &p))) return rc;
    const PlanDev pd = plan_dev(e, *p, hop, reassign);
    FrameSinks sk;
    sk.power = power; sk.col = col; sk.row = row;
    HIPCHK(e, launch_frames(n, pd, pcm, L, S, frame0, nframes, sk, st));
    return EMSPEC_OK;
}

int emspec_parity_dump(emspec_engine* e, const float* pcm, int32_t S, int64_t L, int32_t n, int32_t hop,
                       int32_t reassign, int64_t frame0, int64_t nframes, float* power, int32_t* col, int32_t* row) {
    if (!e || !pcm || !power || !col || !row) return fail(e, EMSPEC_ERR_INVALID_ARG, "null argument");
    if (e->exact()) return fail(e, EMSPEC_ERR_STATE, "engine is in EXACT mode: use emspec_parity_dump_exact");
    int rc = check_shape(e, n, hop);
    if (rc) return rc;
    if (S < 1 || S > 65535 || L < n) return fail(e, EMSPEC_ERR_INVALID_ARG, "need 1..65535 streams of at least fft-size samples");
    if (frame0 < 0 || nframes < 0 || frame0 + nframes > emspec_num_columns(L, n, hop))
        return fail(e, EMSPEC_ERR_INVALID_ARG, "frame range outside the stream");
    HIPCHK(e, hipSetDevice(e->device));
    const size_t nb = (size_t)S * nframes * (n / 2 + 1);
    const size_t b_pcm = (size_t)S * L * sizeof(float);
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    if ((rc = grow(e, (void**)&e->d_stage, &e->stage_bytes, al(b_pcm) + 3 * al(nb * 4) + 256))) return rc;
    char* base = e->d_stage;
    float* d_pcm = (float*)base; base += al(b_pcm);
    float* d_pw = (float*)base; base += al(nb * 4);
    int32_t* d_col = (int32_t*)base; base += al(nb * 4);
    int32_t* d_row = (int32_t*)base;
    HIPCHK(e, hipMemcpyAsync(d_pcm, pcm, b_pcm, hipMemcpyHostToDevice, e->stream));
    if ((rc = emspec_parity_dump_device(e, d_pcm, S, L, n, hop, reassign, frame0, nframes, d_pw, d_col, d_row, e->stream))) return rc;
    HIPCHK(e, hipMemcpyAsync(power, d_pw, nb * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipMemcpyAsync(col, d_col, nb * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipMemcpyAsync(row, d_row, nb * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return EMSPEC_OK;
}

int emspec_parity_dump_exact(emspec_engine* e, const float* pcm, int32_t S, int64_t L, int32_t n, int32_t hop,
                             int32_t reassign, int64_t frame0, int64_t nframes, double* power, int32_t* col, int32_t* row,
                             int64_t* q) {
    if (!e || !pcm || !power || !col || !row) return fail(e, EMSPEC_ERR_INVALID_ARG, "null argument");
    if (!e->exact()) return fail(e, EMSPEC_ERR_STATE, "engine is in FAST mode: use emspec_parity_dump");
    int rc = check_shape(e, n, hop);
    if (rc) return rc;
    if (S < 1 || S > 65535 || L < n) return fail(e, EMSPEC_ERR_INVALID_ARG, "need 1..65535 streams of at least fft-size samples");
    if (frame0 < 0 || nframes < 0 || frame0 + nframes > emspec_num_columns(L, n, hop))
        return fail(e, EMSPEC_ERR_INVALID_ARG, "frame range outside the stream");
    HIPCHK(e, hipSetDevice(e->device));
    Plan* p;
    if ((rc = get_plan(e, n, &p))) return rc;
    const ExactPlanDev pd = exact_plan_dev(e, *p, hop, reassign);
    const size_t nb = (size_t)S * nframes * (n / 2 + 1);
    const size_t b_pcm = (size_t)S * L * sizeof(float);
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    if ((rc = grow(e, (void**)&e->d_stage, &e->stage_bytes, al(b_pcm) + 2 * al(nb * 8) + 2 * al(nb * 4) + 256))) return rc;
    char* base = e->d_stage;
    float* d_pcm = (float*)base; base += al(b_pcm);
    ExactSinks sk;
    sk.power = (double*)base; base += al(nb * 8);
    sk.q = (long long*)base; base += al(nb * 8);
    sk.col = (int32_t*)base; base += al(nb * 4);
    sk.row = (int32_t*)base;
    HIPCHK(e, hipMemcpyAsync(d_pcm, pcm, b_pcm, hipMemcpyHostToDevice, e->stream));
    HIPCHK(e, launch_exact_frames(n, pd, d_pcm, L, S, frame0, nframes, sk, e->stream));
    HIPCHK(e, hipMemcpyAsync(power, sk.power, nb * 8, hipMemcpyDeviceToHost, e->stream));
    if (q) HIPCHK(e, hipMemcpyAsync(q, sk.q, nb * 8, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipMemcpyAsync(col, sk.col, nb * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipMemcpyAsync(row, sk.row, nb * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return EMSPEC_OK;
}

int emspec_set_display(emspec_engine* e, float smoothing, float agc_strength) {
    if (!e) return EMSPEC_ERR_INVALID_ARG;
    if (!(smoothing >= 0.0f && smoothing <= 0.95f) || !(agc_strength >= 0.0f && agc_strength <= 1.0f))
        return fail(e, EMSPEC_ERR_INVALID_ARG, "smoothing must be in [0,0.95], agc_strength in [0,1]");
    e->smoothing = smoothing;
    e->agc = agc_strength;
    return EMSPEC_OK;
}

int emspec_reset(emspec_engine* e) {
    if (!e) return EMSPEC_ERR_INVALID_ARG;
    live_reset(e);   // both streaming sessions: the single-stream calls' and the live multi-stream one (buffers are kept)
    return EMSPEC_OK;
}

// (the streaming calls - emspec_column, emspec_column_flush, emspec_push_samples, emspec_push_columns and the live multi-stream
// session - live in emspec_live.cpp)

}  // extern "C"
