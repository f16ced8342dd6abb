// emspec_device.h — device-side building blocks shared by every kernel.
//
// Arithmetic specification.  The reference source is private
// (/root/reference/README.md:73), so there is no reference file:line for the
// arithmetic; the stage list is SURVEY.md §8(a).  The operation order below IS
// the specification that oracle/emspec_oracle.c's float32 bit model restates
// on the CPU; compile with -ffp-contract=off so the only fused operations are
// the explicit __builtin_fmaf calls.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace emspec {

// Per-(N,hop,config) constants handed to every kernel by value.
struct PlanDev {
    const float2* tw;    // twiddle table, N/2 entries: (cos, -sin)(2*pi*q/N); tw[N/4] == (0,-1) exactly
    const float* ebin;   // row edges in DFT-bin units, rows+1 entries, strictly increasing
    int rows;            // R
    int log_rows;        // 1: edges are log-spaced (hinted lookup is valid); 0: arbitrary monotone table (binary search)
    int D;               // max |column shift| = ceil(N/(2*hop)) (0 when reassign is off)
    int reassign;        // 0/1
    int hop;
    float tscale;        // (N/2)/hop : normalised time shift -> columns
    float pfloor_abs;    // gate on |X_h|^2
    int shared;          // host side only: 1 = the device is shared with a collective (communicator, world > 1); 2 = with the
                         // engine's own second pipeline lane (emspec_batch): segments capped at 1,024 columns
};

// How a launch of a fused (walking) kernel cuts every stream into segments.
//   uniform plan (short_last = 0): grid = (segments, streams); segment g covers columns [g * seglen, (g + 1) * seglen).
//   shared-device plan (short_last = 1): grid = (streams, segments); segment g covers
//     [g * seglen, (g + 1) * seglen) for g < nlong, then pieces of `tail` columns.  Workgroups are dispatched in linear
//     block order, so with blockIdx.y = segment all streams' long segments start first and the launch ends on the
//     short ones: when another kernel (a collective, the gather's pack / expand) holds some CUs and the workgroups no
//     longer fill whole rounds, what is left over at the end is short.  Stream s takes the segment order rotated by s
//     (long and short segments each among themselves).
//   The uniform plan keeps round 1's dispatch order on purpose: with grid = (streams, segments) the N = 16384 kernel,
//   which re-reads its 64 KB sample window every frame and relies on L2 for it, fetched 56 KB instead of 2.8 KB per
//   column from beyond L2 (FETCH_SIZE, with and without the rotation; same speed).  The N = 4096 kernel reads every
//   sample once and is unaffected (6.2 KB per column either way), and it is the one the N > 1 bench runs.
struct SegPlan { int seglen; int nlong; int tail; int short_last; };
__device__ __forceinline__ bool seg_of_block(const SegPlan& sp, int64_t C, int& s, int64_t& c0, int64_t& c1) {
    if (!sp.short_last) {
        s = (int)blockIdx.y;
        c0 = (int64_t)blockIdx.x * sp.seglen;
        c1 = (c0 + sp.seglen < C) ? c0 + sp.seglen : C;
        return c0 < C;
    }
    s = (int)blockIdx.x;
    const int y = (int)blockIdx.y, ny = (int)gridDim.y;
    const int nl = sp.nlong < ny ? sp.nlong : ny;
    const int g = y < nl ? (y + s) % nl : nl + ((y - nl) + s) % (ny - nl);
    const int64_t len = g < sp.nlong ? sp.seglen : sp.tail;
    c0 = g < sp.nlong ? (int64_t)g * sp.seglen : (int64_t)sp.nlong * sp.seglen + (int64_t)(g - sp.nlong) * sp.tail;
    c1 = (c0 + len < C) ? c0 + len : C;
    return c0 < C;
}

struct DbMap {           // stage "dB + colour"
    float scale;         // 32/(3 N^2) * gain^2 : full-scale sine -> 1.0
    float lo;            // db_top - db_range
    float inv_range;     // 1/db_range
    float gate;          // gate_db
};

// EXACT mode (exact.hip.inc): the same plan in binary64, plus the fixed point of the histogram
struct ExactPlanDev {
    const double2* tw;   // N/2 entries (cos, -sin)(2 pi q / N) in binary64, same symmetries as the float32 table
    const double* ebin;  // rows+1 row edges in DFT-bin units, binary64
    int rows, log_rows, D, reassign, hop;
    double tscale;       // (N/2)/hop
    double pfloor;       // gate on |X_h|^2
    double pmax;         // upper gate = 2^61 / qscale
    double qscale;       // fixed-point units per unit of |X_h|^2: 2^52 / (N/4)^2, a power of two
    // the same gates and scale on den = 64 P (exact powers of two away: the branch-free core compares and converts den itself)
    double pfloor64, pmax64, qscale64;
    // for the branch-free per-bin core of exact_fused.hip.inc (log-spaced rows): the table's ends and the float32 log2 hint
    double e0, eR;       // ebin[0], ebin[rows]
    float l2e0, rscale;  // log2(e0), rows / (log2(eR) - log2(e0))
};
// stage "dB + colour" of the EXACT mode: binary32; sc = (float)(scale / qscale): fixed-point sum -> dB argument
struct ExactDbMap { float sc, lo, inv_range, gate; };

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// canonical complex product d * w  (see oracle fft_dif_f32)
__device__ __forceinline__ float2 cmul_tw(float2 d, float2 w) {
    float t = d.y * w.y;
    float u = d.y * w.x;
    return make_float2(__builtin_fmaf(d.x, w.x, -t), __builtin_fmaf(d.x, w.y, u));
}

// LDS index padding for the in-place exchange buffer: one complex slot of pad
// per 16 keeps the stride-2^b accesses of every pass off a single bank pair.
__device__ __forceinline__ constexpr int padi(int p) { return p + (p >> 4); }
template <int N> struct PaddedSize { static constexpr int value = N + (N >> 4); };

// Orders this wave's earlier LDS writes before its later LDS reads.  The LDS pipe executes one
// wave's instructions in order, so nothing is waited for; this only stops the compiler from
// moving accesses across the point.  Used where a pass reads only what the same wave wrote.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Radix-2 DIF stages S0 .. S0+R-1 of an N = 2^LOG2N point FFT, performed on the
// 2^R values a thread holds in registers.  Register i is the element with
// global index  p = hi*2^(B0+R) + i*2^B0 + lo,  B0 = LOG2N-S0-R.
// Stage s pairs (p, p+m), m = N>>(s+1), twiddle index q = (p mod m) << s.
template <int LOG2N, int S0, int R>
__device__ __forceinline__ void fft_stages(float2 (&v)[1 << R], int lo, const float2* __restrict__ tw) {
    constexpr int N = 1 << LOG2N;
    constexpr int B0 = LOG2N - S0 - R;
#pragma unroll
    for (int u = 0; u < R; ++u) {
        const int mloc = 1 << (R - 1 - u);
        const int s = S0 + u;
#pragma unroll
        for (int i = 0; i < (1 << R); ++i) {
            if (i & mloc) continue;
            float2 a = v[i], b = v[i + mloc];
            v[i] = cadd(a, b);
            float2 d = csub(a, b);
            if constexpr (B0 == 0) {
                // last pass: lo == 0, q is a compile-time constant
                const int q = (i & (mloc - 1)) << s;
                // tw[N/8] = (c, -c) and tw[3N/8] = (-c, -c) with c = float(cos(pi/4)) in every table (emspec_create builds
                // the second quarter from the first by the quarter-turn symmetry; tests/test_oracle.py pins the value):
                // as literals they cost no load - from the table they were two vector loads per frame, pl.tw not being
                // provably read-only
                constexpr float kC8 = 0.70710677f;
                if (q == 0) v[i + mloc] = d;
                else if (q == N / 4) v[i + mloc] = make_float2(d.y, -d.x);
                else if (q == N / 8) v[i + mloc] = cmul_tw(d, make_float2(kC8, -kC8));
                else if (q == 3 * (N / 8)) v[i + mloc] = cmul_tw(d, make_float2(-kC8, -kC8));
                else v[i + mloc] = cmul_tw(d, tw[q]);
            } else {
                // the upper half of a stage's butterflies lies a quarter turn beyond the lower half and the table is
                // quarter-turn symmetric (tw[q + N/4] = (tw[q].y, -tw[q].x) exactly): one load serves both
                const int mm = i & (mloc - 1);
                const bool rot = mloc >= 2 && mm >= mloc / 2;
                const int q = (((rot ? mm - mloc / 2 : mm) << B0) + lo) << s;
                const float2 w = tw[q];   // tw[0]=(1,-0), tw[N/4]=(0,-1): exact
                v[i + mloc] = cmul_tw(d, rot ? make_float2(w.y, -w.x) : w);
            }
        }
    }
}

// The same stages with the twiddle of slot e supplied by w(e); slots are numbered stage by
// stage: slot = 2^R - 2*mloc + (i mod mloc), mloc = 2^(R-1-u) for local stage u.
// LEAN: a scheduling barrier after the first stage keeps the later stages' twiddle reads from being hoisted above
// it (with 16 points that is 14 fewer live registers at the pass's peak; fused_n16384.hip.inc needs them).
template <int R, class W, bool LEAN = false>
__device__ __forceinline__ void fft_stages_w(float2 (&v)[1 << R], const W& w) {
#pragma unroll
    for (int u = 0; u < R; ++u) {
        const int mloc = 1 << (R - 1 - u);
#pragma unroll
        for (int i = 0; i < (1 << R); ++i) {
            if (i & mloc) continue;
            const float2 a = v[i], b = v[i + mloc];
            v[i] = cadd(a, b);
            v[i + mloc] = cmul_tw(csub(a, b), w((1 << R) - 2 * mloc + (i & (mloc - 1))));
        }
        if constexpr (LEAN) { if (u == 0) __builtin_amdgcn_sched_barrier(0); }
    }
}
// twiddle-table index of slot e (radix-16 pass starting at stage S0) for low index lo
template <int LOG2N, int S0>
__device__ __forceinline__ int tw_slot_index16(int e, int lo) {
    constexpr int B0 = LOG2N - S0 - 4;
    const int u = e < 8 ? 0 : (e < 12 ? 1 : (e < 14 ? 2 : 3));
    const int mloc = 8 >> u, mm = e - (16 - 2 * mloc);
    return ((mm << B0) + lo) << (S0 + u);
}
struct TwLdsStrided {   // slot e of column lo at base[e*stride]  (base already offset by lo)
    const float2* base; int stride;
    __device__ __forceinline__ float2 operator()(int e) const { return base[e * stride]; }
};
// The same fifteen slots from EIGHT reads: within a stage the upper half of the slots lies exactly a quarter turn
// (N/4 table entries) beyond the lower half - slot mm + mloc/2 has exponent + (mloc/2 << B0) << (S0 + u) = + 2^(LOG2N-2) -
// and the table is quarter-turn symmetric, tw[q + N/4] = (tw[q].y, -tw[q].x) exactly (enforced where the table is built,
// DESIGN.md §3.1); negation is exact, so the butterflies are bit-identical.  (e is a compile-time constant after
// unrolling; the two uses of a slot are one LDS read.)
struct TwLdsSym16 {
    const float2* base; int stride;
    __device__ __forceinline__ float2 operator()(int e) const {
        const int half = e < 8 ? 4 : (e < 12 ? 2 : (e < 14 ? 1 : 0));
        const int first = e < 8 ? 0 : (e < 12 ? 8 : (e < 14 ? 12 : 14));
        if (half && e - first >= half) {
            const float2 w = base[(e - half) * stride];
            return make_float2(w.y, -w.x);
        }
        return base[e * stride];
    }
};

// row = largest r in [0,R) with ebin[r] <= kh ; -1 unless ebin[0] <= kh < ebin[R].
// Exact float compares against the table: identical to the oracle's binary search.
__device__ __forceinline__ int row_lookup(const float* eb, int R, int wtop, float kh) {
    if (!(kh >= eb[0]) || !(kh < eb[R])) return -1;
    int lo = 0;
    for (int w = wtop; w > 0; w >>= 1) {
        int mid = lo + w;
        if (mid < R && eb[mid] <= kh) lo = mid;
    }
    return lo;
}

// Row lookup with a log2 hint and an exact +-1 correction against the table: same result as
// the binary search.  The hint is off by at most one row (v_log_f32 error and the float32
// rounding of the edges are both << a row, which is >= 0.67 % wide); the two exact compares
// against the table decide.  Verified around every edge by tests/test_gpu_parity.py
// (emspec_debug_row_lookup).  Garbage / NaN inputs map to row -1 through the range test.
struct HintLookup {
    const float* eb; int R; float e0, eR, l2e0, rscale; int wtop;   // wtop > 0: table is not log-spaced, search it
    __device__ __forceinline__ void init(const float* table, const float* gtable, int rows, int log_rows = 1) {
        eb = table; R = rows; e0 = gtable[0]; eR = gtable[rows];
        l2e0 = log2f(e0);
        rscale = (float)rows / (log2f(eR) - l2e0);
        wtop = 0;
        if (!log_rows) { wtop = 1; while (wtop * 2 < rows) wtop *= 2; }
    }
    __device__ __forceinline__ bool in_range(float kh) const { return (kh >= e0) && (kh < eR); }
    // row for an in-range kh (any kh is safe: the hint is clamped into the table)
    __device__ __forceinline__ int row_unchecked(float kh) const {
        if (wtop) {   // wave-uniform: arbitrary monotone edges (emspec_set_row_edges_hz)
            int lo = 0;
            for (int w = wtop; w > 0; w >>= 1) {
                const int mid = lo + w;
                if (mid < R && eb[mid] <= kh) lo = mid;
            }
            return lo;
        }
        int r0 = (int)((__log2f(kh) - l2e0) * rscale);
        r0 = max(0, min(r0, R - 1));
        const float lo = eb[r0], hi = eb[r0 + 1];
        r0 += (kh >= hi) ? 1 : 0;
        r0 -= (kh < lo) ? 1 : 0;
        return r0;
    }
    // Row, or -1 / R when kh lies below / at-or-above the table: the +-1 correction of the clamped hint walks
    // off the table exactly when kh is out of range, so `(unsigned)row < R` replaces the two range compares.
    // (NaN compares false and would read as in range: callers gate on a finite power first.)
    __device__ __forceinline__ int row_signed(float kh) const {
        if (wtop) return in_range(kh) ? row_unchecked(kh) : -1;
        return row_unchecked(kh);
    }
    // row_signed for a table the caller KNOWS is log-spaced (wtop == 0): no branch, so the lookups of a thread's
    // consecutive bins stay in one basic block and their table reads are issued together
    __device__ __forceinline__ int row_signed_log(float kh) const {
        int r0 = (int)((__log2f(kh) - l2e0) * rscale);
        r0 = max(0, min(r0, R - 1));
        const float lo = eb[r0], hi = eb[r0 + 1];
        r0 += (kh >= hi) ? 1 : 0;
        r0 -= (kh < lo) ? 1 : 0;
        return r0;
    }
    __device__ __forceinline__ int operator()(float kh) const { return in_range(kh) ? row_unchecked(kh) : -1; }
};

// Upper power gate of stage "Power + gate" (DESIGN.md §3.4): bins above it are dropped.  1e36 keeps the reciprocal's
// argument 64 P below 2^126, where 1/d is a normal number (a bin this strong needs |x| > 1e14 on input).
constexpr float kPowerMax = 1.0e36f;
// The specialised kernels use recip_normal below; their launcher requires this much power floor (64 P = d > 2^-90)
constexpr float kFastMinFloor = 1.0e-27f;

// 1/d, correctly rounded, for 2^-96 < d < 2^126: what hipcc's IEEE division (v_div_scale x2, v_rcp, 5 fma, v_div_fmas,
// v_div_fixup: 11 instructions) computes when neither scaling nor a special case applies, without the four instructions
// that only serve those cases.  The callers guarantee the range for every bin whose result is used: the power gate
// keeps 64 P = d below 2^126 and the launcher selects these kernels only when the power floor keeps d above 2^-90.
__device__ __forceinline__ float recip_normal(float d) {
    const float r = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r, 1.0f);
    const float r1 = __builtin_fmaf(e, r, r);
    const float rem = __builtin_fmaf(-d, r1, 1.0f);
    const float q1 = __builtin_fmaf(rem, r1, r1);
    const float rem2 = __builtin_fmaf(-d, q1, 1.0f);
    return __builtin_fmaf(rem2, r1, q1);
}

struct BinOut { float power; int dcol; int row; };  // dcol relative to the frame's own column

// (c + c) - s in one instruction: c + c is exact, so fma(2, c, -s) rounds the same real number once, like the
// two-instruction form of the bit model (oracle: (c + c) - s).  Overflow of c + c aside (|c| > 1.7e38), identical.
__device__ __forceinline__ float twice_minus(float c, float s) { return __builtin_fmaf(2.0f, c, -s); }

// conjugate split, scaled by 2:  Y = Z[k] + conj Z[N-k],  T = -j (Z[k] - conj Z[N-k])
struct YT { float yr, yi, tr, ti; };
__device__ __forceinline__ YT split_yt(float2 z, float2 w) {
    YT o;
    o.yr = z.x + w.x; o.yi = z.y - w.y; o.tr = z.y + w.y; o.ti = w.x - z.x;
    return o;
}

struct ExactLookup {   // binary search over the edge table (LDS or global)
    const float* eb; int R; int wtop;
    __device__ __forceinline__ int operator()(float kh) const { return row_lookup(eb, R, wtop, kh); }
};

// Stages "Power + gate", "Reassign", "Index quantise" for bin k from the
// split spectra at k-1, k, k+1.
template <class Lookup>
__device__ __forceinline__ BinOut reassign_core(const PlanDev& pl, const Lookup& lookup, int k,
                                                const YT& m, const YT& c, const YT& p) {
    // spectral Hann identities, scaled by 8
    float Ar = (c.yr + c.yr) - (m.yr + p.yr), Ai = (c.yi + c.yi) - (m.yi + p.yi);
    float Br = (c.tr + c.tr) - (m.tr + p.tr), Bi = (c.ti + c.ti) - (m.ti + p.ti);
    float Dr = m.yr - p.yr, Di = m.yi - p.yi;
    float den = __builtin_fmaf(Ar, Ar, Ai * Ai);
    BinOut o;
    o.power = den * 0.015625f;
    o.dcol = 0;
    o.row = -1;
    if (o.power >= pl.pfloor_abs && o.power <= kPowerMax) {
        if (pl.reassign) {
            float numT = __builtin_fmaf(Br, Ar, Bi * Ai);
            float numF = __builtin_fmaf(Dr, Ar, Di * Ai);
            float inv = 1.0f / den;      // IEEE-correct divide (no fast-math)
            float ts = numT * inv;
            float ks = numF * inv;
            float cf = __builtin_floorf(__builtin_fmaf(ts, pl.tscale, 0.5f));
            if (__builtin_fabsf(cf) <= (float)pl.D) {
                o.dcol = (int)cf;
                o.row = lookup((float)k + ks);
            }
        } else {
            o.row = lookup((float)k);
        }
    }
    return o;
}

// The same stages without a branch, for a plan the caller knows to be "fast" (reassignment ON, log-spaced rows, power
// floor >= kFastMinFloor: emspec_plan_is_fast): everything is computed, the gates select at the end.  Same results as
// reassign_core, bit for bit (recip_normal == 1.0f / den on the gated range; row_signed_log walks off the table exactly
// when k-hat is out of range).
__device__ __forceinline__ BinOut reassign_core_fast(const PlanDev& pl, const HintLookup& lk, int k,
                                                     const YT& m, const YT& c, const YT& p) {
    const float Ar = twice_minus(c.yr, m.yr + p.yr), Ai = twice_minus(c.yi, m.yi + p.yi);   // (c + c) - s, rounded once either way
    const float Br = twice_minus(c.tr, m.tr + p.tr), Bi = twice_minus(c.ti, m.ti + p.ti);
    const float Dr = m.yr - p.yr, Di = m.yi - p.yi;
    const float den = __builtin_fmaf(Ar, Ar, Ai * Ai);
    BinOut o;
    o.power = den * 0.015625f;
    const float numT = __builtin_fmaf(Br, Ar, Bi * Ai);
    const float numF = __builtin_fmaf(Dr, Ar, Di * Ai);
    const float inv = recip_normal(den);
    const float cf = __builtin_floorf(__builtin_fmaf(numT * inv, pl.tscale, 0.5f));
    const bool ok = (o.power >= pl.pfloor_abs) && (o.power <= kPowerMax) && (__builtin_fabsf(cf) <= (float)pl.D);
    const int rr = lk.row_signed_log((float)k + numF * inv);
    o.dcol = ok ? (int)cf : 0;
    o.row = (ok && (unsigned)rr < (unsigned)lk.R) ? rr : -1;
    return o;
}
__device__ __host__ __forceinline__ bool emspec_plan_is_fast(const PlanDev& pl) {
    return pl.reassign && pl.log_rows && pl.pfloor_abs >= kFastMinFloor;
}

//   zm,z0,zp = Z[k-1],Z[k],Z[k+1] ;  wm,w0,wp = Z[N-k+1],Z[N-k],Z[N-k-1]
__device__ __forceinline__ BinOut reassign_bin(const PlanDev& pl, const float* eb, int wtop, int k,
                                               float2 zm, float2 z0, float2 zp,
                                               float2 wm, float2 w0, float2 wp) {
    ExactLookup lk{eb, pl.rows, wtop};
    return reassign_core(pl, lk, k, split_yt(zm, wm), split_yt(z0, w0), split_yt(zp, wp));
}

// Float accumulate into an LDS cell without ds_add_f32 (which retires about one lane
// per 1.5 cycles on gfx950, measured: profiles/README.md).  Integer LDS atomics run at
// full rate, so: exchange a NaN sentinel in (takes the cell), add in the VALU, store the
// sum back (releases it).  A lane that reads the sentinel lost the race to another lane
// or wave and retries; no lane holds a cell across a loop iteration, so the loop always
// drains.  Result: the same float sum as an atomic add, in some order.
// (diagnostic twin: also counts the loop's rounds, for the stamped build only)
__device__ __forceinline__ unsigned lds_accumulate_counted(float* cell, float p, bool want) {
    unsigned* c = reinterpret_cast<unsigned*>(cell);
    bool todo = want;
    unsigned rounds = 0;
    while (__builtin_amdgcn_ballot_w64(todo) != 0ull) {
        ++rounds;
        if (todo) {
            const unsigned old = __hip_atomic_exchange(c, 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (old != 0xFFFFFFFFu) {
                __hip_atomic_store(cell, __uint_as_float(old) + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                todo = false;
            }
        }
    }
    return rounds;
}
// The loop is written out in assembly: hipcc's code for the C++ form (lds_accumulate_ref below) spends ~25 instructions
// per round, most of them scalar exec-mask bookkeeping, and a wave issues its instructions one at a time (one per ~4.75
// cycles); this is 9 per round.  exec is narrowed to the lanes that still have to add, the winners of a round store and
// drop out, exec is restored at the end.  (N = 4096 headline kernel: 11.75 -> 10.47 ms per launch on the same box.)
__device__ __forceinline__ void lds_accumulate(float* cell, float p, bool want) {
    const unsigned addr = (unsigned)reinterpret_cast<uintptr_t>(cell);   // low word of a flat LDS address = the LDS offset
    const unsigned long long wmask = __builtin_amdgcn_ballot_w64(want);
    unsigned old;
    unsigned long long sv, tmp;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "s_and_b64 exec, exec, %[wm]\n\t"
        "s_cbranch_execz 2f\n"
        "1:\n\t"
        "ds_wrxchg_rtn_b32 %[old], %[addr], %[sent]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_cmp_ne_u32_e32 vcc, -1, %[old]\n\t"
        "v_add_f32_e32 %[old], %[old], %[p]\n\t"
        "s_mov_b64 %[tmp], exec\n\t"
        "s_mov_b64 exec, vcc\n\t"
        "ds_write_b32 %[addr], %[old]\n\t"
        "s_andn2_b64 exec, %[tmp], vcc\n\t"
        "s_cbranch_execnz 1b\n"
        "2:\n\t"
        "s_mov_b64 exec, %[sv]"
        : [old] "=&v"(old), [sv] "=&s"(sv), [tmp] "=&s"(tmp)
        : [addr] "v"(addr), [sent] "v"(0xFFFFFFFFu), [p] "v"(p), [wm] "s"(wmask)
        : "vcc", "scc", "memory");
}
// the same loop in C++ (what the assembly above does; kept as its specification)
__device__ __forceinline__ void lds_accumulate_ref(float* cell, float p, bool want) {
    unsigned* c = reinterpret_cast<unsigned*>(cell);
    bool todo = want;
    while (__builtin_amdgcn_ballot_w64(todo) != 0ull) {
        if (todo) {
            const unsigned old = __hip_atomic_exchange(c, 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (old != 0xFFFFFFFFu) {
                __hip_atomic_store(cell, __uint_as_float(old) + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                todo = false;
            }
        }
    }
}

// stage "dB + colour" for one histogram cell
__device__ __forceinline__ float cell_db(const DbMap& m, float e) {
    return 10.0f * log10f(e * m.scale + 1e-20f);
}
// in-kernel variant: v_log_f32 (<= 1 ulp of log2) — within 1e-4 dB of cell_db
__device__ __forceinline__ float cell_db_fast(const DbMap& m, float e) {
    return 3.0102999566398120f * __log2f(e * m.scale + 1e-20f);
}
__device__ __forceinline__ int cell_index(const DbMap& m, float db) {
    float v = (db - m.lo) * m.inv_range;
    v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
    if (db < m.gate) v = 0.0f;
    return (int)(v * 255.0f + 0.5f);
}

}  // namespace emspec
