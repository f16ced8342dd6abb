// kernels.hip — hand-written gfx950 kernels of the reassigned-spectrogram path.
//
// Stage list: SURVEY.md §8(a).  No reference file:line can be cited — the
// reference source is private (/root/reference/README.md:73).
//
//   frames_kernel<LOG2N>   one workgroup per frame: frame gather, packed FFT
//                          (register radix-16 passes over an in-place LDS
//                          buffer), conjugate split + spectral Hann identities,
//                          reassignment, row lookup; results go to the parity
//                          dump and/or a global-atomic histogram.  Generic in
//                          N and hop; also serves the streaming per-frame call.
//   finalize_kernel        histogram -> dB / RGBA / palette index.
//   fused kernels          see fused.hip.inc (LDS column ring, batch path).
#include "emspec_launch.h"
#include "live.hip.inc"

#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <set>
#include <utility>

#ifndef EMSPEC_BINS_UNROLL
#define EMSPEC_BINS_UNROLL 1
#endif
namespace emspec {

// Kernels that need more than 64 KB of dynamic LDS must say so once per (device, kernel).  Engines on different
// threads may launch concurrently, so the "already done" set is guarded; the HIP call itself is idempotent.
static hipError_t allow_max_lds(const void* fn) {
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> g(mu);
    if (done.count({dev, fn})) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) done.insert({dev, fn});
    return e;
}
// Device word raised by a kernel whose bounded wait timed out (the arrival-counter polls of the fused kernels: a protocol
// error - every wave arrives before it polls, so a timeout cannot happen in a correct build - after which the results
// are invalid).  A symbol of the code object, so no kernel carries a pointer for it; emspec_device_status reads it.
__device__ int g_kernel_error = 0;
int read_kernel_error(bool clear) {
    int v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_kernel_error), sizeof(int)) != hipSuccess) { (void)hipGetLastError(); return -1; }
    if (v && clear) {
        const int zero = 0;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_kernel_error), &zero, sizeof(int));
    }
    return v;
}
// compute units of the current device (grid sizing: workgroups per launch are counted in rounds of CUs)
int device_cus() {
    static std::mutex mu;
    static int cus[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> g(mu);
    int& c = cus[dev & 63];
    if (c <= 0) {
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256;
    }
    return c;
}

// ---------------------------------------------------------------------------
// FFT passes over the in-place LDS buffer.  16 points per thread, T = N/16.
// After the first radix-16 pass the transform is 16 independent sub-FFTs of N/16 <= 1024
// points, and with the index maps below every later pass of a wave touches only positions
// that the same wave wrote (wave w owns positions [1024w, 1024w+1024)), so those passes are
// ordered by wave_lds_sync() instead of a workgroup barrier.
// ---------------------------------------------------------------------------
// Middle-pass twiddles depend on the low B0 bits of the thread id only: 15 << B0 entries per
// pass (<= 1020 in total), staged once per workgroup in LDS instead of 15 global loads per
// thread per pass.
// Natural-order buffer swizzle.  In the rewrite after the last pass the 16 lanes of a write group
// hold k's that differ in bits [LOG2N-5 .. LOG2N-8] only (bit-reversed thread id), i.e. 2^(LOG2N-8)
// slots apart: a 16-way bank conflict in a plain buffer.  XOR-ing those bits into the low bits makes
// the group hit 16 distinct positions; reads by consecutive k stay conflict-free (the XOR source is
// constant, or a bijection, across 32 consecutive k).
template <int LOG2N>
__device__ __forceinline__ int natpos(int k) {
    if constexpr (LOG2N >= 12) return k ^ ((k >> (LOG2N - 8)) & 15);
    else if constexpr (LOG2N == 11) return k ^ ((k >> 4) & 7);
    else if constexpr (LOG2N == 10) return k ^ ((k >> 4) & 3);
    else return k;
}

constexpr int mid_tw_entries(int log2n) {
    int s0 = 4, n = 0;
    while (log2n - s0 > 4) { n += 15 << (log2n - s0 - 4); s0 += 4; }
    return (n + 1) & ~1;   // keep the next region 16-byte aligned
}
template <int LOG2N, int S0>
__device__ __forceinline__ void stage_mid_twiddles(float2* stw, const float2* __restrict__ tw, int t, int nthreads) {
    if constexpr (LOG2N - S0 > 4) {
        constexpr int B0 = LOG2N - S0 - 4, NE = 15 << B0;
        for (int x = t; x < NE; x += nthreads) stw[x] = tw[tw_slot_index16<LOG2N, S0>(x >> B0, x & ((1 << B0) - 1))];
        stage_mid_twiddles<LOG2N, S0 + 4>(stw + NE, tw, t, nthreads);
    }
}

// INPLACE: the last pass writes its results back where it read them (position p then holds Z[bitrev(p)]) and
// issues no barrier: the caller synchronises before other waves read the spectrum.
// PRIO: lower the wave's priority by one after each pass (progress-based priority, see fused_n16384.hip.inc).
template <int LOG2N, int S0, bool INPLACE = false, bool PRIO = false>
__device__ __forceinline__ void fft_rest(float2* sm, const float2* stw, int t, const float2* __restrict__ tw) {
    constexpr int REM = LOG2N - S0;
    if constexpr (REM > 4) {
        // middle pass: stages S0..S0+3, one group of 16 per thread, in place
        constexpr int B0 = LOG2N - S0 - 4;
        const int lo = t & ((1 << B0) - 1), hi = t >> B0;
        const int base = (hi << (B0 + 4)) + lo;
        // padi(base + (i << B0)) = padi(base) + (i << B0) + pad_i with pad_i a compile-time constant: base = hi * 2^(B0+4)
        // + lo with lo < 2^B0, so the 1-in-16 pad (p >> 4) of the sum splits exactly - the sixteen positions are ONE
        // address register and immediate offsets instead of a shift and an add each
        const int pb = padi(base);
        auto poff = [](int i) constexpr { return (i << B0) + (B0 >= 4 ? (i << (B0 >= 4 ? B0 - 4 : 0)) : (i >> (B0 < 4 ? 4 - B0 : 0))); };
        float2 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = sm[pb + poff(i)];
        fft_stages_w<4, TwLdsSym16, PRIO>(v, TwLdsSym16{stw + lo, 1 << B0});   // PRIO builds (fused N = 16384) are also the register-lean ones
        if constexpr (INPLACE && REM == 6) {
            // The network's LAST TWO stages in registers (round 5): after this pass thread (hi, lo), lo = t & 3, holds
            // positions 64 hi + lo + 4 i - the four positions of a radix-4 group of the last pass sit in the four lanes of a
            // QUAD, same register.  Partners come through DPP quad permutes (lane ^ 2, then lane ^ 1; the compiler folds them
            // into the adds), a - b is written partner + (-own) (the same IEEE result, signed zeros included), and the one
            // non-trivial twiddle, -j on the group's fourth point, is a swap and a sign.  Same butterflies as
            // fft_stages<LOG2N, S0 + 4, 2>; saves the network's last trip through LDS and its wave-local sync.
            const int l4 = t & 3;
            const unsigned s12 = (unsigned)(l4 & 2) << 30, s13 = (unsigned)(l4 & 1) << 31;
            const bool rot = l4 == 3;
            auto dpp = [](float x, auto ctrl) {
                return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, 0xF, 0xF, true));
            };
            auto neg_if = [](float x, unsigned m) { return __uint_as_float(__float_as_uint(x) ^ m); };
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float pr = dpp(v[i].x, std::integral_constant<int, 0x4E>{}), pi = dpp(v[i].y, std::integral_constant<int, 0x4E>{});
                const float dr = pr + neg_if(v[i].x, s12), di = pi + neg_if(v[i].y, s12);
                const float ar = rot ? di : dr, ai = rot ? -dr : di;
                const float qr = dpp(ar, std::integral_constant<int, 0xB1>{}), qi = dpp(ai, std::integral_constant<int, 0xB1>{});
                v[i] = make_float2(qr + neg_if(ar, s13), qi + neg_if(ai, s13));
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) sm[pb + poff(i)] = v[i];
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(3 - S0 / 4);
            return;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) sm[pb + poff(i)] = v[i];
        wave_lds_sync();
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(3 - S0 / 4);   // S0 = 4 -> 2, S0 = 8 -> 1
        fft_rest<LOG2N, S0 + 4, INPLACE, PRIO>(sm, stw + (15 << B0), t, tw);
    } else {
        // last pass: stages S0..LOG2N-1 (R = REM <= 4), G consecutive groups per thread
        // (positions 16t .. 16t+15: wave-local); output rewritten in natural frequency
        // order (unpadded) after a barrier.
        constexpr int R = REM, G = 16 >> R;
        float2 v[G][1 << R];
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
#pragma unroll
            for (int i = 0; i < (1 << R); ++i) v[gi][i] = sm[17 * t + (gi << R) + i];   // padi(16 t + m) = 17 t + m for m < 16
            fft_stages<LOG2N, S0, R>(v[gi], 0, tw);
        }
        if constexpr (INPLACE) {
#pragma unroll
            for (int gi = 0; gi < G; ++gi)
#pragma unroll
                for (int i = 0; i < (1 << R); ++i) sm[17 * t + (gi << R) + i] = v[gi][i];
            return;
        }
        __syncthreads();
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
            // bitrev(g 2^R + i) = bitrev_R(i) 2^(LOG2N-R) + bitrev_(LOG2N-R)(g), and natpos swizzles with bits below
            // LOG2N-R only: one swizzled base per group, the 2^R slots at compile-time offsets
            const int g = G * t + gi;
            const int kb = natpos<LOG2N>((int)(__brev((unsigned)g) >> (32 - (LOG2N - R))));
#pragma unroll
            for (int i = 0; i < (1 << R); ++i) {
                constexpr int SH = LOG2N - R;
                const int bi = ((i & 1) << 3 | (i & 2) << 1 | (i & 4) >> 1 | (i & 8) >> 3) >> (4 - R);   // bitrev_R(i)
                sm[kb + (bi << SH)] = v[gi][i];
            }
        }
        __syncthreads();
    }
}

// SINK: which outputs are compiled in.  0 = whatever the FrameSinks say at run time (not instantiated since round 6: the
// streaming calls it served are the live form now); 1 = the parity dump only; 2 = the records for the tile scatter only; 3 = a live
// multi-stream launch (live.hip.inc: per-stream frame counts, samples from the ring / the staging block, ring atomics,
// the stream's last workgroup finalises).  With the
// run-time form every bin re-tests five sink pointers and recomputes three 64-bit frame offsets on the scalar unit -
// a few hundred instructions of each wave's stream per frame.
// FASTC: the plan is "fast" (emspec_plan_is_fast): the branch-free per-bin core.
template <int LOG2N, int SINK, bool FASTC = false>
__global__ __launch_bounds__((1 << LOG2N) / 16) void frames_kernel(
    PlanDev pl, const float* __restrict__ pcm, int64_t L, int64_t frame0, int64_t nframes,
    FrameSinks sk) {
    constexpr int N = 1 << LOG2N, T = N / 16, K = N / 2 + 1;
    extern __shared__ float4 smem4[];
    float2* sm = reinterpret_cast<float2*>(smem4);
    float2* stw = sm + PaddedSize<N>::value;                       // middle-pass twiddles
    float* seb = reinterpret_cast<float*>(stw + mid_tw_entries(LOG2N));
    const int t = threadIdx.x;
    // Workgroups are dealt round-robin to the 8 XCDs (each with its own L2) in launch order, so neighbouring frames - which
    // share 15/16 of their samples - would land on different L2s and every L2 would fetch about half of every frame
    // (PMC: 8.3 KB read per column against 1 KB algorithmic).  Re-deal them: XCD x takes the x-th eighth of the
    // (stream, frame) sequence, a contiguous run, and the overlap is served by that XCD's L2.
    int64_t f;                             // frame within the launch
    int s;                                 // stream
    int64_t j, jcol;                       // frame within the pcm buffer; its own absolute column
    LiveBlock lb;                          // live multi-stream launch (live.hip.inc): per-stream frame counts and sample sources
    constexpr bool live = SINK == 3;       // 3 = the run-time form of a live multi-stream launch
    if (live) {
        if (!lb.init(sk.live, pl.hop, T)) return;
        s = lb.s; f = lb.f; j = f; jcol = lb.column();
    } else {
    const int64_t total = (int64_t)gridDim.x * gridDim.y, lin = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
    const int64_t full = total & ~(int64_t)7;
    const int64_t work = lin < full ? (lin & 7) * (full >> 3) + (lin >> 3) : lin;
    // (32-bit division whenever the launch has fewer than 2^32 workgroups, i.e. always in practice: a 64-bit division or
    // modulo is ~140 scalar instructions on this hardware, and every workgroup - one frame - paid two of them)
    if (total <= 0xFFFFFFFFll) {
        const unsigned w32 = (unsigned)work, q32 = w32 / gridDim.x;
        s = (int)q32;
        f = (int64_t)(w32 - q32 * gridDim.x);
    } else {
        f = work % gridDim.x;
        s = (int)(work / gridDim.x);
    }
    j = frame0 + f;
    jcol = j + sk.col_offset;
    }

    for (int r = t; r <= pl.rows; r += T) seb[r] = pl.ebin[r];
    stage_mid_twiddles<LOG2N, 4>(stw, pl.tw, t, T);

    // stage "Frame gather" + packing z = x + j*ramp*x, ramp = (n-N/2)*(2/N) exact
    const float* x = pcm + (size_t)s * L + j * pl.hop;
    float2 v[16];
    const float rs = 2.0f / (float)N;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int n = t + T * i;
        const float xv = live ? lb.sample(n) : x[n];
        v[i] = make_float2(xv, xv * ((float)(n - N / 2) * rs));
    }
    if constexpr (live) live_stamp_after(sk.live, s, (int)f, 2, v[15].x + v[0].x);   // (diagnostic build only: waits for the samples)
    // stage "STFT": first pass (stages 0..3) straight from registers
    fft_stages<LOG2N, 0, 4>(v, t, pl.tw);
#pragma unroll
    for (int i = 0; i < 16; ++i) sm[padi(t) + (T + T / 16) * i] = v[i];   // padi(t + T i) = padi(t) + (T + T/16) i: T is a multiple of 16
    __syncthreads();
    fft_rest<LOG2N, 4>(sm, stw, t, pl.tw);
    if constexpr (live) live_stamp(sk.live, s, (int)f, 3);

    // per-bin stages: k = t + T*i (i = 0..7), plus k = N/2 on thread 0
    HintLookup lk;
    lk.init(seb, pl.ebin, pl.rows, pl.log_rows);
    const size_t fidx = (size_t)s * nframes + f;                   // this frame's index in the per-frame output arrays
    float* const out_power = (SINK == 1 || (SINK == 0 && sk.power)) ? sk.power + fidx * K : nullptr;   // (SINK 3: ring only)
    int32_t* const out_col = out_power ? sk.col + fidx * K : nullptr;
    int32_t* const out_row = out_power ? sk.row + fidx * K : nullptr;
    uint2* const out_rec = (SINK == 2 || (SINK == 0 && sk.records)) ? sk.records + fidx * (K + 1) : nullptr;
    // ring slot of this frame's own column (ring sinks): a bin lands at most D < slots columns away, so its slot is an add
    // and a wrap instead of a 64-bit modulo per bin
    const int slot_j = ((SINK == 0 || SINK == 3) && sk.ring) ? (int)(jcol % sk.hist_slots) : 0;
    auto do_bin = [&](int k, int pzm, int pz0, int pzp, int pwm, int pw0, int pwp) {
        const float2 zm = sm[pzm], z0 = sm[pz0], zp = sm[pzp];
        const float2 wm = sm[pwm], w0 = sm[pw0], wp = sm[pwp];
        const BinOut o = FASTC ? reassign_core_fast(pl, lk, k, split_yt(zm, wm), split_yt(z0, w0), split_yt(zp, wp))
                               : reassign_core(pl, lk, k, split_yt(zm, wm), split_yt(z0, w0), split_yt(zp, wp));
        const int64_t col = jcol + o.dcol;
        if (SINK == 1 || (SINK == 0 && out_power)) {
            out_power[k] = o.power;
            out_col[k] = (int32_t)col;
            out_row[k] = o.row;
        }
        if (SINK == 2 || (SINK == 0 && out_rec)) {
            // frame stride is K+1 (even): 16-byte aligned chunks; the pad record is marked dropped
            const unsigned key = o.row >= 0 ? ((unsigned)(o.dcol + 32768) << 16) | (unsigned)o.row : 0xFFFFFFFFu;
            out_rec[k] = make_uint2(__float_as_uint(o.power), key);
            if (k == N / 2) out_rec[k + 1] = make_uint2(0u, 0xFFFFFFFFu);
        }
        if constexpr (SINK == 0 || SINK == 3) {
            if (sk.hist && o.row >= 0 && col >= 0 && col < sk.total_cols) {
                const int64_t slot = sk.ring ? (int64_t)live_slot(slot_j, o.dcol, (int)sk.hist_slots) : col;
                atomicAdd(sk.hist + ((size_t)s * sk.hist_slots + slot) * pl.rows + o.row, o.power);
            }
        }
    };
    auto do_bin_general = [&](int k) {
        do_bin(k, natpos<LOG2N>((k - 1) & (N - 1)), natpos<LOG2N>(k), natpos<LOG2N>(k + 1), natpos<LOG2N>((N - k + 1) & (N - 1)),
               natpos<LOG2N>((N - k) & (N - 1)), natpos<LOG2N>(N - k - 1));
    };
    do_bin_general(t);                                             // i = 0: the neighbours of bin 0 wrap round
    {
        // i = 1 .. 7: k = t + T i.  The swizzle of natpos reads bits below log2(T) only, so natpos(a + T m) = natpos(a)
        // + T m for a < T: each of the six spectrum positions is a per-thread base plus (or minus) T i - six adds per bin
        // instead of six swizzles (the mirrored positions count down).
        const int bzm = natpos<LOG2N>((t - 1) & (T - 1)) - (t == 0 ? T : 0);
        const int bz0 = natpos<LOG2N>(t);
        const int bzp = natpos<LOG2N>((t + 1) & (T - 1)) + (t == T - 1 ? T : 0);
        const int bwm = natpos<LOG2N>((1 - t) & (T - 1)) + T * (16 - (t >= 2 ? 1 : 0));
        const int bw0 = natpos<LOG2N>((0 - t) & (T - 1)) + T * (16 - (t >= 1 ? 1 : 0));
        const int bwp = natpos<LOG2N>(T - 1 - t) + T * 15;
        // (the live form is latency-bound - one wave per SIMD - and overlaps the bins' LDS round trips instead)
        constexpr int BU = SINK == 3 ? 7 : EMSPEC_BINS_UNROLL;
#pragma unroll BU
        for (int i = 1; i < 8; ++i)
            do_bin(t + T * i, bzm + T * i, bz0 + T * i, bzp + T * i, bwm - T * i, bw0 - T * i, bwp - T * i);
    }
    if (t == 0) do_bin_general(N / 2);                             // the Nyquist bin
    if constexpr (SINK == 3) {
        if (live_last_arrival(sk.live, s, lb.d.frames, sm))
            live_finalize<float>(sk.live, lb.d, s, sk.hist, (int)sk.hist_slots, pl.rows, pl.D, T, LiveConvF32{sk.fin_map});
        live_stamp(sk.live, s, (int)f, 6);
        return;
    }
}

bool supported_fft(int n) { return n == 256 || n == 512 || n == 1024 || n == 2048 || n == 4096 || n == 8192 || n == 16384; }

template <int LOG2N>
static hipError_t launch_frames_t(const PlanDev& pl, const float* pcm, int64_t L, int S, int64_t frame0,
                                  int64_t nframes, const FrameSinks& sk, hipStream_t st) {
    constexpr int N = 1 << LOG2N;
    const size_t lds = (size_t)(PaddedSize<N>::value + mid_tw_entries(LOG2N)) * sizeof(float2) + (size_t)(pl.rows + 1) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    // the sink combination picks the build: dump only, records only, or the run-time form
    const bool plain = !sk.hist;
    const int sink = sk.live.streams ? 3 : (plain && sk.power && !sk.records) ? 1 : ((plain && sk.records && !sk.power) ? 2 : 0);
    if (sink == 0) return hipErrorInvalidValue;   // (no caller mixes the sinks; the run-time form is not instantiated any more)
    const bool fastc = emspec_plan_is_fast(pl);
    const void* fn = sink == 1 ? (fastc ? reinterpret_cast<const void*>(&frames_kernel<LOG2N, 1, true>) : reinterpret_cast<const void*>(&frames_kernel<LOG2N, 1>))
                   : sink == 2 ? (fastc ? reinterpret_cast<const void*>(&frames_kernel<LOG2N, 2, true>) : reinterpret_cast<const void*>(&frames_kernel<LOG2N, 2>))
                               : (fastc ? reinterpret_cast<const void*>(&frames_kernel<LOG2N, 3, true>) : reinterpret_cast<const void*>(&frames_kernel<LOG2N, 3>));
    if (lds > 64 * 1024) {
        const hipError_t e = allow_max_lds(fn);
        if (e != hipSuccess) return e;
    }
    // grid.x is limited to 2^31-1, grid.y to 65535
    if (nframes <= 0 || S <= 0) return hipSuccess;
    if (S > 65535 || nframes > 0x7fffffffLL) return hipErrorInvalidValue;
    dim3 grid((unsigned)nframes, (unsigned)S), block(N / 16);
    if (sink == 1 && fastc) hipLaunchKernelGGL((frames_kernel<LOG2N, 1, true>), grid, block, lds, st, pl, pcm, L, frame0, nframes, sk);
    else if (sink == 1) hipLaunchKernelGGL((frames_kernel<LOG2N, 1>), grid, block, lds, st, pl, pcm, L, frame0, nframes, sk);
    else if (sink == 2 && fastc) hipLaunchKernelGGL((frames_kernel<LOG2N, 2, true>), grid, block, lds, st, pl, pcm, L, frame0, nframes, sk);
    else if (sink == 2) hipLaunchKernelGGL((frames_kernel<LOG2N, 2>), grid, block, lds, st, pl, pcm, L, frame0, nframes, sk);
    else if (sink == 3 && fastc) hipLaunchKernelGGL((frames_kernel<LOG2N, 3, true>), grid, block, lds, st, pl, pcm, L, frame0, nframes, sk);
    else hipLaunchKernelGGL((frames_kernel<LOG2N, 3>), grid, block, lds, st, pl, pcm, L, frame0, nframes, sk);
    return hipGetLastError();
}

hipError_t launch_frames(int n, const PlanDev& pl, const float* pcm, int64_t L, int S, int64_t frame0,
                         int64_t nframes, const FrameSinks& sk, hipStream_t st) {
    switch (n) {
        case 256: return launch_frames_t<8>(pl, pcm, L, S, frame0, nframes, sk, st);
        case 512: return launch_frames_t<9>(pl, pcm, L, S, frame0, nframes, sk, st);
        case 1024: return launch_frames_t<10>(pl, pcm, L, S, frame0, nframes, sk, st);
        case 2048: return launch_frames_t<11>(pl, pcm, L, S, frame0, nframes, sk, st);
        case 4096: return launch_frames_t<12>(pl, pcm, L, S, frame0, nframes, sk, st);
        case 8192: return launch_frames_t<13>(pl, pcm, L, S, frame0, nframes, sk, st);
        case 16384: return launch_frames_t<14>(pl, pcm, L, S, frame0, nframes, sk, st);
        default: return hipErrorInvalidValue;
    }
}

// ---------------------------------------------------------------------------
// tile scatter: the "Scatter" and "dB + colour" stages for shapes without a fused kernel.
// Workgroup (tile, stream) owns columns [c0, c0+TILE): it streams the records of every frame that
// can reach them through an LDS histogram (exchange-based float accumulate, see lds_accumulate),
// then writes the finished columns.  Records are read (TILE+2D)/TILE times in total.
// ---------------------------------------------------------------------------
template <int CH>
__global__ __launch_bounds__(1024) void tile_scatter_kernel(const uint2* __restrict__ rec, int K, int R, int D, int tile,
                                                            DbMap dm, const uint32_t* __restrict__ lut, int64_t C,
                                                            float* __restrict__ db, uint32_t* __restrict__ rgba,
                                                            uint8_t* __restrict__ index) {
    extern __shared__ float4 smem4[];
    float* hist = reinterpret_cast<float*>(smem4);              // [tile][R]
    uint32_t* slut = reinterpret_cast<uint32_t*>(hist + (size_t)tile * R);
    const int tid = threadIdx.x, s = blockIdx.y;
    const int64_t c0 = (int64_t)blockIdx.x * tile;
    const int64_t c1 = (c0 + tile < C) ? c0 + tile : C;
    for (int q = tid; q < tile * R / 4; q += 1024) reinterpret_cast<float4*>(hist)[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < 256) slut[tid] = lut[tid];
    __syncthreads();
    const int64_t jlo = c0 - D > 0 ? c0 - D : 0, jhi = (c1 + D < C) ? c1 + D : C;
    const int span = (int)(c1 - c0);
    // A job is CH consecutive records of one frame (one thread, 8*CH contiguous bytes).  Lanes of a
    // wave therefore work CH bins apart, and a thread pre-sums runs of equal (column,row) keys in
    // registers, so a wide high-frequency row costs one LDS accumulate per thread, not one per bin.
    const int Kp = K + 1;                           // padded frame stride (even)
    const int nch = (Kp + CH - 1) / CH;             // chunks per frame
    const int64_t njobs = (jhi - jlo) * nch;
    const uint2* base = rec + (size_t)s * C * Kp;
    for (int64_t job = tid; job < njobs; job += 1024) {
        const int64_t j = jlo + job / nch;
        const int k0 = (int)(job % nch) * CH;
        const uint4* src = reinterpret_cast<const uint4*>(base + (size_t)j * Kp + k0);
        uint4 q[CH / 2];
#pragma unroll
        for (int u = 0; u < CH / 2; ++u) q[u] = (k0 + 2 * u < Kp) ? src[u] : make_uint4(0u, 0xFFFFFFFFu, 0u, 0xFFFFFFFFu);
        const int jrel = (int)(j - c0);
        unsigned cur = 0xFFFFFFFFu;
        float acc = 0.0f;
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const unsigned pbits = (u & 1) ? q[u >> 1].z : q[u >> 1].x;
            const unsigned key = (u & 1) ? q[u >> 1].w : q[u >> 1].y;
            const bool flush = key != cur;
            // flush the finished run (if it was a live one inside this tile)
            {
                const int colrel = jrel + (int)(cur >> 16) - 32768;
                const bool ok = flush && (cur != 0xFFFFFFFFu) && ((unsigned)colrel < (unsigned)span);
                lds_accumulate(hist + (ok ? colrel * R + (int)(cur & 0xFFFFu) : 0), acc, ok);
            }
            acc = flush ? __uint_as_float(pbits) : acc + __uint_as_float(pbits);
            cur = key;
        }
        {
            const int colrel = jrel + (int)(cur >> 16) - 32768;
            const bool ok = (cur != 0xFFFFFFFFu) && ((unsigned)colrel < (unsigned)span);
            lds_accumulate(hist + (ok ? colrel * R + (int)(cur & 0xFFFFu) : 0), acc, ok);
        }
    }
    __syncthreads();
    const int ncell4 = span * R / 4;
    for (int q = tid; q < ncell4; q += 1024) {
        const float4 e = reinterpret_cast<const float4*>(hist)[q];
        const float d0 = cell_db(dm, e.x), d1 = cell_db(dm, e.y), d2 = cell_db(dm, e.z), d3 = cell_db(dm, e.w);
        const int i0 = cell_index(dm, d0), i1 = cell_index(dm, d1), i2 = cell_index(dm, d2), i3 = cell_index(dm, d3);
        const size_t o = ((size_t)s * C + c0) * R + (size_t)q * 4;
        if (db) *reinterpret_cast<float4*>(db + o) = make_float4(d0, d1, d2, d3);
        if (rgba) *reinterpret_cast<uint4*>(rgba + o) = make_uint4(slut[i0], slut[i1], slut[i2], slut[i3]);
        if (index) *reinterpret_cast<uint32_t*>(index + o) =
            (uint32_t)i0 | ((uint32_t)i1 << 8) | ((uint32_t)i2 << 16) | ((uint32_t)i3 << 24);
    }
}

template <int CH>
static hipError_t launch_tile_scatter_t(const uint2* records, int n, const PlanDev& pl, const DbMap& m, const uint8_t* lut,
                                        int S, int64_t C, float* db, uint8_t* rgba, uint8_t* index, hipStream_t st,
                                        int tile, size_t lds) {
    {
        const hipError_t e = allow_max_lds(reinterpret_cast<const void*>(&tile_scatter_kernel<CH>));
        if (e != hipSuccess) return e;
    }
    const int64_t ntiles = (C + tile - 1) / tile;
    hipLaunchKernelGGL(tile_scatter_kernel<CH>, dim3((unsigned)ntiles, (unsigned)S), dim3(1024), lds, st, records,
                       n / 2 + 1, pl.rows, pl.D, tile, m, reinterpret_cast<const uint32_t*>(lut), C, db,
                       reinterpret_cast<uint32_t*>(rgba), index);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// walk scatter: same job as the tile scatter, but a workgroup WALKS a segment of one stream with a
// (2D+F)-slot LDS ring, F frames per step, so every record is read once (plus a 2D-frame halo per
// segment) instead of (32+2D)/32 times.  Used whenever the ring fits in LDS.
// ---------------------------------------------------------------------------
template <int CH>
__global__ __launch_bounds__(1024) void walk_scatter_kernel(const uint2* __restrict__ rec, int K, int R, int D, int F,
                                                            int seglen, DbMap dm, const uint32_t* __restrict__ lut,
                                                            int64_t C, float* __restrict__ db, uint32_t* __restrict__ rgba,
                                                            uint8_t* __restrict__ index) {
    extern __shared__ float4 smem4[];
    const int slots = 2 * D + F;
    float* ring = reinterpret_cast<float*>(smem4);              // [slots][R]
    uint32_t* slut = reinterpret_cast<uint32_t*>(ring + (size_t)slots * R);
    const int tid = threadIdx.x, s = blockIdx.y;
    const int64_t c0 = (int64_t)blockIdx.x * seglen;
    if (c0 >= C) return;
    const int64_t c1 = (c0 + seglen < C) ? c0 + seglen : C;
    for (int q = tid; q < slots * R / 4; q += 1024) reinterpret_cast<float4*>(ring)[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < 256) slut[tid] = lut[tid];
    __syncthreads();
    const int Kp = K + 1, nch = (Kp + CH - 1) / CH;
    const uint2* base = rec + (size_t)s * C * Kp;
    const int64_t jfirst = c0 - D;                      // frames jfirst .. c1-1+D (clipped to [0,C) when reading)
    const int64_t jstop = c1 + D;
    const int span = (int)(c1 - c0);
    for (int64_t j0 = jfirst; j0 < jstop; j0 += F) {
        // ---- scatter frames j0 .. j0+F-1: job = (frame, chunk of CH consecutive records)
        for (int job = tid; job < F * nch; job += 1024) {
            const int64_t j = j0 + job / nch;
            if (j < 0 || j >= C) continue;
            const int k0 = (job % nch) * CH;
            const uint4* src = reinterpret_cast<const uint4*>(base + (size_t)j * Kp + k0);
            uint4 q[CH / 2];
#pragma unroll
            for (int u = 0; u < CH / 2; ++u) q[u] = (k0 + 2 * u < Kp) ? src[u] : make_uint4(0u, 0xFFFFFFFFu, 0u, 0xFFFFFFFFu);
            const int jrel = (int)(j - c0);
            unsigned cur = 0xFFFFFFFFu;
            float acc = 0.0f;
            auto flush = [&](bool want) {
                const int colrel = jrel + (int)(cur >> 16) - 32768;
                const bool ok = want && (cur != 0xFFFFFFFFu) && ((unsigned)colrel < (unsigned)span);
                // ring slot of absolute column c0+colrel: (c0 + colrel - jfirst) mod slots, always >= 0
                int sl = ok ? (colrel + D) % slots : 0;
                lds_accumulate(ring + sl * R + (ok ? (int)(cur & 0xFFFFu) : 0), acc, ok);
            };
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const unsigned pbits = (u & 1) ? q[u >> 1].z : q[u >> 1].x;
                const unsigned key = (u & 1) ? q[u >> 1].w : q[u >> 1].y;
                const bool newrun = key != cur;
                flush(newrun);
                acc = newrun ? __uint_as_float(pbits) : acc + __uint_as_float(pbits);
                cur = key;
            }
            flush(true);
        }
        __syncthreads();
        // ---- columns j0-D .. j0+F-1-D are complete: finalise, store, clear
        for (int q = tid; q < F * (R / 4); q += 1024) {
            const int cc = q / (R / 4), cell = (q - cc * (R / 4)) << 2;
            const int64_t col = j0 - D + cc;
            if (col >= c0 && col < c1) {
                const int sl = (int)((col - jfirst) % slots);
                float4* cp = reinterpret_cast<float4*>(ring + sl * R + cell);
                const float4 e = *cp;
                *cp = make_float4(0.f, 0.f, 0.f, 0.f);
                const float d0 = cell_db(dm, e.x), d1 = cell_db(dm, e.y), d2 = cell_db(dm, e.z), d3 = cell_db(dm, e.w);
                const int i0 = cell_index(dm, d0), i1 = cell_index(dm, d1), i2 = cell_index(dm, d2), i3 = cell_index(dm, d3);
                const size_t o = ((size_t)s * C + col) * R + cell;
                if (db) *reinterpret_cast<float4*>(db + o) = make_float4(d0, d1, d2, d3);
                if (rgba) *reinterpret_cast<uint4*>(rgba + o) = make_uint4(slut[i0], slut[i1], slut[i2], slut[i3]);
                if (index) *reinterpret_cast<uint32_t*>(index + o) =
                    (uint32_t)i0 | ((uint32_t)i1 << 8) | ((uint32_t)i2 << 16) | ((uint32_t)i3 << 24);
            }
        }
        __syncthreads();
    }
}

template <int CH>
static hipError_t launch_walk_scatter_t(const uint2* records, int n, const PlanDev& pl, const DbMap& m, const uint8_t* lut,
                                        int S, int64_t C, float* db, uint8_t* rgba, uint8_t* index, hipStream_t st, int F,
                                        size_t lds) {
    {
        const hipError_t e = allow_max_lds(reinterpret_cast<const void*>(&walk_scatter_kernel<CH>));
        if (e != hipSuccess) return e;
    }
    int64_t seg = (S * C + 4 * device_cus() - 1) / (4 * device_cus());   // >= 4 workgroups per CU when there is enough work
    seg = seg < 128 ? 128 : (seg > 1024 ? 1024 : seg);
    seg = (seg + F - 1) / F * F;
    const int64_t nseg = (C + seg - 1) / seg;
    hipLaunchKernelGGL(walk_scatter_kernel<CH>, dim3((unsigned)nseg, (unsigned)S), dim3(1024), lds, st, records, n / 2 + 1,
                       pl.rows, pl.D, F, (int)seg, m, reinterpret_cast<const uint32_t*>(lut), C, db,
                       reinterpret_cast<uint32_t*>(rgba), index);
    return hipGetLastError();
}

hipError_t launch_tile_scatter(const uint2* records, int n, const PlanDev& pl, const DbMap& m, const uint8_t* lut, int S,
                               int64_t C, float* db, uint8_t* rgba, uint8_t* index, hipStream_t st) {
    if (S <= 0 || C <= 0) return hipSuccess;
    if (S > 65535) return hipErrorInvalidValue;
    {   // walking ring when it fits: F frames per step so that F * chunks-per-frame covers the 1024 threads
        const int ch = n >= 8192 ? 32 : (n >= 2048 ? 8 : 4);
        const int nch = (n / 2 + 2 + ch - 1) / ch;
        int F = (1024 + nch - 1) / nch;
        F = F < 1 ? 1 : (F > 8 ? 8 : F);
        const size_t wl = (size_t)(2 * pl.D + F) * pl.rows * 4 + 1024;
#ifdef EMSPEC_DIAG
        static int use_walk = -1;
        if (use_walk < 0) { const char* ev = getenv("EMSPEC_NO_WALK"); use_walk = (ev && ev[0] == '1') ? 0 : 1; }   // A/B aid
#else
        constexpr bool use_walk = true;
#endif
        // measured: the walk wins when the tiles would re-read every record >= 2x (D >= 16: N=16384/512
        // 1.04e7 vs 0.94e7 col/s); for small D the tiles' independent workgroups win (N=1024: 1.8e8 vs 1.6e8)
        if (use_walk && pl.D >= 16 && wl <= 156 * 1024) {
            if (n >= 8192) return launch_walk_scatter_t<32>(records, n, pl, m, lut, S, C, db, rgba, index, st, F, wl);
            if (n >= 2048) return launch_walk_scatter_t<8>(records, n, pl, m, lut, S, C, db, rgba, index, st, F, wl);
            return launch_walk_scatter_t<4>(records, n, pl, m, lut, S, C, db, rgba, index, st, F, wl);
        }
    }
    int tile = (int)((150 * 1024) / ((size_t)pl.rows * 4));
    tile = tile > 32 ? 32 : tile;
    if (tile < 1) return hipErrorInvalidValue;
    const size_t lds = (size_t)tile * pl.rows * 4 + 1024;
    // chunk = consecutive bins per thread: wide enough that adjacent lanes rarely share a row
    if (n >= 8192) return launch_tile_scatter_t<32>(records, n, pl, m, lut, S, C, db, rgba, index, st, tile, lds);
    if (n >= 2048) return launch_tile_scatter_t<8>(records, n, pl, m, lut, S, C, db, rgba, index, st, tile, lds);
    return launch_tile_scatter_t<4>(records, n, pl, m, lut, S, C, db, rgba, index, st, tile, lds);
}

#ifdef EMSPEC_DIAG
// diagnostic: both row-lookup implementations on arbitrary inputs (tests only)
__global__ void row_lookup_probe_kernel(const float* __restrict__ ebin, int rows, const float* __restrict__ kh,
                                        int64_t count, int32_t* __restrict__ out_hint, int32_t* __restrict__ out_exact) {
    extern __shared__ float4 smem4[];
    float* seb = reinterpret_cast<float*>(smem4);
    for (int r = threadIdx.x; r <= rows; r += blockDim.x) seb[r] = ebin[r];
    __syncthreads();
    HintLookup lk;
    lk.init(seb, ebin, rows, 1);
    int wtop = 1;
    while (wtop * 2 < rows) wtop *= 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        out_hint[i] = lk(kh[i]);
        out_exact[i] = row_lookup(seb, rows, wtop, kh[i]);
    }
}
hipError_t launch_row_lookup_probe(const float* ebin, int rows, const float* kh, int64_t count, int32_t* out_hint,
                                   int32_t* out_exact, hipStream_t st) {
    hipLaunchKernelGGL(row_lookup_probe_kernel, dim3(256), dim3(256), (size_t)(rows + 4) * 4, st, ebin, rows, kh, count,
                       out_hint, out_exact);
    return hipGetLastError();
}

// Diagnostic: the short reciprocal beside the IEEE division (tests/test_gpu_parity.py sweeps whole binades)
__global__ void recip_probe_kernel(const float* __restrict__ d, int64_t count, float* __restrict__ out_short,
                                   float* __restrict__ out_ieee) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        out_short[i] = recip_normal(d[i]);
        out_ieee[i] = 1.0f / d[i];
    }
}
hipError_t launch_recip_probe(const float* d, int64_t count, float* out_short, float* out_ieee, hipStream_t st) {
    hipLaunchKernelGGL(recip_probe_kernel, dim3(1024), dim3(256), 0, st, d, count, out_short, out_ieee);
    return hipGetLastError();
}

// Diagnostic: hold `groups` workgroups of 256 threads on the device for about `usec` microseconds
// (constant 100 MHz wall clock; the iteration cap bounds the spin whatever the clock does).  Stands in
// for a communication kernel that occupies compute units beside the column kernels.
__global__ void occupy_kernel(long long ticks, unsigned* sink) {
    const long long t0 = wall_clock64();
    unsigned n = 0;
    for (long long it = 0; it < 400000000LL; ++it) {
        if (wall_clock64() - t0 >= ticks) break;
        __builtin_amdgcn_s_sleep(32);
        ++n;
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) *sink = n;
}
hipError_t launch_occupy(int groups, int usec, unsigned* sink, hipStream_t st) {
    hipLaunchKernelGGL(occupy_kernel, dim3(groups), dim3(256), 0, st, (long long)usec * 100, sink);
    return hipGetLastError();
}
#endif  // EMSPEC_DIAG

}  // namespace emspec

#include "fused.hip.inc"
#include "post.hip.inc"
#include "pack.hip.inc"
#include "exact.hip.inc"
#include "exact_fused.hip.inc"
#include "exact_fused_lr.hip.inc"
#include "live_launch.hip.inc"
