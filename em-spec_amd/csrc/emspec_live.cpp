// emspec_live.cpp — the streaming calls behind the C ABI (include/emspec.h): the live multi-stream session (emspec_columns,
// emspec_columns_flush, emspec_push_samples_multi, emspec_push_columns_multi, emspec_reset_stream, emspec_live_streams) and,
// since round 6 on the same machinery with one stream, the renderer's own calls: emspec_column (= computeSpectrogramColumn),
// emspec_column_flush, emspec_push_samples, emspec_push_columns.  Two independent sessions per engine: e->live and e->one.
//
// What it serves: BASELINE.json configs[2] is "64 concurrent 48 kHz streams" and north_star's renderer call is per frame
// (computeSpectrogramColumn(audioFrame, fftSize, hop, reassign)); /root/reference/README.md:36 ("automatically start
// visualizing your system audio") is the live case.  With one engine per stream that is S launches + S synchronisations
// per hop on the host thread; here the S streams of one engine advance together: ONE kernel launch and ONE stream
// synchronisation per call, samples read by the kernel from page-locked host memory, finished columns written by the kernel
// into page-locked host memory (the caller's own buffers when they come from emspec_host_alloc) - no copy engine involved.
// No reference file:line exists (the reference source is private, README.md:73); SURVEY.md §8(f) row 4 is the streaming glue.
// Device side: live.hip.inc / live_launch.hip.inc.
#include "emspec_engine.h"
#ifdef EMSPEC_DIAG
#pragma GCC visibility push(default)
#include "../../include/emspec_debug.h"
#pragma GCC visibility pop
#endif

#include <algorithm>
#include <cstring>

using namespace emspec;

namespace {
// per-sample-block form: frames per stream and launch, at most - about 2,048 workgroups per launch, 8..64 per stream
int live_frames_per_launch(int S) { return std::max(8, std::min(64, 2048 / std::max(1, S))); }
constexpr int kInlineFinalize = 2;   // a launch that completes more columns per stream than this finalises them in a second kernel

int64_t frames_after(int64_t total, int n, int hop) { return total >= n ? (total - n) / hop + 1 : 0; }

int pinned_grow(emspec_engine* e, void** p, size_t* have, size_t want) {
    if (*have >= want) return EMSPEC_OK;
    if (*p) { (void)hipHostFree(*p); *p = nullptr; *have = 0; }
    HIPCHK(e, hipHostMalloc(p, want, hipHostMallocDefault));
    *have = want;
    return EMSPEC_OK;
}

// the address the device uses for page-locked host memory, or null when p is not such memory
void* device_view(const void* p) {
    if (!p || !host_pinned(p)) return nullptr;
    void* d = nullptr;
    if (hipHostGetDevicePointer(&d, const_cast<void*>(p), 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return d;
}

void live_forget(LiveState& lv) {
    lv.S = 0; lv.n = 0; lv.hop = 0; lv.reassign = -1; lv.D = 0; lv.form = 0; lv.slots = 0; lv.mmax = 0; lv.cap = 0; lv.ring_mask = 0;
    lv.fed.clear(); lv.emitted.clear(); lv.seen.clear(); lv.newbase.clear(); lv.pend.clear();
}

// first call of a session: every allocation, then the state
int live_open(emspec_engine* e, LiveState& lv, int S, int n, int hop, int reassign, int form) {
    const int R = e->cfg.rows;
    const int D = latency(n, hop, reassign);
    // (a staging block of at most 2^17 samples per stream: at a large hop fewer frames per launch instead of megabytes pinned)
    const int mmax = form == 1 ? 1 : std::max(1, std::min(live_frames_per_launch(S), (1 << 17) / hop));
    const int slots = 2 * D + mmax;
    const int64_t cap = form == 1 ? n : (int64_t)mmax * hop;
    int ring = 1;
    while (ring < n + cap) ring <<= 1;
    const size_t cellb = e->exact() ? 8 : 4;
    int rc;
    if ((rc = grow(e, &lv.d_cells, &lv.cells_bytes, (size_t)S * slots * R * cellb))) return rc;
    if (form == 2 && (rc = grow(e, (void**)&lv.d_sring, &lv.sring_bytes, (size_t)S * ring * 4))) return rc;
    if ((rc = grow(e, (void**)&lv.d_done, &lv.done_bytes, (size_t)S * 4))) return rc;
    if ((rc = pinned_grow(e, &lv.h_desc, &lv.desc_bytes, (size_t)S * sizeof(LiveStream)))) return rc;
    if (form == 2 && (rc = pinned_grow(e, (void**)&lv.h_fresh, &lv.fresh_bytes, (size_t)S * cap * 4))) return rc;
    HIPCHK(e, hipMemsetAsync(lv.d_cells, 0, (size_t)S * slots * R * cellb, e->stream));
    HIPCHK(e, hipMemsetAsync(lv.d_done, 0, (size_t)S * 4, e->stream));
    if (lv.d_pstate) HIPCHK(e, hipMemsetAsync(lv.d_pstate, 0, lv.pstate_bytes, e->stream));
    lv.S = S; lv.n = n; lv.hop = hop; lv.reassign = reassign; lv.D = D; lv.form = form; lv.slots = slots; lv.mmax = mmax;
    lv.cap = cap; lv.ring_mask = ring - 1;
    lv.fed.assign(S, 0); lv.emitted.assign(S, 0); lv.seen.assign(S, 0); lv.newbase.assign(S, 0); lv.pend.assign(S, 0);
    return EMSPEC_OK;
}

int live_check(emspec_engine* e, const LiveState& lv, int S, int n, int hop, int reassign, int rows, int form) {
    int rc = check_shape(e, n, hop);
    if (rc) return rc;
    if (rows != e->cfg.rows) return fail(e, EMSPEC_ERR_INVALID_ARG, "rows does not match the engine configuration");
    if (S < 1 || S > 65535) return fail(e, EMSPEC_ERR_INVALID_ARG, "streams must be in 1..65535");
    if (lv.form != 0 && (S != lv.S || n != lv.n || hop != lv.hop || reassign != lv.reassign || form != lv.form))
        return fail(e, EMSPEC_ERR_STATE, "streams / fft size / hop / reassign / feeding mode changed mid-stream; call emspec_reset() first");
    // A flush emits columns that later frames would still have added to: the stream is at its end.  Feeding it again would emit
    // those columns a second time, holding only the new frames' energy.
    for (int s = 0; s < lv.S; ++s)
        if (lv.emitted[s] > std::max<int64_t>(lv.fed[s] - lv.D, 0))
            return fail(e, EMSPEC_ERR_STATE, "stream " + std::to_string(s) + " was flushed: reset it (emspec_reset / emspec_reset_stream) before feeding it again");
    return EMSPEC_OK;
}

// the display post-process needs the raw columns on the device and its per-stream state
int live_post_buffers(emspec_engine* e, LiveState& lv) {
    const int R = e->cfg.rows;
    int rc;
    if ((rc = grow(e, (void**)&lv.d_raw, &lv.raw_bytes, (size_t)lv.S * lv.mmax * R * 4))) return rc;
    const size_t want = (size_t)lv.S * (R + 4) * 4;
    if (lv.pstate_bytes < want) {
        if ((rc = grow(e, (void**)&lv.d_pstate, &lv.pstate_bytes, want))) return rc;
        HIPCHK(e, hipMemsetAsync(lv.d_pstate, 0, lv.pstate_bytes, e->stream));
    }
    return EMSPEC_OK;
}

// One launch of the session: the frame kernel (or, flush = true, the flush kernel) and, with the display post-process on,
// the post kernel behind it.  dst_db / dst_rgba: device-visible destinations laid out [S][out_cols][rows].
// raw_cols: an upper bound of the output slots any stream uses (out_at + the columns it emits): the stride of the raw block - the
// caller's max_columns may be far larger than what a call completes.
int live_launch(emspec_engine* e, LiveState& lv, const float* fresh, int64_t fresh_stride, int mlaunch, bool flush, float* dst_db,
                uint8_t* dst_rgba, int out_cols, int raw_cols, bool empty_col) {
    Plan* p;
    int rc;
    if ((rc = get_plan(e, lv.n, &p))) return rc;
    const bool post = e->smoothing > 0.0f || e->agc > 0.0f;
    if (post && (rc = live_post_buffers(e, lv))) return rc;
    LiveSinks ls;
    ls.streams = reinterpret_cast<const LiveStream*>(lv.h_desc);
    ls.fresh = fresh;
    ls.fresh_stride = fresh_stride;
    ls.sring = lv.form == 2 ? lv.d_sring : nullptr;
    ls.ring_mask = lv.ring_mask;
    ls.done = lv.d_done;
    ls.out_cols = out_cols;
    ls.empty_col = empty_col ? 1 : 0;
    ls.lut = reinterpret_cast<const uint32_t*>(e->d_lut);
    {   // every stream in the same state: the descriptor goes into the kernel arguments
        const LiveStream* dsc = reinterpret_cast<const LiveStream*>(lv.h_desc);
        bool same = true;
        for (int s = 1; s < lv.S && same; ++s) same = std::memcmp(&dsc[s], &dsc[0], sizeof(LiveStream)) == 0;
        ls.uniform = same ? 1 : 0;
        ls.uni = dsc[0];
    }
#ifdef EMSPEC_DIAG
    ls.stamps = lv.stamps;
#endif
    // with the post-process the frame kernel's columns are raw dB on the device, laid out like the destination
    if (post) {
        raw_cols = std::max(1, std::min(raw_cols, out_cols));
        if ((size_t)lv.S * raw_cols * e->cfg.rows * 4 > lv.raw_bytes &&
            (rc = grow(e, (void**)&lv.d_raw, &lv.raw_bytes, (size_t)lv.S * raw_cols * e->cfg.rows * 4))) return rc;
        ls.out_db = lv.d_raw;
        ls.out_rgba = nullptr;
        ls.out_cols = raw_cols;
    } else {
        ls.out_db = dst_db;
        ls.out_rgba = reinterpret_cast<uint32_t*>(dst_rgba);
    }
    const DbMap m = db_map(e, lv.n);
    const bool exact = e->exact();
    const ExactPlanDev xpd = exact ? exact_plan_dev(e, *p, lv.hop, lv.reassign) : ExactPlanDev{};
    const ExactDbMap xm = exact ? exact_db_map(e, lv.n, xpd) : ExactDbMap{};
    // many columns per stream: the frame kernel only scatters and a second kernel finalises them, one workgroup per column
    // (inline they are one workgroup's serial round trips to the memory side, ~2 us per column)
    // ... and so does a small transform: its workgroup has n / 16 (EXACT: n / 8) threads, and 1024 rows through 16 threads
    // are 8 serial batches of round trips (emspec_column at N = 256: 32.6 us per call against 25.7 at N = 4096)
    const bool defer = !flush && (mlaunch > kInlineFinalize || lv.n < 2048);
    ls.defer_finalize = defer ? 1 : 0;
    if (flush) {
        HIPCHK(e, launch_live_flush(exact, ls, lv.d_cells, lv.slots, e->cfg.rows, lv.D, m, xm, lv.S, 1, e->stream));
    } else if (exact) {
        ExactSinks xs;
        xs.hist = reinterpret_cast<unsigned long long*>(lv.d_cells);
        xs.hist_slots = lv.slots; xs.total_cols = INT64_MAX; xs.ring = 1;
        xs.live = ls;
        xs.fin_map = xm;
        HIPCHK(e, launch_exact_frames(lv.n, xpd, nullptr, 0, lv.S, 0, (int64_t)mlaunch + 1, xs, e->stream));
    } else {
        FrameSinks sk;
        sk.hist = reinterpret_cast<float*>(lv.d_cells);
        sk.hist_slots = lv.slots; sk.total_cols = INT64_MAX; sk.ring = 1;
        sk.live = ls;
        sk.fin_map = m;
        HIPCHK(e, launch_frames(lv.n, plan_dev(e, *p, lv.hop, lv.reassign), nullptr, 0, lv.S, 0, (int64_t)mlaunch + 1, sk, e->stream));
    }
    if (defer) HIPCHK(e, launch_live_flush(exact, ls, lv.d_cells, lv.slots, e->cfg.rows, lv.D, m, xm, lv.S, mlaunch, e->stream));
    if (post) {
        ls.out_db = dst_db;
        ls.out_rgba = reinterpret_cast<uint32_t*>(dst_rgba);
        ls.out_cols = out_cols;
        HIPCHK(e, launch_live_post(ls, lv.d_raw, raw_cols, e->cfg.rows, lv.D, e->smoothing, e->agc, e->cfg.db_top, m, lv.d_pstate, lv.S, e->stream));
    }
    return EMSPEC_OK;
}

// A failure between a launch and its synchronisation leaves the session half advanced and kernels in flight on buffers the
// caller owns: drain the stream and drop the session, so that the caller restarts from emspec_reset() semantics.
int live_abandon(emspec_engine* e, LiveState& lv, int code) {
    const std::string msg = e->err;
    (void)hipStreamSynchronize(e->stream);
    live_forget(lv);
    e->err = msg + " (stream state was reset)";
    return code;
}

// staging for the outputs when the caller's buffers are not page-locked: [S][cols][rows] x 4 bytes each
int live_out_staging(emspec_engine* e, LiveState& lv, bool want_db, bool want_rgba, int cols) {
    const size_t bytes = (size_t)lv.S * cols * e->cfg.rows * 4;
    int rc;
    if (want_db && (rc = pinned_grow(e, (void**)&lv.h_odb, &lv.odb_bytes, bytes))) return rc;
    if (want_rgba && (rc = pinned_grow(e, (void**)&lv.h_orgba, &lv.orgba_bytes, bytes))) return rc;
    return EMSPEC_OK;
}
}  // namespace

namespace {
// ---- the calls, on one of the engine's two sessions ----

int columns_impl(emspec_engine* e, LiveState& lv, const float* frames, int32_t streams, int32_t n, int32_t hop, int32_t reassign,
                 float* out_db, uint8_t* out_rgba, int32_t rows, int64_t* out_columns) {
    if (!e || !frames) return fail(e, EMSPEC_ERR_INVALID_ARG, "null argument");
    reassign = reassign ? 1 : 0;
    int rc = live_check(e, lv, streams, n, hop, reassign, rows, 1);
    if (rc) return rc;
    HIPCHK(e, hipSetDevice(e->device));
    if (lv.form == 0 && (rc = live_open(e, lv, streams, n, hop, reassign, 1))) return rc;
    const int S = lv.S, R = e->cfg.rows;
    // the frames: read by the kernel where they are when the caller's block is page-locked, else staged
    const float* src = reinterpret_cast<const float*>(device_view(frames));
    if (!src) {
        if ((rc = pinned_grow(e, (void**)&lv.h_fresh, &lv.fresh_bytes, (size_t)S * n * 4))) return rc;
        std::memcpy(lv.h_fresh, frames, (size_t)S * n * 4);
        src = lv.h_fresh;
    }
    float* ddb = reinterpret_cast<float*>(device_view(out_db));
    uint8_t* drgba = reinterpret_cast<uint8_t*>(device_view(out_rgba));
    const bool stage_db = out_db && !ddb, stage_rgba = out_rgba && !drgba;
    if ((stage_db || stage_rgba) && (rc = live_out_staging(e, lv, stage_db, stage_rgba, 1))) return rc;
    if (stage_db) ddb = lv.h_odb;
    if (stage_rgba) drgba = lv.h_orgba;
    LiveStream* desc = reinterpret_cast<LiveStream*>(lv.h_desc);
    for (int s = 0; s < S; ++s) desc[s] = LiveStream{lv.fed[s], lv.fed[s] * (long long)hop, 1, 0, 0, 0};
    if ((rc = live_launch(e, lv, src, n, 1, false, ddb, drgba, 1, 1, true))) return live_abandon(e, lv, rc);
    if (hipStreamSynchronize(e->stream) != hipSuccess) return live_abandon(e, lv, fail(e, EMSPEC_ERR_HIP, "hipStreamSynchronize failed"));
    if (stage_db) std::memcpy(out_db, lv.h_odb, (size_t)S * R * 4);
    if (stage_rgba) std::memcpy(out_rgba, lv.h_orgba, (size_t)S * R * 4);
    for (int s = 0; s < S; ++s) {
        const int64_t c = lv.fed[s] - lv.D;
        lv.fed[s] += 1;
        if (c >= 0) lv.emitted[s] = c + 1;
        if (out_columns) out_columns[s] = c >= 0 ? c : -1;
    }
    return EMSPEC_OK;
}

int flush_impl(emspec_engine* e, LiveState& lv, float* out_db, uint8_t* out_rgba, int32_t rows, int64_t* out_columns) {
    if (!e) return EMSPEC_ERR_INVALID_ARG;
    if (rows != e->cfg.rows) return fail(e, EMSPEC_ERR_INVALID_ARG, "rows does not match the engine configuration");
    bool any = false;
    for (int s = 0; s < lv.S; ++s) any = any || lv.emitted[s] < lv.fed[s];
    if (lv.form == 0 || !any) return fail(e, EMSPEC_ERR_STATE, "no pending column");
    HIPCHK(e, hipSetDevice(e->device));
    const int S = lv.S, R = e->cfg.rows;
    int rc;
    float* ddb = reinterpret_cast<float*>(device_view(out_db));
    uint8_t* drgba = reinterpret_cast<uint8_t*>(device_view(out_rgba));
    const bool stage_db = out_db && !ddb, stage_rgba = out_rgba && !drgba;
    if ((stage_db || stage_rgba) && (rc = live_out_staging(e, lv, stage_db, stage_rgba, lv.mmax))) return rc;
    if (stage_db) ddb = lv.h_odb;
    if (stage_rgba) drgba = lv.h_orgba;
    LiveStream* desc = reinterpret_cast<LiveStream*>(lv.h_desc);
    for (int s = 0; s < S; ++s) {
        const bool has = lv.emitted[s] < lv.fed[s];
        // (a stream with nothing pending emits the empty column: "column -1")
        desc[s] = LiveStream{has ? lv.emitted[s] + lv.D : (long long)lv.D - 1, 0, 0, 0, 0, 1};
    }
    if ((rc = live_launch(e, lv, nullptr, 0, 0, true, ddb, drgba, 1, 1, true))) return live_abandon(e, lv, rc);
    if (hipStreamSynchronize(e->stream) != hipSuccess) return live_abandon(e, lv, fail(e, EMSPEC_ERR_HIP, "hipStreamSynchronize failed"));
    if (stage_db) std::memcpy(out_db, lv.h_odb, (size_t)S * R * 4);
    if (stage_rgba) std::memcpy(out_rgba, lv.h_orgba, (size_t)S * R * 4);
    for (int s = 0; s < S; ++s) {
        const bool has = lv.emitted[s] < lv.fed[s];
        if (out_columns) out_columns[s] = has ? lv.emitted[s] : -1;
        if (has) lv.emitted[s] += 1;
    }
    return EMSPEC_OK;
}

int64_t push_columns_impl(const emspec_engine* e, const LiveState& lv, int64_t count, int32_t n, int32_t hop, int32_t reassign) {
    if (!e || count < 0 || !supported_fft(n) || hop < 1 || hop > n) return -1;
    const int D = latency(n, hop, reassign ? 1 : 0);
    auto cols = [&](int64_t fed, int64_t seen) {
        const int64_t after = frames_after(seen + count, n, hop);
        return (after > D ? after - D : 0) - (fed > D ? fed - D : 0);
    };
    if (lv.form != 2) return cols(0, 0);
    int64_t most = 0;
    for (int s = 0; s < lv.S; ++s) most = std::max(most, cols(lv.fed[s], lv.seen[s]));
    return most;
}

int push_impl(emspec_engine* e, LiveState& lv, const float* samples, int32_t streams, int64_t count, int64_t stride, int32_t n,
              int32_t hop, int32_t reassign, float* out_db, uint8_t* out_rgba, int32_t rows, int64_t max_columns,
              int64_t* out_counts, int64_t* out_first_columns) {
    if (!e || (!samples && count > 0) || count < 0 || stride < count) return fail(e, EMSPEC_ERR_INVALID_ARG, "null argument, negative count or stride < count");
    reassign = reassign ? 1 : 0;
    int rc = live_check(e, lv, streams, n, hop, reassign, rows, 2);
    if (rc) return rc;
    if (max_columns < 0) return fail(e, EMSPEC_ERR_INVALID_ARG, "max_columns must be >= 0");
    const int64_t expect = push_columns_impl(e, lv, count, n, hop, reassign);
    if ((out_db || out_rgba) && expect > max_columns)
        return fail(e, EMSPEC_ERR_INVALID_ARG, "output holds fewer columns than this block completes (" + std::to_string(expect) +
                                                   "); size it with emspec_push_columns() / emspec_push_columns_multi()");
    HIPCHK(e, hipSetDevice(e->device));
    if (lv.form == 0 && (rc = live_open(e, lv, streams, n, hop, reassign, 2))) return rc;
    const int S = lv.S, R = e->cfg.rows, D = lv.D;
    float* ddb = reinterpret_cast<float*>(device_view(out_db));
    uint8_t* drgba = reinterpret_cast<uint8_t*>(device_view(out_rgba));
    // page-locked outputs are written in place ([S][max_columns][rows]); others through a staging block per launch
    const bool direct = (!out_db || ddb) && (!out_rgba || drgba) && max_columns <= 0x7fffffff;
    if (!direct) {
        if ((rc = live_out_staging(e, lv, out_db != nullptr, out_rgba != nullptr, lv.mmax))) return rc;
        ddb = out_db ? lv.h_odb : nullptr;
        drgba = out_rgba ? lv.h_orgba : nullptr;
    }
    std::vector<int64_t> produced(S, 0), first(S, -1), nc(S, 0);
    std::vector<int> M(S, 0);
    LiveStream* desc = reinterpret_cast<LiveStream*>(lv.h_desc);
    int64_t used = 0;
    bool inflight = false;
    while (used < count) {
        // the kernel of the previous round reads the staging block and the descriptors: wait before refilling them
        if (inflight) {
            if (hipStreamSynchronize(e->stream) != hipSuccess) return live_abandon(e, lv, fail(e, EMSPEC_ERR_HIP, "hipStreamSynchronize failed"));
            inflight = false;
        }
        int maxpend = 0;
        for (int s = 0; s < S; ++s) maxpend = std::max(maxpend, lv.pend[s]);
        const int64_t take = std::min<int64_t>(count - used, lv.cap - maxpend);
        for (int s = 0; s < S; ++s) {
            std::memcpy(lv.h_fresh + (size_t)s * lv.cap + lv.pend[s], samples + (size_t)s * stride + used, (size_t)take * 4);
            lv.pend[s] += (int)take;
            lv.seen[s] += take;
        }
        used += take;
        int mx = 0;
        for (int s = 0; s < S; ++s) {
            M[s] = (int)(frames_after(lv.seen[s], n, hop) - lv.fed[s]);
            mx = std::max(mx, M[s]);
        }
        // A block that completes no frame (an audio worklet hands over 128 samples at a time) only joins the staging block:
        // no launch, no synchronisation until a frame is due or the block is full.
        if (mx == 0 && maxpend + take < lv.cap) continue;
        for (int s = 0; s < S; ++s) {
            const int64_t c0 = std::max<int64_t>(lv.fed[s] - D, 0), c1 = lv.fed[s] + M[s] - D;
            nc[s] = c1 > c0 ? c1 - c0 : 0;
            desc[s] = LiveStream{lv.fed[s], lv.newbase[s], M[s], lv.pend[s], direct ? (int)produced[s] : 0, 0};
            if (nc[s] > 0 && first[s] < 0) first[s] = c0;
        }
        if ((rc = live_launch(e, lv, lv.h_fresh, lv.cap, mx, false, ddb, drgba, direct ? (int)max_columns : lv.mmax,
                              direct ? (int)std::min<int64_t>(std::max<int64_t>(expect, 1), 0x7fffffff) : lv.mmax, false)))
            return live_abandon(e, lv, rc);
        inflight = true;
        if (!direct && (out_db || out_rgba)) {
            if (hipStreamSynchronize(e->stream) != hipSuccess) return live_abandon(e, lv, fail(e, EMSPEC_ERR_HIP, "hipStreamSynchronize failed"));
            inflight = false;
            for (int s = 0; s < S; ++s) {
                if (nc[s] <= 0) continue;
                const size_t from = (size_t)s * lv.mmax * R, to = ((size_t)s * max_columns + produced[s]) * R;
                if (out_db) std::memcpy(out_db + to, lv.h_odb + from, (size_t)nc[s] * R * 4);
                if (out_rgba) std::memcpy(out_rgba + to * 4, lv.h_orgba + from * 4, (size_t)nc[s] * R * 4);
            }
        }
        for (int s = 0; s < S; ++s) {
            lv.newbase[s] = lv.seen[s];
            lv.pend[s] = 0;
            lv.fed[s] += M[s];
            if (nc[s] > 0) { produced[s] += nc[s]; lv.emitted[s] = lv.fed[s] - D; }
        }
    }
    if (inflight && hipStreamSynchronize(e->stream) != hipSuccess) return live_abandon(e, lv, fail(e, EMSPEC_ERR_HIP, "hipStreamSynchronize failed"));
    for (int s = 0; s < S; ++s) {
        if (out_counts) out_counts[s] = produced[s];
        if (out_first_columns) out_first_columns[s] = first[s];
    }
    return EMSPEC_OK;
}

void live_free(LiveState& lv) {
    (void)hipFree(lv.d_cells); (void)hipFree(lv.d_sring); (void)hipFree(lv.d_done); (void)hipFree(lv.d_raw); (void)hipFree(lv.d_pstate);
    if (lv.h_desc) (void)hipHostFree(lv.h_desc);
    if (lv.h_fresh) (void)hipHostFree(lv.h_fresh);
    if (lv.h_odb) (void)hipHostFree(lv.h_odb);
    if (lv.h_orgba) (void)hipHostFree(lv.h_orgba);
    lv = LiveState{};
}
}  // namespace

namespace emspec {
void live_destroy(emspec_engine* e) { live_free(e->live); live_free(e->one); }
void live_reset(emspec_engine* e) { live_forget(e->live); live_forget(e->one); }
bool live_pending(const emspec_engine* e) {
    for (const LiveState* lv : {&e->live, &e->one})
        for (int s = 0; s < lv->S; ++s)
            if (lv->fed[s] > lv->emitted[s]) return true;
    return false;
}
}  // namespace emspec

extern "C" {

// ---- the live multi-stream session (e->live)
int32_t emspec_live_streams(const emspec_engine* e) { return e ? e->live.S : 0; }

int emspec_columns(emspec_engine* e, const float* frames, int32_t streams, int32_t n, int32_t hop, int32_t reassign,
                   float* out_db, uint8_t* out_rgba, int32_t rows, int64_t* out_columns) {
    if (!e) return EMSPEC_ERR_INVALID_ARG;
    return columns_impl(e, e->live, frames, streams, n, hop, reassign, out_db, out_rgba, rows, out_columns);
}

int emspec_columns_flush(emspec_engine* e, float* out_db, uint8_t* out_rgba, int32_t rows, int64_t* out_columns) {
    if (!e) return EMSPEC_ERR_INVALID_ARG;
    return flush_impl(e, e->live, out_db, out_rgba, rows, out_columns);
}

int64_t emspec_push_columns_multi(const emspec_engine* e, int64_t count, int32_t n, int32_t hop, int32_t reassign) {
    return e ? push_columns_impl(e, e->live, count, n, hop, reassign) : -1;
}

int emspec_push_samples_multi(emspec_engine* e, const float* samples, int32_t streams, int64_t count, int64_t stride,
                              int32_t n, int32_t hop, int32_t reassign, float* out_db, uint8_t* out_rgba, int32_t rows,
                              int64_t max_columns, int64_t* out_counts, int64_t* out_first_columns) {
    if (!e) return EMSPEC_ERR_INVALID_ARG;
    return push_impl(e, e->live, samples, streams, count, stride, n, hop, reassign, out_db, out_rgba, rows, max_columns, out_counts,
                     out_first_columns);
}

// ---- the renderer's own calls: ONE stream, the same machinery on the engine's second session (e->one).  Until round 5
// these ran separate code: a frame copy + one to three launches per call (27.8 us, EXACT 39.7 us per emspec_column call).
int emspec_column(emspec_engine* e, const float* frame, int32_t n, int32_t hop, int32_t reassign, float* out_db,
                  uint8_t* out_rgba, int32_t rows, int64_t* out_column) {
    if (!e) return EMSPEC_ERR_INVALID_ARG;
    int64_t c = -1;
    const int rc = columns_impl(e, e->one, frame, 1, n, hop, reassign, out_db, out_rgba, rows, &c);
    if (rc == EMSPEC_OK && out_column) *out_column = c;
    return rc;
}

int emspec_column_flush(emspec_engine* e, float* out_db, uint8_t* out_rgba, int32_t rows, int64_t* out_column) {
    if (!e) return EMSPEC_ERR_INVALID_ARG;
    int64_t c = -1;
    const int rc = flush_impl(e, e->one, out_db, out_rgba, rows, &c);
    if (rc == EMSPEC_OK && out_column) *out_column = c;
    return rc;
}

int64_t emspec_push_columns(const emspec_engine* e, int64_t count, int32_t n, int32_t hop, int32_t reassign) {
    return e ? push_columns_impl(e, e->one, count, n, hop, reassign) : -1;
}

int emspec_push_samples(emspec_engine* e, const float* samples, int64_t count, int32_t n, int32_t hop, int32_t reassign,
                        float* out_db, uint8_t* out_rgba, int32_t rows, int64_t max_columns, int64_t* out_count,
                        int64_t* out_first_column) {
    if (!e) return EMSPEC_ERR_INVALID_ARG;
    int64_t cnt = 0, first = -1;
    const int rc = push_impl(e, e->one, samples, 1, count, count, n, hop, reassign, out_db, out_rgba, rows, max_columns, &cnt, &first);
    if (rc == EMSPEC_OK) {
        if (out_count) *out_count = cnt;
        if (out_first_column) *out_first_column = first;
    }
    return rc;
}

#ifdef EMSPEC_DIAG
// diagnostic build (include/emspec_debug.h): where the live frame kernel stamps its phases - stamps[S][8], page-locked host
// memory of the caller (NULL: off)
int emspec_debug_live_stamps(emspec_engine* e, uint64_t* stamps) {
    if (!e) return EMSPEC_ERR_INVALID_ARG;
    e->live.stamps = reinterpret_cast<unsigned long long*>(device_view(stamps));
    if (stamps && !e->live.stamps) return fail(e, EMSPEC_ERR_INVALID_ARG, "stamps must be page-locked host memory");
    return EMSPEC_OK;
}
#endif

int emspec_reset_stream(emspec_engine* e, int32_t stream) {
    if (!e) return EMSPEC_ERR_INVALID_ARG;
    LiveState& lv = e->live;
    if (lv.form == 0) return fail(e, EMSPEC_ERR_STATE, "no live session");
    if (stream < 0 || stream >= lv.S) return fail(e, EMSPEC_ERR_INVALID_ARG, "stream out of range");
    HIPCHK(e, hipSetDevice(e->device));
    const size_t cellb = e->exact() ? 8 : 4, per = (size_t)lv.slots * e->cfg.rows * cellb;
    HIPCHK(e, hipMemsetAsync(reinterpret_cast<char*>(lv.d_cells) + (size_t)stream * per, 0, per, e->stream));
    if (lv.d_pstate)
        HIPCHK(e, hipMemsetAsync(lv.d_pstate + (size_t)stream * (e->cfg.rows + 4), 0, (size_t)(e->cfg.rows + 4) * 4, e->stream));
    lv.fed[stream] = 0; lv.emitted[stream] = 0; lv.seen[stream] = 0; lv.newbase[stream] = 0; lv.pend[stream] = 0;
    return EMSPEC_OK;
}

}  // extern "C"
