// emspec_launch.h — host-callable launchers of the HIP kernels (kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "emspec_device.h"

namespace emspec {

// ---- live multi-stream streaming (emspec_columns / emspec_push_samples_multi; live.hip.inc) ----
// One launch serves S live streams: grid = (frames per stream + 1, S).  Workgroup (f, s) with f < frames[s] transforms
// frame j0[s] + f of stream s and scatters it into that stream's column ring in HBM (device-scope atomics); the LAST of a
// stream's workgroups to finish (an arrival counter per stream) finalises the columns the launch completed, straight into the
// caller's (page-locked) output, and clears their ring slots.  Workgroup (gridDim.x - 1, s) moves the stream's new samples
// from the page-locked staging block into its device sample ring for later launches.
struct LiveStream {        // per stream and launch; read by the kernel from page-locked host memory
    long long j0;          // absolute index of the stream's first frame in this launch (= frames fed so far)
    long long newbase;     // absolute index of fresh[s][0]: samples below it are in the device sample ring
    int frames;            // frames of this stream in this launch (0 .. gridDim.x - 1)
    int newcount;          // samples in fresh[s] to move into the sample ring
    int out_at;            // first column slot of the stream's output block this launch writes
    int flush;             // != 0: no frames; finalise `flush` pending columns starting at column j0 - D (emspec_columns_flush)
};
struct LiveSinks {
    const LiveStream* streams = nullptr;   // [S]; null: not a live launch
    LiveStream uni{};                      // uniform != 0: every stream's descriptor (the kernel then does not read `streams`)
    int uniform = 0;
    const float* fresh = nullptr;          // [S][fresh_stride] samples the device ring does not hold yet
    long long fresh_stride = 0;
    float* sring = nullptr;                // [S][ring_mask + 1] device sample ring: absolute sample a sits at a & ring_mask
    int ring_mask = 0;
    unsigned* done = nullptr;              // [S] arrival counters, zero between launches
    float* out_db = nullptr;               // [S][out_cols][rows], or null
    uint32_t* out_rgba = nullptr;          // [S][out_cols][rows] RGBA8, or null
    int out_cols = 0;
    int empty_col = 0;                     // per-frame form: a frame that completes no column yet emits the empty column
    int defer_finalize = 0;                // the frame kernel only scatters; live_finalize_kernel (one workgroup per column) follows
    const uint32_t* lut = nullptr;
#ifdef EMSPEC_DIAG
    unsigned long long* stamps = nullptr;  // [S][8] 100 MHz wall-clock stamps of each stream's first workgroup (tools/live_phases.py)
#endif
};

// Where frames_kernel sends its per-bin results.
struct FrameSinks {
    // parity dump (all three or none): [stream][frame][K]
    float* power = nullptr;
    int32_t* col = nullptr;
    int32_t* row = nullptr;
    // compact per-bin records for the tile-scatter kernel: [stream][frame][K] of
    // (float bits of |X_h|^2, key) with key = (dcol+32768)<<16 | row, or 0xFFFFFFFF when dropped
    uint2* records = nullptr;
    // histogram scatter (global float atomics): hist[stream][slots][rows]
    float* hist = nullptr;
    int64_t hist_slots = 0;    // slots per stream in `hist`
    int64_t total_cols = 0;    // valid absolute columns are [0,total_cols)
    int32_t ring = 0;          // !=0: slot = col % hist_slots (streaming ring), else slot = col
    int64_t col_offset = 0;    // added to the frame index to get its absolute column (streaming)
    LiveSinks live{};                 // live multi-stream launch (uses hist / hist_slots / fin_map)
    DbMap fin_map{};                  // ... its "dB + colour" stage
};

// Where the exact-mode frame kernels send their per-bin results (exact.hip.inc)
struct ExactSinks {
    // parity dump (power/col/row together, q optional): [stream][frame][K]
    double* power = nullptr;
    int32_t* col = nullptr;
    int32_t* row = nullptr;
    long long* q = nullptr;
    // per-bin records for exact_tile_scatter_kernel: [stream][frame][exact_record_stride(n)] of the bin's fixed-point
    // energy and key = (dcol+32768)<<16 | row, or 0xFFFFFFFF when dropped
    long long* rec_q = nullptr;
    uint32_t* rec_key = nullptr;
    // histogram (global u64 atomics; the streaming ring): hist[stream][slots][rows]
    unsigned long long* hist = nullptr;
    int64_t hist_slots = 0;
    int64_t total_cols = 0;
    int32_t ring = 0;
    int64_t col_offset = 0;
    int32_t edges_lds = 1;     // set by the launcher: the binary64 edge table is staged in LDS (0: read from global memory)
    LiveSinks live{};          // live multi-stream launch (uses hist / hist_slots / fin_map)
    ExactDbMap fin_map{};
};
hipError_t launch_exact_frames(int n, const ExactPlanDev& pl, const float* pcm, int64_t L, int S, int64_t frame0,
                               int64_t nframes, const ExactSinks& sinks, hipStream_t st);
int exact_record_stride(int n);
// (ebin_f32 / low / low_bytes: the plan's float32 edge table on the host and a zeroed u64 scratch of
// exact_scatter_scratch_bytes - with them a ring that does not fit LDS is walked with its sparse low rows in that scratch,
// every record read once; without them, or on an axis with > 6 % of the bins down there: 16-column tiles, records read 3x)
size_t exact_scatter_scratch_bytes(int n, const ExactPlanDev& pl, int S, int64_t C, const float* ebin_f32);
hipError_t launch_exact_tile_scatter(const long long* rec_q, const uint32_t* rec_key, int n, const ExactPlanDev& pl,
                                     const ExactDbMap& m, const uint8_t* lut, int S, int64_t C, float* db, uint8_t* rgba,
                                     uint8_t* index, hipStream_t st, const float* ebin_f32 = nullptr,
                                     unsigned long long* low = nullptr, size_t low_bytes = 0);
// EXACT mode, one kernel (exact_fused.hip.inc): N = 4096 at every hop whose u64 column ring fits in LDS
bool exact_fused_supported(int n, const ExactPlanDev& pl);
hipError_t launch_exact_fused(int n, const ExactPlanDev& pl, const ExactDbMap& m, const uint8_t* lut, const float* pcm,
                              int64_t L, int S, int64_t C, float* db, uint8_t* rgba, uint8_t* index, hipStream_t st,
                              unsigned long long* stamps = nullptr, int64_t* stamp_groups = nullptr);
// EXACT mode, one kernel without parking (exact_fused_lr.hip.inc): the ring's rows >= rl in LDS beside the planes, rows < rl
// in a per-workgroup scratch in global memory (device-scope atomics only).  exact_fused_lr_low_rows: rl for the shape
// (rows, D), or -1 when the shape is not served; whether the AXIS is (few bins below row rl) is the caller's decision.
int exact_fused_lr_low_rows(int n, const ExactPlanDev& pl);
size_t exact_fused_lr_scratch_bytes(int n, const ExactPlanDev& pl, int rl, int S, int64_t C);
hipError_t launch_exact_fused_lr(int n, const ExactPlanDev& pl, const ExactDbMap& m, const uint8_t* lut, const float* pcm,
                                 int64_t L, int S, int64_t C, int rl, unsigned long long* low, size_t low_bytes, float* db,
                                 uint8_t* rgba, uint8_t* index, hipStream_t st, unsigned long long* stamps = nullptr,
                                 int64_t* stamp_groups = nullptr);

// the device word a kernel raises when a bounded wait times out (kernels.hip: g_kernel_error); -1 if it cannot be read
int read_kernel_error(bool clear);

bool supported_fft(int n);

// One workgroup per frame: frames [frame0, frame0+nframes) of each of S streams.
hipError_t launch_frames(int n, const PlanDev& pl, const float* pcm, int64_t L, int S,
                         int64_t frame0, int64_t nframes, const FrameSinks& sinks, hipStream_t st);

// records of all C frames of S streams -> finished columns.  One workgroup per (32-column tile,
// stream): LDS histogram of the tile, frames [c0-D, c0+tile+D) streamed through it.
hipError_t launch_tile_scatter(const uint2* records, int n, const PlanDev& pl, const DbMap& m, const uint8_t* lut,
                               int S, int64_t C, float* db, uint8_t* rgba, uint8_t* index, hipStream_t st);

// Fused batch path (LDS column ring): returns hipErrorNotSupported when (n,hop,rows)
// has no fused specialisation; the caller then uses launch_frames + launch_tile_scatter.
hipError_t launch_fused(int n, const PlanDev& pl, const DbMap& m, const uint8_t* lut,
                        const float* pcm, int64_t L, int S, int64_t total_cols,
                        float* db, uint8_t* rgba, uint8_t* index, hipStream_t st,
                        unsigned long long* stamps = nullptr, int64_t* stamp_groups = nullptr);
// display post-process (AGC + temporal smoothing) over finished columns, see post.hip.inc
hipError_t launch_postprocess(const float* db, float* out_db, uint8_t* rgba, uint8_t* index, int S, int64_t C, int R,
                              float sm, float agc, float db_top, const DbMap& dm, const uint8_t* lut, float* peak,
                              float* gain, hipStream_t st);
// live multi-stream calls (live_launch.hip.inc): flush of pending columns, display post-process of a launch's columns.
// The frame launches themselves go through launch_frames / launch_exact_frames with sinks.live set and
// nframes = (largest per-stream frame count) + 1.
// (ncol: the largest number of columns any stream emits - grid (ncol, S), one workgroup per column)
hipError_t launch_live_flush(bool exact, const LiveSinks& lv, void* cells, int slots, int rows, int D, const DbMap& m,
                             const ExactDbMap& xm, int S, int ncol, hipStream_t st);
hipError_t launch_live_post(const LiveSinks& lv, const float* raw, int raw_cols, int rows, int D, float sm, float agc, float db_top,
                            const DbMap& dm, float* pstate, int S, hipStream_t st);
// the gather's wire image (pack.hip.inc)
int64_t wire_bound_bytes(int64_t columns, int rows);
int64_t wire_fixed_bytes(int64_t columns, int rows);
size_t wire_scratch_bytes(int64_t columns);
hipError_t launch_wire_pack(const uint8_t* index, int64_t columns, int rows, uint8_t* wire, void* scratch, hipStream_t st);
hipError_t launch_wire_unpack(const uint8_t* wire, int64_t columns, int rows, uint8_t* index, hipStream_t st);
bool fused_supported(int n, int hop, int rows, int reassign);
int device_cus();           // compute units of the current device
#ifdef EMSPEC_DIAG          // diagnostic build only (libemspec_diag.so, include/emspec_debug.h)
int fused_waves_per_group();
int fused_read_errflag();   // non-zero if a bounded spin of the decoupled-team kernel ever timed out
hipError_t launch_row_lookup_probe(const float* ebin, int rows, const float* kh, int64_t count, int32_t* out_hint,
                                   int32_t* out_exact, hipStream_t st);
hipError_t launch_occupy(int groups, int usec, unsigned* sink, hipStream_t st);
hipError_t launch_recip_probe(const float* d, int64_t count, float* out_short, float* out_ieee, hipStream_t st);
hipError_t launch_recip64_probe(const double* d, int64_t count, double* out_short, double* out_ieee, hipStream_t st);
#endif

}  // namespace emspec
