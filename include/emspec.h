/*
 * emspec.h — C ABI of libemspec, the MI355X reassigned-spectrogram engine.
 *
 * This is the drop-in boundary.  The reference
 * (effree/EM-Spec) ships binaries only: its source is private
 * (/root/reference/README.md:73), so NO reference FFI interface exists to
 * cite.  The only interface the reference side names is the renderer call
 *
 *     computeSpectrogramColumn(audioFrame, fftSize, hop, reassign)
 *
 * (BASELINE.json:north_star).  Every entry point below is therefore
 * [BUILD-DEFINED]; the comment on each one says which part of that call (or
 * which documented feature of the reference: README.md line) it serves.
 * INTEGRATION.md shows the N-API binding a maintainer adds on the
 * Electron/Node side.
 *
 * Conventions
 *   - plain C, no C++ types, no exceptions cross this boundary;
 *   - every function returns EMSPEC_OK (0) or a negative emspec_status;
 *     the message is available from emspec_last_error();
 *   - the caller owns every input and output buffer; buffers are borrowed
 *     for the duration of the call only;
 *   - one engine = one HIP device + one HIP stream; an engine is not
 *     thread-safe, calls on one engine are serialised by the caller;
 *   - there is NO CPU fallback: if no gfx950 device or no kernel image is
 *     available emspec_create fails with EMSPEC_ERR_NO_DEVICE.
 */
#ifndef EMSPEC_H
#define EMSPEC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 4): emspec_config.mode is read (it was a reserved, ignored field in version 1, so a version-1 library would
 * silently run an EXACT-mode request in float32: emspec_create now rejects the mismatch); added emspec_mode,
 * emspec_build_info, emspec_device_status, emspec_comm_set_timeout; emspec_uses_fused answers for EXACT-mode engines too. */
/* (round 6 added the live multi-stream functions - emspec_columns ... emspec_live_streams - and changed no struct: still 2) */
#define EMSPEC_ABI_VERSION 2

typedef enum emspec_status {
    EMSPEC_OK = 0,
    EMSPEC_ERR_INVALID_ARG = -1,   /* bad size / null pointer / unsupported N,hop */
    EMSPEC_ERR_NO_DEVICE = -2,     /* no HIP device, or not gfx950 */
    EMSPEC_ERR_HIP = -3,           /* a HIP runtime call failed (message has the HIP error) */
    EMSPEC_ERR_OUT_OF_MEMORY = -4,
    EMSPEC_ERR_STATE = -5,         /* call sequence error (e.g. N/hop changed mid-stream without reset) */
    EMSPEC_ERR_COMM = -6           /* an RCCL call failed, or the ranks of a gather disagree (message has the detail) */
} emspec_status;

/* Output selection bits for emspec_out / emspec_out_device. */
#define EMSPEC_OUT_DB 1u     /* float32 dB per (column,row)               */
#define EMSPEC_OUT_RGBA 2u   /* uint8[4] colour per (column,row)          */
#define EMSPEC_OUT_INDEX 4u  /* uint8 palette index 0..255 per (column,row) */

/*
 * Engine configuration: the display-side settings the reference exposes as
 * sliders (README.md:43-51, assets/settings.png) plus the analysis grid.
 * All fields are plain scalars; zero-initialise then call
 * emspec_default_config().
 */
typedef struct emspec_config {
    int32_t abi_version;   /* must be EMSPEC_ABI_VERSION */
    int32_t device;        /* HIP device ordinal */
    int32_t rows;          /* R: log-frequency rows per column (64..4096), default 1024 */
    int32_t mode;          /* EMSPEC_MODE_FAST (0, default) or EMSPEC_MODE_EXACT; see below */
    float sample_rate;     /* Hz, default 48000 */
    float fmin_hz;         /* lowest row edge, default 20 */
    float fmax_hz;         /* highest row edge, default sample_rate/2 */
    float gain;            /* linear amplitude gain before dB ("Gain", README.md:47) */
    float db_top;          /* dB value mapped to palette index 255, default 0 */
    float db_range;        /* dB span mapped onto the palette ("dB Range", README.md:46) */
    float gate_db;         /* cells below this dB are drawn as index 0 ("Noise Gate", README.md:51) */
    float power_floor;     /* bins with |X_h|^2 below this are not reassigned/accumulated */
} emspec_config;

/*
 * Arithmetic modes [BUILD-DEFINED] (north_star: results must match the reference JS path "within 1e-4 relative on
 * magnitude and exactly on reassigned integer (time,freq) bin indices"; JavaScript arithmetic is IEEE binary64).
 *   EMSPEC_MODE_FAST   float32 throughout: indices are exact against the float32 bit model; against a
 *                      binary64 evaluation a few 1e-4 of the bins that sit on a cell edge land in the neighbouring
 *                      cell, and cells are summed in arrival order (dB reproducible to a few ulp).
 *   EMSPEC_MODE_EXACT  binary64 from the frame to the indices, energy summed in 64-bit fixed point (order-
 *                      independent), dB through a specified binary32 evaluation (no library call, no division; DESIGN.md 3.7;
 *                      within 2e-5 dB of the real value): (column,row) agree
 *                      with a float64 implementation of the three-window method on every bin, and dB / palette index /
 *                      RGBA are bit-reproducible run to run and equal to the CPU bit model's bytes (every table - twiddles,
 *                      log-spaced row edges, dB polynomial - is a specified sequence of IEEE operations, no libm call, so the
 *                      bytes do not depend on the host's C library; "(column,row) agree with a
 *                      float64 implementation on every bin" is a measured statement: 0 mismatches in > 10^8 values, where
 *                      two binary64 evaluations may still differ on a bin within ~1e-13 of an edge).  Inputs must stay
 *                      within |x| <= 4 (the fixed point covers 2^11 full-scale-sine powers per cell).  Same entry points;
 *                      emspec_parity_dump_exact replaces emspec_parity_dump.  Roughly 2.1x slower than the fast mode at N = 4096.
 */
#define EMSPEC_MODE_FAST 0
#define EMSPEC_MODE_EXACT 1

typedef struct emspec_engine emspec_engine;

/* Host-memory outputs of emspec_batch: any pointer may be NULL (not wanted).
 * Layout of each: [streams][columns][rows] (+[4] for rgba), C order. */
typedef struct emspec_out {
    float* db;
    uint8_t* rgba;
    uint8_t* index;
} emspec_out;

/* Fill *cfg with the defaults listed above. */
int emspec_default_config(emspec_config* cfg);

/* Create an engine on cfg->device.  Serves: engine construction behind the
 * renderer's first computeSpectrogramColumn call. */
int emspec_create(const emspec_config* cfg, emspec_engine** out_engine);
void emspec_destroy(emspec_engine* e);

/* The arithmetic mode this engine really runs in (EMSPEC_MODE_FAST / EMSPEC_MODE_EXACT), -1 for NULL: lets a binding
 * assert that a request for bit-exact results was honoured. */
int32_t emspec_mode(const emspec_engine* e);

/* What this library was built from: "emspec abi=2 sources=<16 hex digits> arch=gfx950", the digits being the sha1 of the
 * kernel / C-ABI sources (tools/sources_sha.py) taken when the library was compiled.  bench.py quotes profile-derived
 * numbers only while this, the source tree and the profile agree.  Never NULL; static storage. */
const char* emspec_build_info(void);

/* Synchronises the engine's device and reports what its kernels may have flagged since the last call: EMSPEC_OK, or
 * EMSPEC_ERR_HIP when a kernel's bounded wait timed out (the fused kernels order some phases with arrival counters in
 * LDS; every wait is bounded so that a protocol error cannot hang the device - the results of that launch are then
 * invalid) or the device reports an asynchronous error.  The synchronous batch entry points (emspec_batch,
 * emspec_batch_gather) check it themselves; callers of emspec_batch_device call this after their own stream
 * synchronisation when they want the check (the streaming calls run kernels without such waits). */
int emspec_device_status(emspec_engine* e);

/* Last error message of this engine (or of the failed emspec_create when
 * e == NULL).  Never NULL; valid until the next call on the same thread. */
const char* emspec_last_error(const emspec_engine* e);

/* Upload a 256-entry RGBA palette ("Color Map", README.md:15).  The default
 * is the 5-stop gradient measured from assets/settings.png. */
int emspec_set_colormap(emspec_engine* e, const uint8_t* rgba256x4);

/*
 * Frequency axis.  By default the R rows are log-spaced between fmin_hz and fmax_hz.
 * emspec_set_row_edges_hz installs any strictly increasing table of rows+1 edges in
 * (0, sample_rate/2] instead — the hook for the reference's "Frequency Scale" (zoom) and
 * "Low-End Boost" sliders (README.md:48-49), whose laws are undocumented and therefore
 * left to the host (em-spec_amd/js/index.js ships one).  NULL restores the log axis.
 * Row r collects the bins whose reassigned frequency lies in [edge[r], edge[r+1]).
 * emspec_get_row_edges_hz returns the table in use: the inverse map the shift+hover
 * frequency read-out needs (README.md:39).  Not allowed while streaming columns are pending.
 */
int emspec_set_row_edges_hz(emspec_engine* e, const float* edges_hz, int32_t count);

/*
 * Display-parameter laws [BUILD-DEFINED] (the reference's slider laws are undocumented,
 * README.md:44-51), host-independent so that every binding draws the same picture.  No engine needed.
 * emspec_warped_edges_hz: rows+1 edges for emspec_set_row_edges_hz from "Frequency Scale" (zoom:
 * the axis spans fmin .. fmin*(fmax/fmin)^(1/freq_scale)) and "Low-End Boost" (row r sits at
 * u = (r/rows)^low_end_boost along the log range; > 1 gives the low end more rows); (1, 1) is the
 * plain log axis.  emspec_make_colormap: the reference's 5-stop ramp scaled by
 * "Brightness" (0.5 = as measured), 256 RGBA entries for emspec_set_colormap.
 */
int emspec_warped_edges_hz(int32_t rows, float fmin_hz, float fmax_hz, float low_end_boost,
                           float freq_scale, float* out_edges_hz);
int emspec_make_colormap(float brightness, uint8_t* out_rgba256x4);
int emspec_get_row_edges_hz(emspec_engine* e, float* edges_hz, int32_t count);

/*
 * Display post-process over finished columns, per stream and in time order (README.md:14 "adaptive
 * brightness", README.md:50 "temporal smoothing"; laws [BUILD-DEFINED]):
 *   agc_strength in [0,1]: a column-peak follower (fast attack, slow release) shifts every column by
 *                          agc_strength * (db_top - level) dB, limited to +-40 dB;
 *   smoothing    in [0,0.95]: y[c] = smoothing*y[c-1] + (1-smoothing)*x[c] per row, on the dB values.
 * Both 0 (the default) = off.  Applies to emspec_batch / _device and to the streaming call (whose
 * state emspec_reset clears).  The dB output is the post-processed value; index and RGBA follow it.
 */
int emspec_set_display(emspec_engine* e, float smoothing, float agc_strength);

/* Number of columns a stream of L samples yields: (L-n)/hop+1, or 0. */
int64_t emspec_num_columns(int64_t L, int32_t n, int32_t hop);

/* Time reassignment moves energy up to ceil(n/(2*hop)) columns either way,
 * so the streaming call returns column j-D when fed frame j.  This is D. */
int32_t emspec_latency_columns(int32_t n, int32_t hop, int32_t reassign);

/*
 * Streaming, one frame per call — THE call the renderer makes:
 *   computeSpectrogramColumn(audioFrame, fftSize, hop, reassign)
 * frame: n float32 samples (frame j of the stream = samples [j*hop, j*hop+n)).
 * The engine keeps the pending-column ring on the device.  With D =
 * emspec_latency_columns(), call number j (0-based) writes finished column
 * j-D; for j < D the output is the empty column (all cells at the dB floor)
 * and *out_column (if non-NULL) is set to -1.  out_db (rows floats) and
 * out_rgba (rows*4 bytes) may each be NULL.
 * Changing n/hop/reassign between calls without emspec_reset() is
 * EMSPEC_ERR_STATE.
 */
int emspec_column(emspec_engine* e, const float* frame, int32_t n, int32_t hop,
                  int32_t reassign, float* out_db, uint8_t* out_rgba,
                  int32_t rows, int64_t* out_column);

/* Flush: emit the next of the D columns still pending after the last frame
 * (feeds no new samples).  Returns EMSPEC_ERR_STATE when nothing is pending.
 * A flushed stream is at its end - later frames would still have added to the
 * columns a flush emits - so feeding it again without emspec_reset() is
 * EMSPEC_ERR_STATE. */
int emspec_column_flush(emspec_engine* e, float* out_db, uint8_t* out_rgba,
                        int32_t rows, int64_t* out_column);

/*
 * Streaming by sample blocks: the audio callback hands over whatever block it
 * has (any count >= 0); the engine keeps the unconsumed tail of the stream on
 * the device (sample ring) next to the pending-column ring, and every `hop`
 * new samples complete one frame.  Finished columns (column c is complete once
 * frame c+D has been seen) are written oldest first to out_db [max_columns][rows]
 * and/or out_rgba [max_columns][rows][4]; *out_count receives how many,
 * *out_first_column the absolute index of the first (-1 if none).  Unlike
 * emspec_column no empty columns are produced while the ring primes.
 * emspec_push_columns() tells in advance how many columns a block of `count`
 * samples will complete (for sizing the outputs; -1 on invalid arguments); a
 * block that completes more than max_columns is rejected before any state changes.
 * The D columns pending after the last block are drained with emspec_column_flush;
 * samples short of a hop are dropped by emspec_reset.  One stream must be fed
 * through either emspec_column or emspec_push_samples, not both (EMSPEC_ERR_STATE).
 */
int64_t emspec_push_columns(const emspec_engine* e, int64_t count, int32_t n,
                            int32_t hop, int32_t reassign);
int emspec_push_samples(emspec_engine* e, const float* samples, int64_t count,
                        int32_t n, int32_t hop, int32_t reassign, float* out_db,
                        uint8_t* out_rgba, int32_t rows, int64_t max_columns,
                        int64_t* out_count, int64_t* out_first_column);

/* Drop all per-stream state (sample position, pending ring), of the single-stream calls above and of the live
 * multi-stream session below. */
int emspec_reset(emspec_engine* e);

/*
 * ---- Live multi-stream streaming: S streams advance together, ONE kernel launch and ONE synchronisation per call.
 * Serves: BASELINE.json configs[2] ("64 concurrent 48 kHz streams") in the form the renderer calls it - north_star's
 * per-frame computeSpectrogramColumn(audioFrame, fftSize, hop, reassign), README.md:36 ("automatically start visualizing
 * your system audio") - where one engine per stream would cost S launches + S synchronisations per hop on the host
 * thread.  [BUILD-DEFINED]; same arithmetic, same results as emspec_column / emspec_push_samples / emspec_batch on every
 * stream (FAST mode: the float32 sums are taken in arrival order, dB equal to a few ulp; EXACT mode: equal bytes).
 *
 * The first call fixes the session: streams, fft size, hop, reassign and the feeding form (per frame OR per sample block);
 * changing any of them later is EMSPEC_ERR_STATE until emspec_reset().  The session is independent of the single-stream
 * state of emspec_column / emspec_push_samples (which are the one-stream case of the same code, on a session of their own).  Every stream has its own position: emspec_reset_stream(e, s) restarts
 * stream s (its sample position, pending columns and display post-process state) while the others continue.
 * Buffers: any host memory works; when a buffer is page-locked (emspec_host_alloc) the kernel reads / writes it in place
 * (no staging copy on the host thread) - for 64 streams of n = 4096 that is 1 MB of frames per emspec_columns call.
 *
 * emspec_columns: per-frame form.  frames[streams][n]: frame j_s of each stream (stream s has been fed j_s frames so far).
 *   out_db[streams][rows] / out_rgba[streams][rows][4] (each may be NULL) receive column j_s - D of every stream
 *   (D = emspec_latency_columns), the empty column while a stream's ring primes; out_columns[streams] (optional) the
 *   column index, -1 for the empty column.
 * emspec_columns_flush: every stream that still has pending columns emits its next one (others: the empty column, -1).
 *   EMSPEC_ERR_STATE when no stream has any.  Works for both feeding forms.  A flushed stream is at its end: feeding the
 *   session again is EMSPEC_ERR_STATE until that stream is restarted (emspec_reset_stream) or the session reset.
 * emspec_push_samples_multi: per-sample-block form.  samples: `count` new samples of every stream, stream s at
 *   samples + s * stride (stride >= count: a window of a larger [streams][...] array works).  Finished columns of stream s
 *   go to out_db[s][0 .. out_counts[s]) of out_db[streams][max_columns][rows] (and / or out_rgba, same layout + [4]),
 *   oldest first; out_first_columns[s] = absolute index of the first, -1 if none.  No empty columns.  A block that
 *   completes more than max_columns columns on some stream is rejected before any state changes
 *   (emspec_push_columns_multi = the largest per-stream count a block of `count` samples will complete).  Blocks that
 *   complete no frame only join a staging block on the host: no launch.
 * emspec_live_streams: streams of the current session (0 = none).
 */
int emspec_columns(emspec_engine* e, const float* frames, int32_t streams, int32_t n, int32_t hop, int32_t reassign,
                   float* out_db, uint8_t* out_rgba, int32_t rows, int64_t* out_columns);
int emspec_columns_flush(emspec_engine* e, float* out_db, uint8_t* out_rgba, int32_t rows, int64_t* out_columns);
int64_t emspec_push_columns_multi(const emspec_engine* e, int64_t count, int32_t n, int32_t hop, int32_t reassign);
int emspec_push_samples_multi(emspec_engine* e, const float* samples, int32_t streams, int64_t count, int64_t stride,
                              int32_t n, int32_t hop, int32_t reassign, float* out_db, uint8_t* out_rgba, int32_t rows,
                              int64_t max_columns, int64_t* out_counts, int64_t* out_first_columns);
int emspec_reset_stream(emspec_engine* e, int32_t stream);
int32_t emspec_live_streams(const emspec_engine* e);

/*
 * Batched throughput entry point, host buffers: S streams of L samples
 * (pcm[S][L], row-major) -> every finished column of every stream.
 * Copies in, runs the fused column kernel, copies the selected outputs back.
 * When every buffer is page-locked (emspec_host_alloc) the call runs a three-stage pipeline over ~16 chunks of streams -
 * host->device copies, kernels, device->host copies on three HIP streams of the engine - in either arithmetic mode and with
 * the display post-process; what bounds it is PCIe (uint8 index out: ~45 GB/s each way at once on a Gen5 x16 link).
 * Ordinary (pageable) buffers run the same pipeline: the runtime's copies block the calling thread, so the library starts a
 * second host thread for the copies out (and three that touch the destination's pages ahead of it - a result array allocated
 * per call has none resident), all joined before the call returns: the page-locked rate when the pages are resident (4.3e7
 * columns/s on the bench shape), a third of it into a fresh array.  With fewer than sixteen streams (one long recording) the
 * pipeline's units are runs of >= 16,384 columns of a stream, each computed from the frames that reach it - the same columns
 * (EXACT mode: the same bytes) as in one piece; a batch too short for two units takes one copy in, the kernels, one copy out.
 * With the display post-process on, units are whole streams.  Serves: the renderer-side batched computeColumns of the N-API
 * addon (em-spec_amd/js/index.js).
 */
int emspec_batch(emspec_engine* e, const float* pcm, int32_t S, int64_t L,
                 int32_t n, int32_t hop, int32_t reassign, const emspec_out* out);

/*
 * Page-locked host memory for the buffers of emspec_batch / emspec_column / the live calls: copies from and to it
 * are asynchronous and need no extra host thread, and the streaming kernels read / write it in place (pageable
 * memory costs the streaming calls a staging copy each way).  Optional: any host memory works.  Needs no engine;
 * free with emspec_host_free.
 */
int emspec_host_alloc(size_t bytes, void** out);
void emspec_host_free(void* p);

/*
 * Same, device-resident: pcm and the outputs are device pointers on the
 * engine's device; the kernels are enqueued on hip_stream (a hipStream_t
 * passed as void*, NULL = the HIP default stream) and the call returns
 * without synchronising.  Output pointers may be NULL.  Calls on one engine
 * share its device workspaces: enqueue them on one stream, or order them
 * yourself (events) when using several.
 */
int emspec_batch_device(emspec_engine* e, const float* pcm_dev, int32_t S,
                        int64_t L, int32_t n, int32_t hop, int32_t reassign,
                        float* db_dev, uint8_t* rgba_dev, uint8_t* index_dev,
                        void* hip_stream);

/*
 * Parity dump: per-bin results of frames [frame0, frame0+nframes) of each of
 * the S streams, before the histogram scatter.  K = n/2+1 bins per frame.
 *   power[s][f][k] = |X_h[k]|^2          (float32)
 *   col  [s][f][k] = absolute reassigned column index (int32)
 *   row  [s][f][k] = log-frequency row, or -1 if the bin is gated/out of range
 * Host pointers.  Used by the parity tests (exact col/row, 1e-4 on power).
 */
int emspec_parity_dump(emspec_engine* e, const float* pcm, int32_t S, int64_t L,
                       int32_t n, int32_t hop, int32_t reassign,
                       int64_t frame0, int64_t nframes,
                       float* power, int32_t* col, int32_t* row);

/*
 * Parity dump of an EXACT-mode engine (EMSPEC_ERR_STATE on a fast-mode engine, and vice versa for emspec_parity_dump):
 *   power[s][f][k] = |X_h[k]|^2 (binary64), col / row as above, q[s][f][k] (optional, may be NULL) = the bin's
 *   fixed-point energy as it is added to the histogram (0 when the bin is dropped): round(power * 2^52 / (n/4)^2).
 * Host pointers.
 */
int emspec_parity_dump_exact(emspec_engine* e, const float* pcm, int32_t S, int64_t L,
                             int32_t n, int32_t hop, int32_t reassign,
                             int64_t frame0, int64_t nframes,
                             double* power, int32_t* col, int32_t* row, int64_t* q);

/* Device-pointer form of emspec_parity_dump (all five buffers on the device). */
int emspec_parity_dump_device(emspec_engine* e, const float* pcm_dev, int32_t S,
                              int64_t L, int32_t n, int32_t hop, int32_t reassign,
                              int64_t frame0, int64_t nframes,
                              float* power_dev, int32_t* col_dev, int32_t* row_dev,
                              void* hip_stream);

/*
 * ---- Multi-GPU: shard the streams, gather the finished columns (BASELINE.json north_star: "many independent
 * audio streams shard embarrassingly across the 8 GPUs of one node with a single RCCL gather over xGMI to collect
 * finished columns").  The reference has no GPU path and no collectives: [BUILD-DEFINED].
 *
 * One process (or thread) per GPU, each with its own engine.  The host splits the streams into shards (equal, or a
 * smaller one for the root, which also expands what it receives), every rank runs emspec_batch_device on its shard
 * (no exchange during compute), then all ranks call emspec_gather_columns: the palette-index columns (uint8, 1 byte
 * per cell - the float32 dB columns stay on the producing GPU) of every rank arrive on `root`, the ranks' blocks
 * [columns_r][rows] one after the other in rank order.
 *
 * Rank 0 obtains an id with emspec_comm_unique_id and hands its 128 bytes to the other ranks by whatever channel
 * the host has (Node: IPC / a file; Python: torch.distributed's store); every rank then calls emspec_comm_init
 * (collective: returns when all `world` ranks have called it).  The communicator lives as long as the engine.
 */
#define EMSPEC_COMM_ID_BYTES 128
int emspec_comm_unique_id(uint8_t* id_out /* [EMSPEC_COMM_ID_BYTES] */);
int emspec_comm_init(emspec_engine* e, const uint8_t* id /* [EMSPEC_COMM_ID_BYTES] */, int32_t rank, int32_t world);
int emspec_comm_destroy(emspec_engine* e);
/* Bound (seconds, default 120; 0 = none) on the one host wait inside emspec_gather_columns - the size exchange every rank
 * must enter.  When it expires the communicator is aborted (ncclCommAbort) and the call returns EMSPEC_ERR_COMM: a rank
 * that died does not hang its peers for ever.  May be called before emspec_comm_init. */
int emspec_comm_set_timeout(emspec_engine* e, double seconds);
int32_t emspec_comm_rank(const emspec_engine* e);    /* -1 without a communicator */
int32_t emspec_comm_world(const emspec_engine* e);   /* 0 without a communicator */

/*
 * Collective over the engine's communicator.  index_dev: this rank's finished palette-index columns
 * [columns][rows] on its device (columns = streams * columns-per-stream of the shard; ranks may differ).
 * gathered_dev (root only; ignored elsewhere): the ranks' blocks in rank order on the root's device, sum(columns_r) *
 * rows bytes; gathered_capacity = the bytes available there (the call fails with EMSPEC_ERR_INVALID_ARG on the root
 * before writing anything if the announced shards do not fit).
 * xGMI is point-to-point - each rank reaches the root over one link - so the columns travel as a lossless packed
 * image (per-column offset + bit mask of the non-zero cells + their indices; emspec_wire_* below) and are expanded on the root.
 * The call enqueues everything on hip_stream and synchronises that stream ONCE (the packed sizes differ per rank
 * and must be known on the host before the transfers can be posted): enqueue the next chunk's
 * emspec_batch_device on another stream BEFORE calling it, and the gather overlaps that compute.
 * *wire_bytes_sent (optional) receives the size of this rank's packed image (0 on the root).
 * Failure semantics: whatever can fail on one rank alone (arguments, device memory, the pack launch) is detected before the
 * size exchange, and that rank still enters the exchange with an error mark in place of its size - so ALL ranks return
 * from the same call: the failing rank with its own error, the others with EMSPEC_ERR_COMM, nothing transferred, the
 * communicator still usable.  An RCCL error, or a peer that does not show up within the timeout, aborts the communicator
 * (later calls: EMSPEC_ERR_STATE).
 * flags: EMSPEC_GATHER_LOOPBACK makes the root's own columns take the wire as well (self send/recv; exercising
 * the whole path on a single GPU).
 * EMSPEC_GATHER_PACKED: the root does NOT expand: gathered_dev receives a directory (per rank: u64 offset, u64 image bytes,
 * u64 columns, u64 0; padded to 256 B) followed by every rank's wire image (the root's own included), 256-byte aligned,
 * in rank order; capacity needed: align256(32 * world) + sum over ranks of align256(emspec_wire_bound(columns_r, rows))
 * (align256(x) = x rounded up to a multiple of 256).  Nothing is
 * lost - emspec_wire_unpack(gathered_dev + offset, bytes, columns, ...) expands any rank's block on demand (the layout
 * is also available on the host: emspec_gather_packed_layout) - and the root, which otherwise expands world-1 images
 * per gather beside its own column kernel, only receives.
 */
#define EMSPEC_GATHER_LOOPBACK 1u
#define EMSPEC_GATHER_PACKED 2u
int emspec_gather_columns(emspec_engine* e, const uint8_t* index_dev, int64_t columns, int32_t root,
                          uint8_t* gathered_dev, int64_t gathered_capacity, uint32_t flags, void* hip_stream,
                          int64_t* wire_bytes_sent);

/* Root, after an EMSPEC_GATHER_PACKED gather: where rank `rank`'s image sits in gathered_dev (any pointer may be NULL). */
int emspec_gather_packed_layout(const emspec_engine* e, int32_t rank, int64_t* offset, int64_t* bytes, int64_t* columns);

/*
 * Host-buffer convenience for a rank process (the N-API addon's computeColumnsGather): this rank's S streams
 * (pcm[S][L], host memory) -> finished columns on the device -> emspec_gather_columns -> on `root` the palette-index
 * columns of all ranks in host memory, gathered_index[world][S][columns][rows] (ignored elsewhere).  db_local
 * (optional, any rank): this rank's own float32 dB columns [S][columns][rows].  Synchronous; collective.
 */
int emspec_batch_gather(emspec_engine* e, const float* pcm, int32_t S, int64_t L, int32_t n, int32_t hop,
                        int32_t reassign, int32_t root, uint8_t* gathered_index, float* db_local,
                        int64_t* wire_bytes_sent);

/*
 * The wire image by itself, for hosts that bring their own transport: header (32 B) + a u32 payload offset per
 * column + ceil(rows/32) mask words per column + the non-zero indices (column after column, rows ascending), zero-padded
 * to a multiple of 16 B: every byte of an image is specified, so equal columns give equal images (round 5; see
 * em-spec_amd/csrc/pack.hip.inc, restated in oracle/wire_ref.py).
 * emspec_wire_bound: capacity a destination needs for `columns` columns (-1 on invalid arguments).
 * emspec_wire_pack:  index_dev [columns][rows] -> wire_dev; *wire_bytes (optional) = the image size (reading it
 *                    synchronises hip_stream; pass NULL to stay asynchronous).
 * emspec_wire_unpack: validates the header against this engine (rows) and `columns`, then expands.
 * All pointers are device pointers on the engine's device; at most 2^32 cells per call.
 */
int64_t emspec_wire_bound(int64_t columns, int32_t rows);
int emspec_wire_pack(emspec_engine* e, const uint8_t* index_dev, int64_t columns, uint8_t* wire_dev,
                     int64_t* wire_bytes, void* hip_stream);
int emspec_wire_unpack(emspec_engine* e, const uint8_t* wire_dev, int64_t wire_bytes, int64_t columns,
                       uint8_t* index_dev, void* hip_stream);

/*
 * Host-buffer batch whose palette-index columns leave the device PACKED: the same columns as emspec_batch(out->index), but
 * what crosses PCIe is the lossless wire image above (~186 B instead of 1,024 B per column on typical audio), ONE IMAGE
 * PER STREAM, tightly packed into `wire` (host memory, pinned for full speed) in stream order: stream s occupies
 * wire[offsets[s] .. offsets[s+1]) (offsets: streams + 1 entries, offsets[0] = 0; every image STARTS on a 16-byte boundary,
 * so up to 12 bytes of unspecified slack may follow an image inside its slot - an image's own size is in its header, and
 * emspec_wire_unpack_host / emspec_wire_unpack accept the slot as it is).
 * wire_capacity = streams x emspec_wire_bound(columns, rows) always suffices; EMSPEC_ERR_INVALID_ARG when the images do not
 * fit.  Runs the three-stage pipeline of emspec_batch (H2D | kernels + pack | D2H on three HIP streams); rows % 4 == 0.
 * Serves: the renderer-side batched computeColumns when the host keeps or forwards the columns compressed
 * (north_star: "host code stays in JavaScript/Node calling HIP through a thin C-ABI N-API addon").
 * emspec_wire_unpack_host: expand one image on the host's own cores (plain C, no device, no engine):
 * wire -> index_out [columns][rows]; validates the header and every offset (a damaged image is rejected, never
 * read or written out of range).
 */
int emspec_batch_packed(emspec_engine* e, const float* pcm, int32_t streams, int64_t samples_per_stream, int32_t fft_size,
                        int32_t hop, int32_t reassign, uint8_t* wire, int64_t wire_capacity, int64_t* offsets);
int emspec_wire_unpack_host(const uint8_t* wire, int64_t wire_bytes, int64_t columns, int32_t rows, uint8_t* index_out);

/* Copy out the tables the kernels use for (n): row edges in bin units
 * (rows+1 floats) and the twiddle table (n/2 complex = n floats, re,im
 * interleaved).  Either pointer may be NULL.  For table-parity tests. */
int emspec_get_tables(emspec_engine* e, int32_t n, float* edges_bins, float* twiddle);

/* 1 if emspec_batch/_device would run a fused LDS-ring kernel for this shape
 * (n = 1024, 2048, 4096 at any hop whose column ring fits in LDS; n = 8192 at
 * hop 512 or 1024; n = 16384 at any hop whose ring has at most 33 slots of 1024
 * rows, e.g. hop 512; at most 1024 rows), 0 if the generic two-kernel path
 * (per-bin records + LDS tile / walking-ring scatter).  EXACT-mode engines:
 * 1 for n = 4096, 2048 and 1024 when the u64 column ring (2 D + 2 * 4096 / n slots) fits in LDS
 * beside the 64 KB of binary64 planes - whole, or with its low rows in a per-workgroup scratch in
 * global memory when at most 6 % of a frame's bins fall there (exact_fused_lr.hip.inc; on the default
 * log axis e.g. n = 4096 / hop 256, n = 2048 / hop 128 or 256, n = 1024 / hop 128 or 256; other
 * axes at n = 4096 run exact_fused.hip.inc's kernel, which parks the ring under the planes) - else 0:
 * n = 8192 and 16384 run the two-kernel records path.  Same results either way. */
int emspec_uses_fused(const emspec_engine* e, int32_t n, int32_t hop, int32_t reassign);

/* Name of the device the engine runs on, e.g. "gfx950". */
const char* emspec_device_arch(const emspec_engine* e);

#ifdef __cplusplus
}
#endif
#endif /* EMSPEC_H */
