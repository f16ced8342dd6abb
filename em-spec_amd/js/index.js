'use strict';
/*
 * emspec — Node/Electron host side of the MI355X reassigned-spectrogram engine.
 *
 * Exposes the call the renderer already makes (BASELINE.json north_star):
 *
 *     computeSpectrogramColumn(audioFrame, fftSize, hop, reassign) -> Float32Array(rows) of dB
 *
 * plus engine management and the batched entry point for throughput runs
 * (SURVEY.md §8(b)).  All numeric work happens in libemspec (HIP, gfx950);
 * this file only marshals typed arrays through the N-API addon.  There is no
 * JS/CPU fallback: without the addon or without a gfx950 device the calls throw.
 */
const native = require('./emspec.node');

class Engine {
  /** config: {device, rows, sampleRate, fminHz, fmaxHz, gain, dbTop, dbRange, gateDb, powerFloor, exact, streams}
   *  exact: true = EMSPEC_MODE_EXACT (binary64 arithmetic, 64-bit fixed-point histogram; include/emspec.h)
   *  streams: S > 1 = a live multi-stream engine: computeSpectrogramColumns / pushSamplesMulti advance all S streams per
   *  call in ONE kernel launch; its frame / column blocks are page-locked (engine.frames, engine.columnsDb, ...), so the
   *  kernel reads and writes them in place */
  constructor(config = {}) {
    this.rows = native.rows(config);
    this._h = native.create(config);
    this._db = new Float32Array(this.rows);
    this.streams = Math.max(1, config.streams | 0);
    this.columnIndex = new Float64Array(this.streams);   // per stream: index of the column the last call returned (-1: empty)
  }

  /** Page-locked Float32Array / Uint8Array views for the live calls, (re)made when the shape changes. */
  _liveBlocks(fftSize, wantRgba) {
    const S = this.streams, R = this.rows;
    if (fftSize > 0 && (!this.frames || this.frames.length !== S * fftSize)) this.frames = new Float32Array(native.allocPinned(4 * S * fftSize));
    if (!this.columnsDb) this.columnsDb = new Float32Array(native.allocPinned(4 * S * R));
    if (wantRgba && !this.columnsRgba) this.columnsRgba = new Uint8Array(native.allocPinned(4 * S * R));
  }

  /**
   * The live multi-stream form of computeSpectrogramColumn: one frame of each of the engine's S streams in, one finished
   * column of each out, ONE launch (emspec_columns).  frames: Float32Array(S * fftSize), stream after stream - pass
   * engine.frames (page-locked, filled by the caller) to avoid a staging copy; any Float32Array works.
   * Returns engine.columnsDb: Float32Array(S * rows) of dB (page-locked, overwritten by the next call); with wantRgba,
   * engine.columnsRgba holds the colours.  engine.columnIndex[s] = index of stream s's column, -1 while its ring primes.
   */
  computeSpectrogramColumns(frames, fftSize, hop, reassign = true, wantRgba = false) {
    this._liveBlocks(fftSize, wantRgba);
    native.columns(this._h, frames, this.streams, fftSize, hop, !!reassign, this.columnsDb, wantRgba ? this.columnsRgba : undefined,
      this.columnIndex);
    return this.columnsDb;
  }

  /** computeSpectrogramColumns off the JS thread (libuv pool): resolves with engine.columnsDb.  Do not touch the blocks or
   *  this engine until the promise settles. */
  computeSpectrogramColumnsAsync(frames, fftSize, hop, reassign = true, wantRgba = false) {
    this._liveBlocks(fftSize, wantRgba);
    return native.columnsAsync(this._h, frames, this.streams, fftSize, hop, !!reassign, this.columnsDb,
      wantRgba ? this.columnsRgba : undefined, this.columnIndex).then(() => this.columnsDb);
  }

  /** Every stream that still has pending columns emits its next one (others: the empty column, columnIndex -1);
   *  throws EMSPEC_ERR_STATE when no stream has any.  Returns engine.columnsDb. */
  flushColumns(wantRgba = false) {
    this._liveBlocks(0, wantRgba);   // (the column blocks only: a session fed by sample blocks never made a frame block)
    native.columnsFlush(this._h, this.columnsDb, wantRgba ? this.columnsRgba : undefined, this.columnIndex);
    return this.columnsDb;
  }

  /**
   * Live streaming by sample blocks for all S streams (emspec_push_samples_multi): samples = Float32Array(S * count),
   * `count` new samples of every stream, stream after stream (engine.sampleBlock(count) is a page-locked one).
   * Returns { maxColumns, counts, first, db, rgba }: stream s completed counts[s] columns,
   * db.subarray((s * maxColumns + i) * rows, ...) is its i-th, first[s] the absolute index of its first (-1 if none).
   * db / rgba / counts / first are engine-owned page-locked blocks, overwritten by the next call (copy what you keep).
   * Drain the pending columns with flushColumns().
   */
  pushSamplesMulti(samples, fftSize, hop, reassign = true, wantRgba = false) {
    const S = this.streams, count = samples.length / S;
    const maxColumns = native.pushColumnsMulti(this._h, count, fftSize, hop, !!reassign);
    let o = this._push;
    if (!o || o.maxColumns < maxColumns || (wantRgba && !o.rgbaAll)) {
      const cap = Math.max(maxColumns, 1);
      o = this._push = { maxColumns: cap, dbAll: new Float32Array(native.allocPinned(4 * S * cap * this.rows)),
        rgbaAll: wantRgba ? new Uint8Array(native.allocPinned(4 * S * cap * this.rows)) : undefined,
        counts: new Float64Array(S), first: new Float64Array(S) };
    }
    native.pushMulti(this._h, samples, S, fftSize, hop, !!reassign, o.maxColumns, o.dbAll, wantRgba ? o.rgbaAll : undefined, o.counts, o.first);
    return { maxColumns: o.maxColumns, counts: o.counts, first: o.first, db: o.dbAll, rgba: wantRgba ? o.rgbaAll : undefined };
  }

  /** A page-locked Float32Array(S * count) for pushSamplesMulti (kept per count). */
  sampleBlock(count) {
    if (!this._blk || this._blk.length !== this.streams * count) this._blk = new Float32Array(native.allocPinned(4 * this.streams * count));
    return this._blk;
  }

  /** Restart stream s of the live session (its position, pending columns, display state); the others continue. */
  resetStream(s) { native.resetStream(this._h, s); }

  /**
   * One frame in, one finished column out.  With time reassignment on, the column
   * returned for call j is column j - latencyColumns(fftSize, hop) (energy can move
   * that many columns either way); the first calls return the empty column and set
   * engine.lastColumn = -1.  Returns a Float32Array(rows) of dB owned by the caller.
   */
  computeSpectrogramColumn(audioFrame, fftSize, hop, reassign = true, outRgba = undefined) {
    const out = new Float32Array(this.rows);
    this.lastColumn = native.column(this._h, audioFrame, fftSize, hop, !!reassign, out, outRgba);
    return out;
  }

  /**
   * Streaming by sample blocks (what an audio callback delivers): feed any number of new samples; every
   * `hop` of them completes a frame.  Returns { first, count, db, rgba }: `count` finished columns,
   * oldest first, `first` = absolute index of the first (-1 when count is 0), db = Float32Array(count*rows),
   * rgba = Uint8Array(4*count*rows) when wantRgba.  Drain the last D columns with flush().
   */
  pushSamples(samples, fftSize, hop, reassign = true, wantRgba = false) {
    const count = native.pushColumns(this._h, samples.length, fftSize, hop, !!reassign);
    const db = new Float32Array(count * this.rows);
    const rgba = wantRgba ? new Uint8Array(4 * count * this.rows) : undefined;
    const first = native.push(this._h, samples, fftSize, hop, !!reassign, this.rows, db, rgba);
    if (count > 0) this.lastColumn = first + count - 1;
    return { first, count, db, rgba };
  }

  /** Emit one of the columns still pending after the last frame; throws EMSPEC_ERR_STATE when none. */
  flush(outRgba = undefined) {
    const out = new Float32Array(this.rows);
    this.lastColumn = native.flush(this._h, out, outRgba);
    return out;
  }

  /**
   * Batched: pcm = Float32Array(S*L) (S streams of L samples, row-major) ->
   * out.db Float32Array(S*C*rows) and/or out.rgba Uint8Array(4*S*C*rows) and/or out.index Uint8Array(S*C*rows).
   * Returns C, the columns per stream.
   */
  computeColumns(pcm, S, L, fftSize, hop, reassign, out) {
    return native.batch(this._h, pcm, S, L, fftSize, hop, !!reassign, out.db, out.rgba, out.index);
  }

  /**
   * Throughput entry with the palette-index columns kept compressed across PCIe (emspec_batch_packed): one lossless wire
   * image per stream, ~186 B instead of 1,024 B per column on typical audio.  wire: Uint8Array (allocPinned for full
   * speed; S * wireBound(C, rows) always suffices), offsets: Float64Array(S + 1).  Stream s is
   * wire.subarray(offsets[s], offsets[s + 1]); expand it on the host with unpackWire(image, C, rows, out) or keep /
   * forward it as it is.  Returns C, the columns per stream.
   */
  computeColumnsPacked(pcm, S, L, fftSize, hop, reassign, wire, offsets) {
    return native.batchPacked(this._h, pcm, S, L, fftSize, hop, !!reassign, wire, offsets);
  }

  /** computeColumnsPacked off the JS thread (libuv pool): resolves with C; offsets are filled when it settles. */
  computeColumnsPackedAsync(pcm, S, L, fftSize, hop, reassign, wire, offsets) {
    return native.batchPackedAsync(this._h, pcm, S, L, fftSize, hop, !!reassign, wire, offsets);
  }

  /** Same as computeColumns, off the JS thread: resolves with C.  Do not touch the arrays or this
   *  engine until the promise settles (an engine is not thread-safe). */
  computeColumnsAsync(pcm, S, L, fftSize, hop, reassign, out) {
    return native.batchAsync(this._h, pcm, S, L, fftSize, hop, !!reassign, out.db, out.rgba, out.index);
  }

  /** Synchronise the device and throw if a kernel flagged a protocol error since the last check (emspec_device_status). */
  deviceStatus() { native.deviceStatus(this._h); }

  /** Multi-GPU (one node process per GPU): join the gather communicator.  id = commUniqueId() of rank 0, handed over by
   *  the host's own channel (IPC, a file); collective over all `world` rank processes. */
  commInit(id, rank, world) { native.commInit(this._h, id, rank, world); this.commRank = rank; this.commWorld = world; }

  /** This rank's shard of the streams -> finished columns, gathered on `root` over RCCL (packed on the wire).
   *  out.allIndex (root): Uint8Array(world*S*C*rows), rank-major; out.db (optional): this rank's own dB columns.
   *  Returns the bytes this rank put on the wire.  Collective: every rank process calls it. */
  computeColumnsGather(pcm, S, L, fftSize, hop, reassign, root, out) {
    return native.batchGather(this._h, pcm, S, L, fftSize, hop, !!reassign, root, out.allIndex, out.db);
  }

  setColormap(rgba256) { native.setColormap(this._h, rgba256); }

  /** Temporal smoothing (0..0.95) and adaptive brightness / AGC strength (0..1); 0,0 = off. */
  setDisplay(smoothing = 0, agcStrength = 0) { native.setDisplay(this._h, smoothing, agcStrength); }

  /** Install any strictly increasing frequency axis: Float32Array(rows+1) of edges in Hz; null = log axis. */
  setRowEdges(edgesHz) { native.setRowEdges(this._h, edgesHz); }
  /** The axis in use (rows+1 edges in Hz): the inverse map for the shift+hover frequency read-out. */
  getRowEdges() { const e = new Float32Array(this.rows + 1); native.getRowEdges(this._h, e); return e; }
  /** Frequency (Hz) at fractional row y, geometric within the row. */
  rowToHz(y) {
    const e = this.getRowEdges();
    const r = Math.min(this.rows - 1, Math.max(0, Math.floor(y)));
    return e[r] * Math.pow(e[r + 1] / e[r], Math.min(1, Math.max(0, y - r)));
  }
  reset() { native.reset(this._h); }
  destroy() { if (this._h) { native.destroy(this._h); this._h = null; } }
}

/**
 * One [BUILD-DEFINED] law for the reference's "Frequency Scale" (zoom) and "Low-End Boost"
 * sliders (their real laws are undocumented): the axis spans fmin .. fmin*(fmax/fmin)^(1/freqScale)
 * and row r sits at u = (r/rows)^lowEndBoost along the log range, so lowEndBoost > 1 gives the
 * low end more rows.  Returns Float32Array(rows+1) for Engine#setRowEdges.
 */
function warpedEdges(rows, fminHz, fmaxHz, lowEndBoost = 1, freqScale = 1) {
  return native.warpedEdges(rows, fminHz, fmaxHz, lowEndBoost, freqScale);   // one implementation for every host: emspec_warped_edges_hz
}

/** The reference's colour ramp (5-stop gradient measured from its screenshot) with a brightness factor: Uint8Array(1024). */
function makeColormap(brightness = 0.5, stops = undefined) {
  if (stops === undefined) return native.referenceColormap(brightness);   // emspec_make_colormap, shared with the other bindings
  const lut = new Uint8Array(1024);
  const n = stops.length - 1;
  for (let i = 0; i < 256; i++) {
    const v = Math.min(1, (i / 255) * (brightness / 0.5));
    const t = v * n, s = Math.min(n - 1, Math.floor(t)), f = t - s;
    for (let c = 0; c < 3; c++) lut[4 * i + c] = Math.round(stops[s][c] + f * (stops[s + 1][c] - stops[s][c]));
    lut[4 * i + 3] = 255;
  }
  return lut;
}

/** Gradient stop tables for makeColormap().  'reference' is the ramp measured from the reference's
 *  settings screenshot (SURVEY.md §4); the others are plain conveniences, not claims about the
 *  reference's other (undocumented) maps. */
const colormapStops = {
  reference: [[0, 0, 0], [80, 0, 80], [200, 50, 50], [255, 150, 0], [255, 255, 200]],
  grayscale: [[0, 0, 0], [255, 255, 255]],
  ice: [[0, 0, 0], [0, 40, 120], [0, 140, 200], [140, 230, 255], [255, 255, 255]],
  green: [[0, 0, 0], [0, 70, 20], [40, 180, 60], [200, 255, 140], [255, 255, 255]],
};

let defaultEngine = null;
let defaultMulti = null;

/** Drop-in for a renderer that draws S streams: frames = Float32Array(S * fftSize) -> Float32Array(S * rows) of dB
 *  (one launch for all streams).  Lazily creates one engine per stream count. */
function computeSpectrogramColumns(frames, fftSize, hop, reassign = true) {
  const S = frames.length / fftSize;
  if (!defaultMulti || defaultMulti.streams !== S) {
    if (defaultMulti) defaultMulti.destroy();
    defaultMulti = new Engine({ streams: S });
  }
  return defaultMulti.computeSpectrogramColumns(frames, fftSize, hop, reassign);
}

/** Drop-in for the renderer: lazily creates one engine with the default configuration. */
function computeSpectrogramColumn(audioFrame, fftSize, hop, reassign = true) {
  if (!defaultEngine) defaultEngine = new Engine();
  return defaultEngine.computeSpectrogramColumn(audioFrame, fftSize, hop, reassign);
}

module.exports = {
  Engine,
  createEngine: (config) => new Engine(config),
  computeSpectrogramColumn,
  computeSpectrogramColumns,
  /** ArrayBuffer of page-locked host memory: typed arrays over it move at full PCIe speed. */
  allocPinned: native.allocPinned,
  warpedEdges,
  makeColormap,
  colormapStops,
  commUniqueId: native.commUniqueId,
  numColumns: native.numColumns,
  /** Bytes that always hold the wire image of `columns` columns of `rows` rows (emspec_wire_bound). */
  wireBound: native.wireBound,
  /** Expand one wire image (Uint8Array) into out: Uint8Array(columns * rows) on the host's own cores - no device, no engine. */
  unpackWire: native.wireUnpack,
  latencyColumns: native.latencyColumns,
  /** 'emspec abi=2 sources=<sha16> arch=gfx950': what the loaded libemspec was built from. */
  buildInfo: native.buildInfo,
};
