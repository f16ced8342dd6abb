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
  /** config: {device, rows, sampleRate, fminHz, fmaxHz, gain, dbTop, dbRange, gateDb, powerFloor} */
  constructor(config = {}) {
    this.rows = native.rows(config);
    this._h = native.create(config);
    this._db = new Float32Array(this.rows);
  }

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

  setColormap(rgba256) { native.setColormap(this._h, rgba256); }
  reset() { native.reset(this._h); }
  destroy() { if (this._h) { native.destroy(this._h); this._h = null; } }
}

let defaultEngine = null;

/** Drop-in for the renderer: lazily creates one engine with the default configuration. */
function computeSpectrogramColumn(audioFrame, fftSize, hop, reassign = true) {
  if (!defaultEngine) defaultEngine = new Engine();
  return defaultEngine.computeSpectrogramColumn(audioFrame, fftSize, hop, reassign);
}

module.exports = {
  Engine,
  createEngine: (config) => new Engine(config),
  computeSpectrogramColumn,
  numColumns: native.numColumns,
  latencyColumns: native.latencyColumns,
};
