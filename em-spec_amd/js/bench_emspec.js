'use strict';
/*
 * Node-driven throughput figures of the host side north_star names ("host code stays in JavaScript/Node calling HIP through
 * a thin C-ABI N-API addon"): the batched entries on BASELINE configs[2] from page-locked buffers (allocPinned) -
 * computeColumnsAsync (palette index out) and computeColumnsPackedAsync - and the live multi-stream calls (S streams, one
 * hop per call).  bench.py runs this beside its ctypes figures and reports the result under configs["node host ..."].
 *
 *   node bench_emspec.js <pcm.f32 | -> <S> <L> <fftSize> <hop> [runs] [liveCalls]
 *
 * pcm.f32: S*L float32 samples, stream after stream (bench.py writes its synthetic batch there so that both hosts time the
 * same input); "-" = a cheap built-in signal.  Prints ONE JSON line.
 */
const fs = require('fs');
const em = require('./index.js');

function median(a) { const b = Array.from(a).sort((x, y) => x - y); return b[b.length >> 1]; }
function now() { const t = process.hrtime(); return t[0] + t[1] * 1e-9; }

function fill(pcm, S, L, file) {
  if (file !== '-') {
    const fd = fs.openSync(file, 'r');
    const bytes = Buffer.from(pcm.buffer, pcm.byteOffset, pcm.byteLength);
    let at = 0;
    while (at < bytes.length) {
      const got = fs.readSync(fd, bytes, at, Math.min(1 << 28, bytes.length - at), at);
      if (got <= 0) throw new Error('short read of ' + file);
      at += got;
    }
    fs.closeSync(fd);
    return;
  }
  let s = 12345;
  for (let i = 0; i < L; i++) {
    s = (Math.imul(s, 1103515245) + 12345) | 0;
    pcm[i] = 0.5 * Math.sin(2 * Math.PI * 440 * i / 48000) + 0.25 * Math.sin(2 * Math.PI * (1000 + 2000 * i / 48000) * i / 48000) +
      ((s >>> 8) / 16777216 - 0.5) * 2e-3;
  }
  for (let k = 1; k < S; k++) pcm.copyWithin(k * L, 0, L);
}

async function main() {
  const [file, S, L, n, hop] = [process.argv[2] || '-', +process.argv[3] || 8, +process.argv[4] || (1 << 20), +process.argv[5] || 4096, +process.argv[6] || 256];
  const runs = +process.argv[7] || 5, liveCalls = +process.argv[8] || 2000;
  const out = { node: process.version, streams: S, samples_per_stream: L, fft: n, hop };
  const C = em.numColumns(L, n, hop);
  const pcm = new Float32Array(em.allocPinned(4 * S * L));
  fill(pcm, S, L, file);
  for (const exact of [false, true]) {
    const eng = em.createEngine({ exact });
    const R = eng.rows;
    const index = new Uint8Array(em.allocPinned(S * C * R));
    const wire = new Uint8Array(em.allocPinned(S * em.wireBound(C, R)));
    const offs = new Float64Array(S + 1);
    const key = exact ? 'exact' : 'fast';
    out[key] = {};
    await eng.computeColumnsAsync(pcm, S, L, n, hop, true, { index });              // warm-up (allocations, clocks)
    let ts = [];
    for (let r = 0; r < runs; r++) { const t0 = now(); await eng.computeColumnsAsync(pcm, S, L, n, hop, true, { index }); ts.push(now() - t0); }
    out[key].index_out = { columns_per_s: S * C / median(ts), ms: median(ts) * 1e3, runs: ts.map((t) => +(t * 1e3).toFixed(2)) };
    // the same call from ORDINARY typed arrays (what a caller that knows nothing of allocPinned writes), reused across calls
    if (!exact) {
      const plainPcm = new Float32Array(pcm), plainIndex = new Uint8Array(S * C * R);
      await eng.computeColumnsAsync(plainPcm, S, L, n, hop, true, { index: plainIndex });
      ts = [];
      for (let r = 0; r < runs; r++) { const t0 = now(); await eng.computeColumnsAsync(plainPcm, S, L, n, hop, true, { index: plainIndex }); ts.push(now() - t0); }
      let same = true;
      for (let i = 0; i < plainIndex.length && same; i += 4099) same = Math.abs(plainIndex[i] - index[i]) <= 1;
      out[key].index_out_plain_arrays = { columns_per_s: S * C / median(ts), ms: median(ts) * 1e3, runs: ts.map((t) => +(t * 1e3).toFixed(2)), sampled_cells_equal_to_pinned: same };
    }
    await eng.computeColumnsPackedAsync(pcm, S, L, n, hop, true, wire, offs);
    ts = [];
    for (let r = 0; r < runs; r++) { const t0 = now(); await eng.computeColumnsPackedAsync(pcm, S, L, n, hop, true, wire, offs); ts.push(now() - t0); }
    out[key].packed = { columns_per_s: S * C / median(ts), ms: median(ts) * 1e3, wire_bytes_per_column: offs[S] / (S * C), runs: ts.map((t) => +(t * 1e3).toFixed(2)) };
    eng.destroy();
    // live: S streams, one hop per call (page-locked blocks; the per-call time is what the JS thread spends in the call)
    const live = em.createEngine({ exact, streams: S });
    const calls = Math.min(liveCalls, Math.floor((L - n) / hop));
    const blk = live.sampleBlock(hop);
    const prime = new Float32Array(S * (n - hop));
    for (let s = 0; s < S; s++) prime.set(pcm.subarray(s * L, s * L + n - hop), s * (n - hop));
    live.pushSamplesMulti(prime, n, hop, true);
    ts = [];
    let cols = 0;
    for (let i = 0; i < calls; i++) {
      const at = n - hop + i * hop;
      for (let s = 0; s < S; s++) blk.set(pcm.subarray(s * L + at, s * L + at + hop), s * hop);    // the audio callback's copy: not timed
      const t0 = now();
      const r = live.pushSamplesMulti(blk, n, hop, true);
      ts.push(now() - t0);
      cols += r.counts[0];
    }
    ts = ts.slice(ts.length / 10 | 0);
    const med = median(ts);
    out[key].live_push = { columns_per_s: S / med, us_per_call: med * 1e6, p90_us: ts.sort((a, b) => a - b)[(ts.length * 0.9) | 0] * 1e6, calls, columns_stream0: cols };
    live.reset();
    live._liveBlocks(n, false);
    ts = [];
    for (let i = 0; i < calls; i++) {
      for (let s = 0; s < S; s++) live.frames.set(pcm.subarray(s * L + i * hop, s * L + i * hop + n), s * n);   // the renderer's own frames: not timed
      const t0 = now();
      live.computeSpectrogramColumns(live.frames, n, hop, true);
      ts.push(now() - t0);
    }
    ts = ts.slice(ts.length / 10 | 0);
    const medf = median(ts);
    out[key].live_frames = { columns_per_s: S / medf, us_per_call: medf * 1e6, p90_us: ts.sort((a, b) => a - b)[(ts.length * 0.9) | 0] * 1e6, calls };
    live.destroy();
  }
  console.log(JSON.stringify(out));
}

main().catch((e) => { console.error(e); process.exit(1); });
