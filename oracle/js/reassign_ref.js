'use strict';
/*
 * reassign_ref.js — plain-JavaScript restatement of the three-window reassigned spectrogram.
 *
 * TEST INFRASTRUCTURE ONLY (oracle/): never loaded by the product (em-spec_amd/js).
 * PARITY UNPINNED: the reference's JS source is private (/root/reference/README.md:73); this is
 * NOT the reference's code.  It restates the published method (SURVEY.md §8a: Hann, time-ramp and
 * derivative windows; t-hat, f-hat; log-frequency rows; dB) in ordinary double-precision JS with its
 * own radix-2 FFT, and serves two purposes:
 *   check  — a third independent implementation for the golden vectors (tests/test_oracle.py);
 *   bench  — the "plain JS path" CPU baseline proxy of BASELINE.md (single thread, node).
 *
 * usage:
 *   node reassign_ref.js check <pcm.f32> <n> <hop> <frame0> <nframes> <reassign 0|1> <out prefix>
 *   node reassign_ref.js bench <n> <hop> <seconds>
 */
const fs = require('fs');

const twCache = new Map();
function twiddles(n) {        // cos/sin table, as any JS FFT library keeps
  let t = twCache.get(n);
  if (!t) {
    t = { c: new Float64Array(n / 2), s: new Float64Array(n / 2) };
    for (let k = 0; k < n / 2; k++) { t.c[k] = Math.cos(-2 * Math.PI * k / n); t.s[k] = Math.sin(-2 * Math.PI * k / n); }
    twCache.set(n, t);
  }
  return t;
}

function fftInPlace(re, im) {
  const n = re.length;
  const tw = twiddles(n);
  for (let i = 1, j = 0; i < n; i++) {
    let bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) { let t = re[i]; re[i] = re[j]; re[j] = t; t = im[i]; im[i] = im[j]; im[j] = t; }
  }
  for (let len = 2; len <= n; len <<= 1) {
    const half = len >> 1, step = n / len;
    for (let i = 0; i < n; i += len) {
      for (let k = 0; k < half; k++) {
        const wr = tw.c[k * step], wi = tw.s[k * step];
        const xr = re[i + k + half], xi = im[i + k + half];
        const tr = xr * wr - xi * wi, ti = xr * wi + xi * wr;
        re[i + k + half] = re[i + k] - tr; im[i + k + half] = im[i + k] - ti;
        re[i + k] += tr; im[i + k] += ti;
      }
    }
  }
}

function makePlan(n, hop, opts) {
  const o = Object.assign({ rows: 1024, fs: 48000, fmin: 20, fmax: 24000, powerFloor: 1e-14, reassign: true }, opts || {});
  const h = new Float64Array(n), th = new Float64Array(n), dh = new Float64Array(n);
  for (let i = 0; i < n; i++) {
    h[i] = 0.5 - 0.5 * Math.cos(2 * Math.PI * i / n);
    th[i] = (i - n / 2) * h[i];
    dh[i] = (Math.PI / n) * Math.sin(2 * Math.PI * i / n);
  }
  const edges = new Float64Array(o.rows + 1);
  for (let r = 0; r <= o.rows; r++) edges[r] = o.fmin * Math.pow(o.fmax / o.fmin, r / o.rows) * n / o.fs;
  // twiddles for the benchmark FFT (the check path recomputes them; speed is irrelevant there)
  return { n, hop, K: n / 2 + 1, D: o.reassign ? Math.ceil(n / (2 * hop)) : 0, h, th, dh, edges, o,
           pfloor: o.powerFloor * (n / 4) * (n / 4) };
}

function rowOf(edges, rows, kh) {
  if (!(kh >= edges[0]) || !(kh < edges[rows])) return -1;
  let lo = 0, hi = rows;
  while (hi - lo > 1) { const mid = (lo + hi) >> 1; if (edges[mid] <= kh) lo = mid; else hi = mid; }
  return lo;
}

/* one frame -> per-bin power, t-hat (samples), k-hat (bins), col, row */
function frame(plan, pcm, j, out, f) {
  const { n, hop, K, D, h, th, dh, edges, o, pfloor } = plan;
  if (!plan.a) { plan.a = [0, 1, 2].map(() => new Float64Array(n)); plan.b = [0, 1, 2].map(() => new Float64Array(n)); }
  const a = plan.a, b = plan.b;
  for (let w = 0; w < 3; w++) b[w].fill(0);
  for (let i = 0; i < n; i++) { const x = pcm[j * hop + i]; a[0][i] = x * h[i]; a[1][i] = x * th[i]; a[2][i] = x * dh[i]; }
  for (let w = 0; w < 3; w++) fftInPlace(a[w], b[w]);
  for (let k = 0; k < K; k++) {
    const hr = a[0][k], hi = b[0][k];
    const P = hr * hr + hi * hi;
    let that = j * hop + n / 2, khat = k, col = j, row = -1;
    if (P >= pfloor && P > 0) {
      if (o.reassign) {
        const ts = (a[1][k] * hr + b[1][k] * hi) / P;
        const ks = -(n / (2 * Math.PI)) * (b[2][k] * hr - a[2][k] * hi) / P;
        that += ts; khat += ks;
        const cf = Math.floor(ts / hop + 0.5);
        if (Math.abs(cf) <= D) { col = j + cf; row = rowOf(edges, o.rows, khat); }
      } else {
        row = rowOf(edges, o.rows, khat);
      }
    }
    const q = f * K + k;
    out.power[q] = P; out.that[q] = that; out.khat[q] = khat; out.col[q] = col; out.row[q] = row;
  }
}

function check(argv) {
  const [file, n, hop, frame0, nframes, reassign, prefix] = [argv[0], +argv[1], +argv[2], +argv[3], +argv[4], +argv[5], argv[6]];
  const buf = fs.readFileSync(file);
  const pcm = new Float32Array(buf.buffer, buf.byteOffset, buf.length / 4);
  const plan = makePlan(n, hop, { reassign: !!reassign });
  const K = plan.K;
  const out = { power: new Float64Array(nframes * K), that: new Float64Array(nframes * K), khat: new Float64Array(nframes * K),
                col: new Int32Array(nframes * K), row: new Int32Array(nframes * K) };
  for (let f = 0; f < nframes; f++) frame(plan, pcm, frame0 + f, out, f);
  for (const key of Object.keys(out)) fs.writeFileSync(prefix + '.' + key, Buffer.from(out[key].buffer));
}

function bench(argv) {
  const n = +argv[0], hop = +argv[1], seconds = +argv[2];
  const plan = makePlan(n, hop, { reassign: true });
  const frames = 64, L = n + hop * (frames - 1), K = plan.K, R = plan.o.rows;
  const pcm = new Float32Array(L);
  let s = 12345;
  for (let i = 0; i < L; i++) { s = (Math.imul(s, 1103515245) + 12345) | 0; pcm[i] = 0.4 * Math.sin(2 * Math.PI * 997 * i / 48000) + ((s >>> 8) / 16777216 - 0.5) * 1e-2; }
  const out = { power: new Float64Array(K), that: new Float64Array(K), khat: new Float64Array(K), col: new Int32Array(K), row: new Int32Array(K) };
  const hist = new Float64Array(frames * R), db = new Float32Array(frames * R);
  const scale = 32 / (3 * n * n);
  let cols = 0;
  const t0 = process.hrtime.bigint();
  let dt = 0;
  do {
    hist.fill(0);
    for (let j = 0; j < frames; j++) {
      frame(plan, pcm, j, out, 0);
      for (let k = 0; k < K; k++) { const c = out.col[k], r = out.row[k]; if (r >= 0 && c >= 0 && c < frames) hist[c * R + r] += out.power[k]; }
    }
    for (let i = 0; i < hist.length; i++) db[i] = 10 * Math.log10(hist[i] * scale + 1e-20);
    cols += frames;
    dt = Number(process.hrtime.bigint() - t0) / 1e9;
  } while (dt < seconds);
  console.log(JSON.stringify({ columns_per_s: cols / dt, columns: cols, seconds: dt, n, hop, node: process.version }));
}

/* finished dB columns of one stream, float64 throughout (histogram scatter, 10 log10): what a plain-JS renderer
 * would draw.  Returns Float64Array(frames * rows). */
function columnsDb(pcm, n, hop, reassign, frames, opts) {
  const plan = makePlan(n, hop, Object.assign({ reassign: !!reassign }, opts || {}));
  const K = plan.K, R = plan.o.rows;
  const out = { power: new Float64Array(K), that: new Float64Array(K), khat: new Float64Array(K), col: new Int32Array(K), row: new Int32Array(K) };
  const hist = new Float64Array(frames * R), db = new Float64Array(frames * R);
  for (let j = 0; j < frames; j++) {
    frame(plan, pcm, j, out, 0);
    for (let k = 0; k < K; k++) { const c = out.col[k], r = out.row[k]; if (r >= 0 && c >= 0 && c < frames) hist[c * R + r] += out.power[k]; }
  }
  const scale = 32 / (3 * n * n);
  for (let i = 0; i < hist.length; i++) db[i] = 10 * Math.log10(hist[i] * scale + 1e-20);
  return db;
}

if (require.main === module) {
  const mode = process.argv[2];
  if (mode === 'check') check(process.argv.slice(3));
  else if (mode === 'bench') bench(process.argv.slice(3));
  else { console.error('usage: node reassign_ref.js check|bench ...'); process.exit(2); }
} else {
  module.exports = { makePlan, frame, columnsDb };
}
