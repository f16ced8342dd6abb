'use strict';
/* Node smoke test of the drop-in call on a GPU box: BASELINE config 1 (FFT 1024, hop 256,
 * reassignment off) frame by frame, and config 2 shape (4096/256, reassignment on) against
 * the batched entry point.  Exits non-zero on any mismatch.  Run: node test_emspec.js */
const path = require('path');
const em = require('./index.js');
// the plain-JS float64 restatement (TEST INFRASTRUCTURE: oracle/js/reassign_ref.js) - the checker, never the product
const ref = require(path.join(__dirname, '..', '..', 'oracle', 'js', 'reassign_ref.js'));

/* The addon against the plain-JS float64 three-window method (an implementation that shares nothing with the HIP
 * kernels): streaming computeSpectrogramColumn + flush, and the batched call.  float32 and float64 disagree on a bin
 * that sits on a cell edge and cells near the display floor carry float32 rounding noise, so the bound is statistical:
 * of the cells the float64 method puts above -60 dB, >= 99 % within 0.01 dB, median error < 1e-4 dB. */
function checkAgainstJsOracle(fftSize, hop, reassign, frames) {
  const eng = em.createEngine({});
  const L = fftSize + hop * (frames - 1);
  const pcm = synth(L);
  const R = eng.rows;
  const want = ref.columnsDb(pcm, fftSize, hop, reassign, frames);
  const D = em.latencyColumns(fftSize, hop, reassign);
  const got = new Float32Array(frames * R);
  for (let j = 0; j < frames; j++) {
    const col = eng.computeSpectrogramColumn(pcm.subarray(j * hop, j * hop + fftSize), fftSize, hop, reassign);
    if (eng.lastColumn >= 0) got.set(col, eng.lastColumn * R);
  }
  for (let k = 0; k < Math.min(D, frames); k++) { const col = eng.flush(); got.set(col, eng.lastColumn * R); }
  const batch = new Float32Array(frames * R);
  eng.computeColumns(pcm, 1, L, fftSize, hop, reassign, { db: batch });
  for (const [name, arr] of [['streaming', got], ['batch', batch]]) {
    const errs = [];
    for (let i = 0; i < want.length; i++) if (want[i] > -60) errs.push(Math.abs(arr[i] - want[i]));
    errs.sort((a, b) => a - b);
    const within = errs.filter((e) => e < 1e-2).length / errs.length, med = errs[errs.length >> 1];
    if (!(errs.length > 100 && within > 0.99 && med < 1e-4))
      throw new Error(name + ' vs plain-JS float64 oracle: ' + errs.length + ' strong cells, ' + within + ' within 0.01 dB, median ' + med);
  }
  eng.destroy();
}

/* EXACT mode ({exact: true}: binary64 + 64-bit fixed-point histogram) against the plain-JS float64 method: every cell the
 * float64 method puts above -60 dB within 8.7e-4 dB (= 1e-4 relative on magnitude; no bin changes cell), and the streaming
 * call gives the SAME float32 bits as the batched call (integer accumulation is order-independent). */
function checkExactAgainstJsOracle(fftSize, hop, frames) {
  const eng = em.createEngine({ exact: true });
  const L = fftSize + hop * (frames - 1);
  const pcm = synth(L);
  const R = eng.rows;
  const want = ref.columnsDb(pcm, fftSize, hop, true, frames);
  const D = em.latencyColumns(fftSize, hop, true);
  const got = new Float32Array(frames * R);
  for (let j = 0; j < frames; j++) {
    const col = eng.computeSpectrogramColumn(pcm.subarray(j * hop, j * hop + fftSize), fftSize, hop, true);
    if (eng.lastColumn >= 0) got.set(col, eng.lastColumn * R);
  }
  for (let k = 0; k < Math.min(D, frames); k++) { const col = eng.flush(); got.set(col, eng.lastColumn * R); }
  const batch = new Float32Array(frames * R);
  eng.computeColumns(pcm, 1, L, fftSize, hop, true, { db: batch });
  let strong = 0, worst = 0;
  for (let i = 0; i < want.length; i++) {
    if (got[i] !== batch[i]) throw new Error('exact mode: streaming and batch differ at cell ' + i);
    if (want[i] > -60) { strong++; worst = Math.max(worst, Math.abs(batch[i] - want[i])); }
  }
  if (!(strong > 100 && worst < 8.7e-4)) throw new Error('exact mode vs plain-JS float64 oracle: ' + strong + ' strong cells, worst ' + worst + ' dB');
  eng.deviceStatus();      // ABI 2: synchronise and throw if a kernel flagged a protocol error (the fused kernels' bounded waits)
  eng.destroy();
  return worst;
}

/* ABI 2: what the loaded libemspec was built from */
/* the packed throughput entry: one wire image per stream over PCIe, expanded on the host's own cores */
function checkPacked() {
  const eng = em.createEngine({ exact: true });      // EXACT: the two calls give the same bytes
  const fftSize = 4096, hop = 256, frames = 60, S = 5, L = fftSize + hop * (frames - 1);
  const R = eng.rows;
  const pcm = new Float32Array(em.allocPinned(S * L * 4));
  for (let s = 0; s < S; s++) pcm.set(synth(L).map((v, i) => v * (1 - 0.1 * s) * ((i + s) % 5 ? 1 : 0.7)), s * L);
  const idx = new Uint8Array(em.allocPinned(S * frames * R));
  eng.computeColumns(pcm, S, L, fftSize, hop, true, { index: idx });
  const wire = new Uint8Array(em.allocPinned(S * em.wireBound(frames, R)));
  const offs = new Float64Array(S + 1);
  const C = eng.computeColumnsPacked(pcm, S, L, fftSize, hop, true, wire, offs);
  if (C !== frames || offs[0] !== 0 || !(offs[S] < idx.length)) throw new Error('computeColumnsPacked: columns / offsets');
  const out = new Uint8Array(frames * R);
  for (let s = 0; s < S; s++) {
    em.unpackWire(wire.subarray(offs[s], offs[s + 1]), frames, R, out);
    for (let i = 0; i < out.length; i++) if (out[i] !== idx[s * frames * R + i]) throw new Error('packed stream ' + s + ' differs at ' + i);
  }
  let threw = false;
  try { em.unpackWire(wire.subarray(0, 20), frames, R, out); } catch (e) { threw = e.code === 'EMSPEC_ERR_INVALID_ARG'; }
  if (!threw) throw new Error('a truncated image must throw EMSPEC_ERR_INVALID_ARG');
  threw = false;
  try { eng.computeColumnsPacked(pcm, S, L, fftSize, hop, true, wire.subarray(0, 4096), offs); } catch (e) { threw = e.code === 'EMSPEC_ERR_INVALID_ARG'; }
  if (!threw) throw new Error('a wire buffer that is too small must throw EMSPEC_ERR_INVALID_ARG');
  eng.destroy();
}

function checkBuildInfo() {
  const info = em.buildInfo();
  if (!/^emspec abi=2 sources=[0-9a-f]{16} arch=gfx950$/.test(info)) throw new Error('unexpected build info: ' + info);
}

/* multi-GPU entry points on one rank: communicator creation, batch + RCCL gather (world 1: the root's own columns) */
function checkGatherOneRank() {
  const eng = em.createEngine({});
  const fftSize = 4096, hop = 256, frames = 40, S = 2, L = fftSize + hop * (frames - 1);
  const pcm = new Float32Array(S * L);
  pcm.set(synth(L), 0); pcm.set(synth(L).map((v, i) => v * (i % 7 ? 1 : 0.5)), L);
  const R = eng.rows;
  const idx = new Uint8Array(S * frames * R), all = new Uint8Array(S * frames * R), db = new Float32Array(S * frames * R);
  eng.computeColumns(pcm, S, L, fftSize, hop, true, { index: idx });
  let threw = false;
  try { eng.computeColumnsGather(pcm, S, L, fftSize, hop, true, 0, { allIndex: all }); } catch (e) { threw = e.code === 'EMSPEC_ERR_STATE'; }
  if (!threw) throw new Error('gather without a communicator must throw EMSPEC_ERR_STATE');
  const id = em.commUniqueId();
  if (!(id instanceof Uint8Array) || id.length !== 128) throw new Error('commUniqueId');
  eng.commInit(id, 0, 1);
  eng.computeColumnsGather(pcm, S, L, fftSize, hop, true, 0, { allIndex: all, db });
  for (let i = 0; i < idx.length; i++) if (idx[i] !== all[i]) throw new Error('gathered index differs from the batch index at ' + i);
  threw = false;
  try { eng.computeColumnsGather(pcm, S, L, fftSize, hop, true, 0, { allIndex: new Uint8Array(10) }); } catch (e) { threw = e.code === 'EMSPEC_ERR_INVALID_ARG'; }
  if (!threw) throw new Error('short gather buffer must throw EMSPEC_ERR_INVALID_ARG');
  threw = false;   // ADVICE r01: an output sized for fewer rows than the engine's must be refused, not overrun
  try { eng.computeColumns(pcm, S, L, fftSize, hop, true, { db: new Float32Array(S * frames * 512) }); } catch (e) { threw = e.code === 'EMSPEC_ERR_INVALID_ARG'; }
  if (!threw) throw new Error('batch output with the wrong row count must throw EMSPEC_ERR_INVALID_ARG');
  eng.destroy();
}

function synth(L) {
  const x = new Float32Array(L);
  let s = 12345;
  for (let i = 0; i < L; i++) {
    s = (Math.imul(s, 1103515245) + 12345) | 0;
    const noise = ((s >>> 8) / 16777216 - 0.5) * 2e-3;
    x[i] = 0.5 * Math.sin(2 * Math.PI * 440 * i / 48000) + 0.25 * Math.sin(2 * Math.PI * (1000 + 2000 * i / 48000) * i / 48000) + noise;
  }
  x[3000] += 0.9;
  return x;
}

function check(fftSize, hop, reassign, frames) {
  const eng = em.createEngine({});
  const L = fftSize + hop * (frames - 1);
  const pcm = synth(L);
  const R = eng.rows;
  const ref = new Float32Array(frames * R);
  const C = eng.computeColumns(pcm, 1, L, fftSize, hop, reassign, { db: ref });
  if (C !== frames) throw new Error('column count ' + C);
  const D = em.latencyColumns(fftSize, hop, reassign);
  const got = new Float32Array(frames * R);
  let emitted = 0;
  for (let j = 0; j < frames; j++) {
    const col = eng.computeSpectrogramColumn(pcm.subarray(j * hop, j * hop + fftSize), fftSize, hop, reassign);
    if (col.length !== R) throw new Error('column length');
    if (j < D) { if (eng.lastColumn !== -1) throw new Error('priming column index'); continue; }
    if (eng.lastColumn !== j - D) throw new Error('column index ' + eng.lastColumn);
    got.set(col, (j - D) * R); emitted++;
  }
  for (let k = 0; k < D; k++) { const col = eng.flush(); got.set(col, eng.lastColumn * R); emitted++; }
  if (emitted !== frames) throw new Error('emitted ' + emitted);
  let worst = 0;
  for (let i = 0; i < got.length; i++) worst = Math.max(worst, Math.abs(got[i] - ref[i]));
  if (!(worst < 8.7e-4)) throw new Error('streaming vs batch dB mismatch ' + worst);
  let threw = false;
  try { eng.flush(); } catch (e) { threw = e.code === 'EMSPEC_ERR_STATE'; }
  if (!threw) throw new Error('flush past the end must throw EMSPEC_ERR_STATE');
  eng.destroy();
  return worst;
}

/* sample-block streaming: uneven blocks (an audio callback's sizes) must give the batch columns */
function checkPush(fftSize, hop, reassign, frames) {
  const eng = em.createEngine({});
  const L = fftSize + hop * (frames - 1) + 77;     // a ragged tail that completes no frame
  const pcm = synth(L);
  const R = eng.rows;
  const ref = new Float32Array(frames * R);
  eng.computeColumns(pcm, 1, L, fftSize, hop, reassign, { db: ref });
  const got = new Float32Array(frames * R);
  const sizes = [1, 300, 4096, 17, 2048, 20000, 5];
  let pos = 0, next = 0, i = 0;
  while (pos < L) {
    const len = Math.min(sizes[i++ % sizes.length], L - pos);
    const r = eng.pushSamples(pcm.subarray(pos, pos + len), fftSize, hop, reassign);
    pos += len;
    if (r.count > 0) {
      if (r.first !== next) throw new Error('push: first column ' + r.first + ' expected ' + next);
      got.set(r.db, next * R); next += r.count;
    } else if (r.first !== -1) throw new Error('push: empty block must report -1');
  }
  const D = em.latencyColumns(fftSize, hop, reassign);
  if (next !== frames - D) throw new Error('push: columns before flush ' + next);
  for (let k = 0; k < D; k++) { const col = eng.flush(); got.set(col, eng.lastColumn * R); }
  let worst = 0;
  for (let k = 0; k < got.length; k++) worst = Math.max(worst, Math.abs(got[k] - ref[k]));
  if (!(worst < 8.7e-4)) throw new Error('pushSamples vs batch dB mismatch ' + worst);
  let threw = false;
  try { eng.computeSpectrogramColumn(pcm.subarray(0, fftSize), fftSize, hop, reassign); } catch (e) { threw = e.code === 'EMSPEC_ERR_STATE'; }
  if (!threw) throw new Error('mixing frame and sample feeding must throw EMSPEC_ERR_STATE');
  eng.destroy();
  return worst;
}

async function checkAsync() {
  const eng = em.createEngine({});
  const fftSize = 4096, hop = 256, frames = 24, L = fftSize + hop * (frames - 1);
  const pcm = synth(L);
  const a = new Float32Array(frames * eng.rows), b = new Float32Array(frames * eng.rows);
  eng.computeColumns(pcm, 1, L, fftSize, hop, true, { db: a });
  const C = await eng.computeColumnsAsync(pcm, 1, L, fftSize, hop, true, { db: b });
  if (C !== frames) throw new Error('async column count');
  let worst = 0;
  for (let i = 0; i < a.length; i++) worst = Math.max(worst, Math.abs(a[i] - b[i]));
  if (!(worst < 2e-4)) throw new Error('async vs sync mismatch ' + worst);
  let rejected = false;
  try { await eng.computeColumnsAsync(pcm, 1, L, 3000, hop, true, { db: b }); } catch (e) { rejected = e.code === 'EMSPEC_ERR_INVALID_ARG'; }
  if (!rejected) throw new Error('async error path');
  // the packed entry off the JS thread: the image expands to the index columns of the synchronous call (float32 mode: +-1 cells)
  const idx = new Uint8Array(frames * eng.rows), wire = new Uint8Array(em.wireBound(frames, eng.rows)), offs = new Float64Array(2);
  eng.computeColumns(pcm, 1, L, fftSize, hop, true, { index: idx });
  const Cp = await eng.computeColumnsPackedAsync(pcm, 1, L, fftSize, hop, true, wire, offs);
  const back = new Uint8Array(frames * eng.rows);
  em.unpackWire(wire.subarray(0, offs[1]), frames, eng.rows, back);
  let off1 = 0;
  for (let i = 0; i < idx.length; i++) { const d = Math.abs(idx[i] - back[i]); if (d > 1) throw new Error('packed async differs at ' + i); off1 += d; }
  if (Cp !== frames || !(offs[1] > 32) || off1 > 1e-3 * idx.length) throw new Error('packed async: columns / offsets / +-1 share');
  rejected = false;
  try { await eng.computeColumnsPackedAsync(pcm, 1, L, fftSize, hop, true, wire.subarray(0, 64), offs); } catch (e) { rejected = e.code === 'EMSPEC_ERR_INVALID_ARG'; }
  if (!rejected) throw new Error('packed async error path');
  const edges = em.warpedEdges(eng.rows, 30, 20000, 2.0, 1.5);
  eng.setRowEdges(edges);
  const got = eng.getRowEdges();
  for (let i = 0; i < edges.length; i++) if (got[i] !== edges[i]) throw new Error('row edges round trip');
  const hz = eng.rowToHz(0.5);
  if (!(hz > edges[0] && hz < edges[1])) throw new Error('rowToHz');
  eng.setRowEdges(null);
  eng.setColormap(em.makeColormap(0.44));
  // pinned buffers: same results, faster copies
  const pin = new Float32Array(em.allocPinned(pcm.length * 4));
  pin.set(pcm);
  const pout = new Float32Array(em.allocPinned(a.length * 4));
  eng.setColormap(em.makeColormap(0.5));
  eng.computeColumns(pin, 1, L, fftSize, hop, true, { db: pout });
  let w2 = 0;
  for (let i = 0; i < a.length; i++) w2 = Math.max(w2, Math.abs(a[i] - pout[i]));
  if (!(w2 < 2e-4)) throw new Error('pinned-buffer batch mismatch ' + w2);
  eng.destroy();
}

/* Live multi-stream engine ({streams: S}): computeSpectrogramColumns (one frame of every stream per call, one launch),
 * flushColumns, pushSamplesMulti, resetStream and the async-work form.  EXACT mode: every stream's columns must equal the
 * batched call's float32 bits; stream 0 is also held against the plain-JS float64 method (8.7e-4 dB on strong cells). */
async function checkLive() {
  const S = 6, fftSize = 4096, hop = 256, frames = 64, L = fftSize + hop * (frames - 1);
  const eng = em.createEngine({ exact: true, streams: S });
  const R = eng.rows, D = em.latencyColumns(fftSize, hop, true);
  const pcm = new Float32Array(S * L);
  const base = synth(L);
  for (let s = 0; s < S; s++) pcm.set(base.map((v, i) => v * (1 - 0.12 * s) * ((i + 3 * s) % 7 ? 1 : 0.6)), s * L);
  const batch = new Float32Array(S * frames * R);
  const one = em.createEngine({ exact: true });
  one.computeColumns(pcm, S, L, fftSize, hop, true, { db: batch });
  one.destroy();
  const batchShort = new Float32Array(10 * R);   // stream 0 cut after ten frames (its last D columns differ from the long run's)
  {
    const o2 = em.createEngine({ exact: true });
    o2.computeColumns(pcm.subarray(0, fftSize + 9 * hop), 1, fftSize + 9 * hop, fftSize, hop, true, { db: batchShort });
    o2.destroy();
  }
  const got = new Float32Array(S * frames * R);
  const take = (db) => {
    for (let s = 0; s < S; s++) if (eng.columnIndex[s] >= 0) got.set(db.subarray(s * R, (s + 1) * R), (s * frames + eng.columnIndex[s]) * R);
  };
  eng._liveBlocks(fftSize, false);
  for (let j = 0; j < frames; j++) {
    for (let s = 0; s < S; s++) eng.frames.set(pcm.subarray(s * L + j * hop, s * L + j * hop + fftSize), s * fftSize);
    // alternate the synchronous and the async-work form
    const db = (j & 1) ? await eng.computeSpectrogramColumnsAsync(eng.frames, fftSize, hop, true) : eng.computeSpectrogramColumns(eng.frames, fftSize, hop, true);
    for (let s = 0; s < S; s++) if (eng.columnIndex[s] !== (j >= D ? j - D : -1)) throw new Error('live: column index of stream ' + s + ' at call ' + j);
    take(db);
  }
  for (let k = 0; k < D; k++) take(eng.flushColumns());
  let threw = false;
  try { eng.flushColumns(); } catch (e) { threw = e.code === 'EMSPEC_ERR_STATE'; }
  if (!threw) throw new Error('live: flush past the end must throw EMSPEC_ERR_STATE');
  for (let i = 0; i < got.length; i++) if (got[i] !== batch[i]) throw new Error('live exact columns differ from the batch at cell ' + i);
  const want = ref.columnsDb(pcm.subarray(0, L), fftSize, hop, true, frames);
  let strong = 0, worst = 0;
  for (let i = 0; i < want.length; i++) if (want[i] > -60) { strong++; worst = Math.max(worst, Math.abs(got[i] - want[i])); }
  if (!(strong > 100 && worst < 8.7e-4)) throw new Error('live vs plain-JS float64 oracle: ' + strong + ' strong cells, worst ' + worst + ' dB');
  // sample blocks: a hop per call for every stream, stream 2 restarted half way on stream 0's audio
  eng.reset();
  const got2 = new Float32Array(S * frames * R);
  const cut = fftSize + hop * 19;                      // samples fed when the restart happens (frame 19 complete)
  let fed = 0;
  const blk = new Float32Array(S * hop);
  const newCols = [];
  while (fed < L) {
    const cnt = Math.min(hop, L - fed);
    if (fed === cut) eng.resetStream(2);
    const b = cnt === hop ? blk : new Float32Array(S * cnt);
    for (let s = 0; s < S; s++) {
      const src = (s === 2 && fed >= cut) ? pcm.subarray(fed - cut, fed - cut + cnt) : pcm.subarray(s * L + fed, s * L + fed + cnt);
      b.set(src, s * cnt);
    }
    const r = eng.pushSamplesMulti(b, fftSize, hop, true);
    for (let s = 0; s < S; s++) for (let i = 0; i < r.counts[s]; i++) {
      const colv = r.db.subarray((s * r.maxColumns + i) * R, (s * r.maxColumns + i + 1) * R);
      if (s === 2 && fed >= cut) newCols.push(Float32Array.from(colv));
      else got2.set(colv, (s * frames + r.first[s] + i) * R);
    }
    fed += cnt;
  }
  for (let s = 0; s < S; s++) {
    const upto = (s === 2 ? 20 : frames) - D;          // complete columns before the restart / before the flush
    for (let i = 0; i < upto * R; i++) if (got2[s * frames * R + i] !== batch[s * frames * R + i]) throw new Error('live push: stream ' + s + ' differs at ' + i);
  }
  if (newCols.length < 8) throw new Error('live push: restarted stream produced ' + newCols.length + ' columns');
  for (let c = 0; c < newCols.length; c++) for (let r = 0; r < R; r++)
    if (newCols[c][r] !== batch[c * R + r]) throw new Error('live push: restarted stream column ' + c + ' differs from stream 0');
  threw = false;
  try { eng.computeSpectrogramColumns(eng.frames, fftSize, hop, true); } catch (e) { threw = e.code === 'EMSPEC_ERR_STATE'; }
  if (!threw) throw new Error('live: mixing frame and sample feeding must throw EMSPEC_ERR_STATE');
  eng.destroy();
  // a session fed by sample blocks only: its pending columns drain through flushColumns (no frame block was ever made)
  const eng2 = em.createEngine({ exact: true, streams: 2 });
  const two = new Float32Array(2 * (fftSize + 9 * hop));
  two.set(pcm.subarray(0, fftSize + 9 * hop), 0); two.set(pcm.subarray(L, L + fftSize + 9 * hop), fftSize + 9 * hop);
  const r2 = eng2.pushSamplesMulti(two, fftSize, hop, true);
  if (r2.counts[0] !== 10 - D || r2.counts[1] !== 10 - D) throw new Error('live push: ' + r2.counts[0] + ' columns from ten frames');
  for (let k = 0; k < D; k++) {
    const db = eng2.flushColumns();
    const c = 10 - D + k;
    if (eng2.columnIndex[0] !== c || eng2.columnIndex[1] !== c) throw new Error('live flush after sample blocks: column index');
    for (let r = 0; r < R; r++) if (db[r] !== batchShort[c * R + r]) throw new Error('live flush after sample blocks: column ' + c + ' differs');
  }
  eng2.destroy();
  // module-level drop-in
  const cols = em.computeSpectrogramColumns(new Float32Array(3 * 1024), 1024, 256, false);
  if (cols.length !== 3 * 1024) throw new Error('module-level multi-stream call');
  return worst;
}

checkAgainstJsOracle(1024, 256, false, 40);
checkAgainstJsOracle(4096, 256, true, 48);
const wx = checkExactAgainstJsOracle(4096, 256, 48);
checkGatherOneRank();
checkPacked();
checkBuildInfo();
const w1 = check(1024, 256, false, 40);
const w2 = check(4096, 256, true, 40);
checkPush(4096, 256, true, 150);
checkPush(1024, 256, false, 90);
const col = em.computeSpectrogramColumn(new Float32Array(1024), 1024, 256, false);
if (col.length !== 1024) throw new Error('module-level call');
checkAsync().then(checkLive).then((wl) => {
  console.log('node addon ok: max |dB| diff streaming vs batch', w1.toExponential(2), w2.toExponential(2), '; exact mode vs plain-JS float64, worst strong cell', wx.toExponential(2), 'dB; live multi-stream (6 streams, exact) == batch bytes, vs float64', wl.toExponential(2), 'dB');
}).catch((e) => { console.error(e); process.exit(1); });
