'use strict';
/*
 * example_live.js — what a renderer that draws many streams does with the live multi-stream engine (INTEGRATION.md §2):
 * S streams, one hop of new samples each per audio callback, ONE call, and the finished RGBA columns of every stream come
 * back in a page-locked block.  Here the "audio callback" is a loop over synthetic audio and the "canvas" is a PPM file of
 * stream 0's scrolling image.   node example_live.js [streams] [seconds] [out.ppm]      (needs a gfx950 GPU)
 */
const fs = require('fs');
const em = require('./index.js');

const S = +process.argv[2] || 8, seconds = +process.argv[3] || 2, out = process.argv[4] || '/tmp/emspec_live.ppm';
const fs48 = 48000, fftSize = 4096, hop = 256;
const engine = em.createEngine({ streams: S, gain: 3.5, dbRange: 58, gateDb: -65 });   // the sliders of the reference's settings panel
engine.setColormap(em.makeColormap(0.44));
const R = engine.rows, hops = Math.floor(seconds * fs48 / hop);
const block = engine.sampleBlock(hop);                    // page-locked Float32Array(S * hop): the kernel reads it in place
const image = Buffer.alloc(3 * R * hops);                 // stream 0: hops columns x R rows, RGB
let phase = new Float64Array(S), drawn = 0, t0 = process.hrtime.bigint(), inCall = 0n;
for (let j = 0; j < hops; j++) {
  for (let s = 0; s < S; s++)                             // "audio callback": a gliding tone per stream + a click every half second
    for (let i = 0; i < hop; i++) {
      const t = (j * hop + i) / fs48, f = 200 * (s + 1) * (1 + 2 * t / seconds);
      phase[s] += 2 * Math.PI * f / fs48;
      block[s * hop + i] = 0.4 * Math.sin(phase[s]) + ((j * hop + i) % 24000 === 0 ? 0.8 : 0);
    }
  const c0 = process.hrtime.bigint();
  const r = engine.pushSamplesMulti(block, fftSize, hop, true, true);   // one launch for all S streams
  inCall += process.hrtime.bigint() - c0;
  for (let i = 0; i < r.counts[0]; i++, drawn++) {        // stream 0's finished columns -> the image (row 0 at the bottom)
    const col = r.rgba.subarray(4 * i * R, 4 * (i + 1) * R);            // stream 0's block starts at 0: (0 * maxColumns + i)
    for (let row = 0; row < R; row++) image.set(col.subarray(4 * row, 4 * row + 3), 3 * ((R - 1 - row) * hops + r.first[0] + i));
  }
}
for (;;) {                                                // the last D columns of every stream
  try { engine.flushColumns(true); } catch (e) { if (e.code === 'EMSPEC_ERR_STATE') break; throw e; }
  const c = engine.columnIndex[0];
  if (c >= 0) { for (let row = 0; row < R; row++) image.set(engine.columnsRgba.subarray(4 * row, 4 * row + 3), 3 * ((R - 1 - row) * hops + c)); drawn++; }
}
const wall = Number(process.hrtime.bigint() - t0) / 1e9;
fs.writeFileSync(out, Buffer.concat([Buffer.from(`P6\n${hops} ${R}\n255\n`), image]));
console.log(`${S} streams x ${hops} hops: ${drawn} columns of stream 0 drawn to ${out}; ${(Number(inCall) / 1e3 / hops).toFixed(1)} us per call in the engine ` +
  `(${(100 * Number(inCall) / 1e9 / (hops * hop / fs48)).toFixed(2)} % of real time), ${wall.toFixed(2)} s wall with the synthetic audio`);
engine.destroy();
