'use strict';
/* Synthetic 48 kHz test audio, the same counter-based definition as em-spec_amd/emspec/synth.py
 * (SURVEY.md §8(d), [BUILD-DEFINED]): stream s uses seed 1000+s; splitmix64 -> uniforms; 8 sinusoids
 * (log-uniform 30 Hz..20 kHz, -40..0 dBFS) + one linear chirp + Gaussian noise at -60 dBFS + a unit
 * click every 24000 samples, scaled into [-1, 1].  Needs BigInt (node >= 10.4). */
const M64 = (1n << 64n) - 1n;

function splitmix64(seed, index) {            // draw number `index` (1-based) of the generator seeded with `seed`
  let z = (BigInt(seed) + BigInt(index) * 0x9E3779B97F4A7C15n) & M64;
  z = ((z ^ (z >> 30n)) * 0xBF58476D1CE4E5B9n) & M64;
  z = ((z ^ (z >> 27n)) * 0x94D049BB133111EBn) & M64;
  return z ^ (z >> 31n);
}
function uniform(seed, index) { return Number(splitmix64(seed, index) >> 11n) / 9007199254740992; }

function stream(s, L, fs = 48000) {
  const seed = 1000 + s;
  const u = []; for (let i = 0; i < 64; i++) u.push(uniform(seed, i + 1));
  const x = new Float64Array(L);
  for (let i = 0; i < 8; i++) {
    const f = 30 * Math.pow(20000 / 30, u[i]), a = Math.pow(10, -40 * u[8 + i] / 20), ph = 2 * Math.PI * u[16 + i];
    for (let n = 0; n < L; n++) x[n] += a * Math.sin(2 * Math.PI * f * (n / fs) + ph);
  }
  const f0 = 200 + 4000 * u[24], rate = 4e4 * (0.25 + 0.75 * u[25]);
  const period = Math.max(1e-3, Math.min(L / fs, (20000 - f0) / rate));
  for (let n = 0; n < L; n++) {
    const t = n / fs, tt = t - Math.floor(t / period) * period;
    x[n] += 0.25 * Math.sin(2 * Math.PI * (f0 * tt + 0.5 * rate * tt * tt));
  }
  const nseed = BigInt(seed) ^ 0x5EEDn;      // Box-Muller pairs on draws 65, 66, ...
  for (let p = 0; 2 * p < L; p++) {
    const u1 = Math.max(uniform(nseed, 64 + 2 * p + 1), 1e-300), u2 = uniform(nseed, 64 + 2 * p + 2);
    const r = Math.sqrt(-2 * Math.log(u1));
    x[2 * p] += 1e-3 * r * Math.cos(2 * Math.PI * u2);
    if (2 * p + 1 < L) x[2 * p + 1] += 1e-3 * r * Math.sin(2 * Math.PI * u2);
  }
  for (let n = 0; n < L; n += 24000) x[n] += 1.0;
  let peak = 1.0;
  for (let n = 0; n < L; n++) peak = Math.max(peak, Math.abs(x[n]));
  const out = new Float32Array(L);
  for (let n = 0; n < L; n++) out[n] = x[n] / peak;
  return out;
}

module.exports = { splitmix64, uniform, stream };

if (require.main === module) {               // node synth.js <stream> <L>  -> JSON array (tests/test_oracle.py)
  const s = parseInt(process.argv[2] || '0', 10), L = parseInt(process.argv[3] || '1024', 10);
  process.stdout.write(JSON.stringify(Array.from(stream(s, L))));
}
