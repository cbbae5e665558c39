// node run_awaited.js <points.bin> <scalars.bin> [reps] [bases]  -- the reference harness's call pattern (ui/Benchmark.tsx:29-39: every call awaited):
// prints {"x","y","median_ms","min_ms","reps","stats"}.
'use strict';
const fs = require('fs');
const { compute_msm, setBases, getStats } = require('./compute_msm.js');
(async () => {
  const points = fs.readFileSync(process.argv[2]);
  const scalars = fs.readFileSync(process.argv[3]);
  const reps = parseInt(process.argv[4] || '30', 10);
  if (process.argv[5] === 'bases') setBases(points);
  const ms = () => Number(process.hrtime.bigint()) / 1e6;
  let r = null;
  for (let i = 0; i < 5; i++) r = await compute_msm(points, scalars, false);
  const ts = [];
  for (let i = 0; i < reps; i++) { const t0 = ms(); r = await compute_msm(points, scalars, false); ts.push(ms() - t0); }
  ts.sort((a, b) => a - b);
  console.log(JSON.stringify({ x: r.x.toString(), y: r.y.toString(), median_ms: ts[ts.length >> 1], min_ms: ts[0], reps, stats: getStats() }));
})().catch((e) => { console.log(JSON.stringify({ error: String(e && e.message ? e.message : e) })); process.exitCode = 3; });
