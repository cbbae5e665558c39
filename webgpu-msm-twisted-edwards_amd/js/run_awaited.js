// node run_awaited.js <points.bin> <scalars.bin> [reps] [bases]  -- the reference harness's call pattern (ui/Benchmark.tsx:29-39: every call awaited):
// prints {"x","y","median_ms","min_ms","reps","first_calls_sync_ms","max_sync_ms","stats"}.
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
  const firstSync = [];
  for (let i = 0; i < 5; i++) { const t0 = ms(); const p = compute_msm(points, scalars, false); firstSync.push(Math.round((ms() - t0) * 1000) / 1000); r = await p; }
  const ts = [];
  // sync_ms: how long the JavaScript thread stays inside the call before it has its promise (the event loop is blocked for that long)
  let syncMax = 0;
  for (let i = 0; i < reps; i++) {
    const t0 = ms(); const p = compute_msm(points, scalars, false); const s = ms() - t0;
    r = await p; ts.push(ms() - t0);
    syncMax = Math.max(syncMax, s);
  }
  ts.sort((a, b) => a - b);
  console.log(JSON.stringify({ x: r.x.toString(), y: r.y.toString(), median_ms: ts[ts.length >> 1], min_ms: ts[0], reps, first_calls_sync_ms: firstSync,
                               max_sync_ms: syncMax, stats: getStats() }));
})().catch((e) => { console.log(JSON.stringify({ error: String(e && e.message ? e.message : e) })); process.exitCode = 3; });
