// CLI used by the tests:  node run_concurrent.js <points.bin> <scalars.bin> <k> [devices|-] [bases]
// One warm-up call, one timed call, then k compute_msm promises in flight at once (a warm-up burst, then the best of three) (the reference's harness awaits each call,
// ui/Benchmark.tsx:32; a prover need not).  Prints {"x","y","single_ms","concurrent_ms","k","all_equal","devices"}.
// devices: "0,0" etc. -> setDevices([...]) before the first call ("-": leave the default).  A fifth argument "bases": setBases(points) first --
// the calls over that Buffer move the scalars only; one more call with a COPY of the points takes the ordinary path and must agree.
// "stats": how the promises were mapped onto the engine (getStats: maxInFlight = tickets in flight at a submit, boundJobs, ...).
'use strict';
const fs = require('fs');
const { compute_msm, setDevices, getDevices, setBases, getStats } = require('./compute_msm.js');
(async () => {
  const points = fs.readFileSync(process.argv[2]);
  const scalars = fs.readFileSync(process.argv[3]);
  const k = parseInt(process.argv[4] || '4', 10);
  if (process.argv[5] && process.argv[5] !== '-') setDevices(process.argv[5].split(',').map((t) => parseInt(t, 10)));
  const bases = process.argv[6] === 'bases';
  try {
    if (bases) setBases(points);
    const ms = () => Number(process.hrtime.bigint()) / 1e6;
    await compute_msm(points, scalars, false);                       // context, buffers, code upload
    let t0 = ms();
    const r = await compute_msm(points, scalars, false);
    const single = ms() - t0;
    await Promise.all(Array.from({ length: k }, () => compute_msm(points, scalars, false)));      // the work sets' first use
    // best of three bursts: the first concurrent copies of a process can carry pauses of the runtime (DESIGN.md section 6)
    let conc = Infinity, same = true;
    for (let rep = 0; rep < 3; rep++) {
      t0 = ms();
      const rs = await Promise.all(Array.from({ length: k }, () => compute_msm(points, scalars, false)));
      conc = Math.min(conc, ms() - t0);
      same = same && rs.every((q) => q.x === r.x && q.y === r.y);
    }
    if (bases) {
      const other = await compute_msm(Buffer.from(points), scalars, false);        // equal bytes, another Buffer: the ordinary path
      same = same && other.x === r.x && other.y === r.y;
      setBases(null);
      const after = await compute_msm(points, scalars, false);                      // unbound again
      same = same && after.x === r.x && after.y === r.y;
    }
    console.log(JSON.stringify({ x: r.x.toString(), y: r.y.toString(), single_ms: single, concurrent_ms: conc, k, all_equal: same,
                                 devices: getDevices(), stats: getStats() }));
  } catch (e) {
    console.log(JSON.stringify({ error: String(e && e.message ? e.message : e) }));
    process.exitCode = 3;
  }
})();
