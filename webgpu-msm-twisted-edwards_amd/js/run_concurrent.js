// CLI used by the tests:  node run_concurrent.js <points.bin> <scalars.bin> <k> [devices]
// One warm-up call, one timed call, then k compute_msm promises in flight at once (a warm-up burst, then the best of three) (the reference's harness awaits each call,
// ui/Benchmark.tsx:32; a prover need not).  Prints {"x","y","single_ms","concurrent_ms","k","all_equal","devices"}.
// devices: "0,0" etc. -> setDevices([...]) before the first call.
'use strict';
const fs = require('fs');
const { compute_msm, setDevices, getDevices } = require('./compute_msm.js');
(async () => {
  const points = fs.readFileSync(process.argv[2]);
  const scalars = fs.readFileSync(process.argv[3]);
  const k = parseInt(process.argv[4] || '4', 10);
  if (process.argv[5]) setDevices(process.argv[5].split(',').map((t) => parseInt(t, 10)));
  try {
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
    console.log(JSON.stringify({ x: r.x.toString(), y: r.y.toString(), single_ms: single, concurrent_ms: conc, k, all_equal: same,
                                 devices: getDevices() }));
  } catch (e) {
    console.log(JSON.stringify({ error: String(e && e.message ? e.message : e) }));
    process.exitCode = 3;
  }
})();
