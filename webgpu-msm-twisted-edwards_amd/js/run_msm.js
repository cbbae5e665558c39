// CLI used by the tests: node run_msm.js <points.bin> <scalars.bin>  ->  prints {"x": "...", "y": "..."}
'use strict';
const fs = require('fs');
const { compute_msm } = require('./compute_msm.js');
(async () => {
  const points = fs.readFileSync(process.argv[2]);
  const scalars = fs.readFileSync(process.argv[3]);
  try {
    const r = await compute_msm(points, scalars, false);
    console.log(JSON.stringify({ x: r.x.toString(), y: r.y.toString() }));
  } catch (e) {
    console.log(JSON.stringify({ error: String(e && e.message ? e.message : e) }));
    process.exitCode = 3;
  }
})();
