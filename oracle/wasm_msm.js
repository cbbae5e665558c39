// Runs the reference's own CPU MSM -- Address.msm of the prebuilt Aleo WASM, the body of
// wasm_compute_msm (reference/reference.ts:29-39) -- under node, in the BUILD CONTAINER ONLY.
// Nothing of the reference is copied: the glue and the .wasm are read from /root/reference at run time.
// usage: node wasm_msm.js <cases.json> <out.json>
//   cases.json: [{name, xs:[decimal x...], ks:[decimal k...]}, ...]  -> out.json: [{name, x}]
const fs = require("fs");
const dir = (process.env.TE_REFERENCE_ROOT || "/root/reference") + "/src/reference/wasm-loader/";
let src = fs.readFileSync(dir + "aleo_wasm_bg.js", "utf8");
const names = [];
src = src.replace(/^export function (\w+)/gm, (m, n) => (names.push(n), "function " + n))
         .replace(/^export class (\w+)/gm, (m, n) => (names.push(n), "class " + n));
src += "\nmodule.exports={" + names.join(",") + "};";
const m = { exports: {}, require };
new Function("module", "exports", "require", src)(m, m.exports, require);
const glue = m.exports;
const inst = new WebAssembly.Instance(new WebAssembly.Module(fs.readFileSync(dir + "aleo_wasm_bg.wasm")),
                                      { "./aleo_wasm_bg.js": glue });
glue.__wbg_set_wasm(inst.exports);

const cases = JSON.parse(fs.readFileSync(process.argv[2], "utf8"));
const out = [];
for (const c of cases) {
  const t0 = Date.now();
  let x;
  if (c.op === "add") x = glue.Address.add_points(c.xs[0] + "group", c.xs[1] + "group");
  else if (c.op === "mul") x = glue.Address.group_scalar_mul(c.xs[0] + "group", c.ks[0] + "scalar");
  else x = glue.Address.msm(c.xs.map((v) => v + "group"), c.ks.map((v) => v + "scalar"));
  out.push({ name: c.name, x: x.replace("group", ""), ms: Date.now() - t0 });
  console.error(c.name, "n=" + c.xs.length, (Date.now() - t0) + " ms");
}
fs.writeFileSync(process.argv[3], JSON.stringify(out));
