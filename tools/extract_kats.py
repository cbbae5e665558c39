"""Extracts the known-answer DATA (decimal strings / integers only) that the reference's own unit tests
hold for this path into tests/golden/reference_kats.json.  BUILD CONTAINER ONLY (reads /root/reference).
Sources (relative to /root/reference/src):
  reference/utils/FieldMath.test.ts:5-95        scalar-mul KATs, x -> y decompression KATs
  reference/utils/wasmFunctions.test.ts:4-49    field add/double, group add, group scalar-mul KATs
  reference/webgpu/utils.test.ts:4-14           bigint <-> 8 x u32 (big-endian words) codec KATs
  submission/miscellaneous/tests/utils.test.ts:17-26,171-178   13-bit limbs of p, compute_misc_params values
  test-data/testCases.ts:11-32                  expected results of the official ZPrize vectors (inputs not in tree)
"""
import json
import os
import re

REF = os.environ.get("TE_REFERENCE_ROOT", "/root/reference") + "/src/"


def nums(text):
    return re.findall(r"'(\d+)(?:field|group|scalar)?'", text)


def main():
    out = {}
    t = open(REF + "reference/utils/FieldMath.test.ts").read()
    mul_part, dec_part = t.split("describe('getPointFromX'")
    v = nums(mul_part)
    out["scalar_mul"] = [{"x": v[i], "y": v[i + 1], "k": v[i + 2], "rx": v[i + 3], "ry": v[i + 4]} for i in range(0, len(v), 5)]
    v = nums(dec_part)
    out["point_from_x"] = [{"x": v[i], "y": v[i + 1]} for i in range(0, len(v), 2)]

    t = open(REF + "reference/utils/wasmFunctions.test.ts").read()
    sec = re.split(r"describe\('(addFields|doubleField|addGroups|groupScalarMul)'", t)
    secs = {sec[i]: sec[i + 1] for i in range(1, len(sec), 2)}
    v = nums(secs["addFields"]); out["add_fields"] = [v[i:i + 3] for i in range(0, len(v), 3)]
    v = nums(secs["doubleField"]); out["double_field"] = [v[i:i + 2] for i in range(0, len(v), 2)]
    v = nums(secs["addGroups"]); out["add_groups_x"] = [v[i:i + 3] for i in range(0, len(v), 3)]
    v = nums(secs["groupScalarMul"]); out["group_scalar_mul_x"] = [v[i:i + 3] for i in range(0, len(v), 3)]

    t = open(REF + "reference/webgpu/utils.test.ts").read()
    body = t.split("const testData")[1].split("];")[0]
    rows = re.findall(r"\[\s*(ALEO_FIELD_MODULUS|BigInt\('?(\d+)'?\)),\s*new Uint32Array\(\[([^\]]*)\]\)\]", body)
    out["u32_codec"] = [{"value": (r[1] if r[1] else "ALEO_FIELD_MODULUS"), "words_be": [int(x) for x in r[2].split(",")]} for r in rows]

    t = open(REF + "submission/miscellaneous/tests/utils.test.ts").read()
    m = re.search(r"\[\s*1,\s*0,\s*0,\s*768[^\]]*\]", t)
    out["p_limbs_13"] = [int(x) for x in re.findall(r"\d+", m.group(0))]
    m = re.search(r"max_terms:\s*(\d+),\s*k:\s*(\d+),\s*nsafe:\s*(\d+),\s*n0:\s*BigInt\((\d+)\)", t)
    out["misc_params_13"] = {"max_terms": int(m.group(1)), "k": int(m.group(2)), "nsafe": int(m.group(3)), "n0": int(m.group(4))}

    t = open(REF + "test-data/testCases.ts").read()
    cases = re.findall(r"case (\d+):\s*return \{ x: BigInt\('(\d+)'\), y: BigInt\('(\d+)'\)", t)
    out["zprize_expected"] = {c[0]: {"x": c[1], "y": c[2]} for c in cases}

    path = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "reference_kats.json")
    json.dump(out, open(path, "w"), indent=1)
    print({k: (len(v) if hasattr(v, "__len__") else v) for k, v in out.items()})


if __name__ == "__main__":
    main()
