#!/usr/bin/env python3
"""Instruction-class histogram of one kernel in the gfx950 ISA listing (`make -C csrc asm`).

usage: tools/isa_hist.py <listing.s> <kernel-substring> [--loop]
  --loop   restrict to the hottest loop: the first loop of the kernel that holds at least one field product
           (k_accumulate's per-point loop)
Prints the count per mnemonic and per class, with the issue cost measured by tools/ubench.hip
(profiles/r01_ubench_instruction_rates.txt, r02_ubench_instruction_rates.txt: cycles per wave instruction at 4 waves/SIMD).
"""
import re
import sys
from collections import Counter

# measured issue cost (cycles / wave instruction, 4 waves per SIMD); unknown VALU mnemonics default to 4.3 (VOP3)
COST = {"v_mad_u64_u32": 4.49, "v_add_u32": 2.62, "v_sub_u32": 2.62, "v_subrev_u32": 2.62, "v_and_b32": 2.62, "v_or_b32": 2.62, "v_xor_b32": 2.62,
        "v_mov_b32": 2.41, "v_lshrrev_b64": 4.17, "v_lshlrev_b64": 4.17, "v_lshl_add_u64": 4.17, "v_cndmask_b32": 4.29, "v_add3_u32": 4.29,
        "v_lshrrev_b32": 2.62, "v_lshlrev_b32": 2.62, "v_alignbit_b32": 4.28, "v_and_or_b32": 4.29, "v_lshl_or_b32": 4.29, "v_lshl_add_u32": 4.29,
        "v_bfe_u32": 4.29, "v_add_co_u32": 4.19, "v_addc_co_u32": 4.19, "v_sub_co_u32": 4.19, "v_subb_co_u32": 4.19, "v_mul_lo_u32": 4.30,
        "v_mul_hi_u32": 4.27,
        # profiles/r02_ubench_instruction_rates.txt
        "v_bitop3_b32": 2.71, "v_not_b32": 2.46, "v_sad_u32": 4.31}


def kernel_body(lines, name):
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^_Z\w*%s\w*:" % re.escape(name), l):
            start = i
            break
    if start is None:
        raise SystemExit("kernel not found: " + name)
    body = []
    for l in lines[start + 1:]:
        if l.startswith("\t.section") or l.startswith(".Lfunc_end") or l.strip().startswith("s_endpgm") and False:
            break
        body.append(l.rstrip("\n"))
    return body


def sources_sha():
    """the rule of bench.py kernel_sources_sha()"""
    import hashlib
    import os
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "webgpu-msm-twisted-edwards_amd", "csrc")
    h = hashlib.sha256()
    for f in ("kernels.hip.hpp", "curve.hpp", "fp.hpp"):
        text = open(os.path.join(csrc, f), "r", errors="replace").read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"//[^\n]*", "", text)
        h.update(re.sub(r"\s+", "", text).encode())
    return h.hexdigest()[:16]


def analyse(path, name, loop_only, quiet=False):
    lines = open(path).read().split("\n")
    body = kernel_body(lines, name)
    # instructions with their label context
    labels, instrs = {}, []
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = len(instrs)
            continue
        s = l.strip()
        if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
            continue
        instrs.append(s.split(";")[0].strip())
    lo, hi = 0, len(instrs)
    if loop_only:
        # the per-point loop: the FIRST loop in program order that holds a field product's worth of multiply-accumulates (the
        # loop that sums the parts of a split bucket at the end of k_accumulate is larger -- nine products -- and comes later)
        best = None
        for i, ins in enumerate(instrs):
            m = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", ins)
            if m and m.group(1) in labels and labels[m.group(1)] <= i:
                start = labels[m.group(1)]
                mads = sum(1 for x in instrs[start:i + 1] if x.startswith("v_mad_u64_u32"))
                if mads >= 150 and (best is None or start < best[0]):
                    best = (start, i + 1, i - start)
        if best is None:
            raise SystemExit("no backward branch found")
        lo, hi = best[0], best[1]
    ops = Counter(ins.split()[0] for ins in instrs[lo:hi])
    ops = Counter({re.sub(r"_e(32|64)$|_dpp$|_sdwa$", "", k): 0 for k in ops}) + Counter()
    for ins in instrs[lo:hi]:
        ops[re.sub(r"_e(32|64)$|_dpp$|_sdwa$", "", ins.split()[0])] += 1
    total = sum(ops.values())
    valu = {k: v for k, v in ops.items() if k.startswith("v_")}
    cyc = sum(COST.get(k, 4.3) * v for k, v in valu.items())
    if quiet:
        return {"valu_issue_cycles": round(cyc, 1), "valu_instructions": sum(valu.values()), "mads": ops.get("v_mad_u64_u32", 0),
                "instructions": total, "s_nop": ops.get("s_nop", 0),
                "note": "%d v_mad_u64_u32 (%.2f clk each) + %d other VALU instructions per 64 accumulated points (tools/isa_hist.py on the built listing)"
                        % (ops.get("v_mad_u64_u32", 0), COST["v_mad_u64_u32"], sum(valu.values()) - ops.get("v_mad_u64_u32", 0))}
    print("kernel %s: %s, %d instructions (%d VALU, %d SALU, %d memory, %d other)" % (
        name, "hottest loop" if loop_only else "whole kernel", total, sum(valu.values()),
        sum(v for k, v in ops.items() if k.startswith("s_") and not k.startswith(("s_waitcnt", "s_nop", "s_cbranch", "s_branch"))),
        sum(v for k, v in ops.items() if k.startswith(("global_", "flat_", "buffer_", "ds_", "scratch_"))),
        sum(v for k, v in ops.items() if k.startswith(("s_waitcnt", "s_nop", "s_cbranch", "s_branch")))))
    print("estimated VALU issue: %.0f cycles per wave pass (mads %.0f)" % (cyc, COST["v_mad_u64_u32"] * ops.get("v_mad_u64_u32", 0)))
    for k, v in sorted(ops.items(), key=lambda kv: -kv[1]):
        print("  %-24s %6d  %5.1f %%%s" % (k, v, 100.0 * v / total, ("   ~%.2f cyc" % COST[k]) if k in COST else ""))


def main():
    if "--json" in sys.argv:
        # usage: isa_hist.py <listing.s> --json out.json : the per-point loop of both accumulation kernels, with the sources' hash
        import json
        path, outp = sys.argv[1], sys.argv[sys.argv.index("--json") + 1]
        j = {"listing": path, "kernel_sources_sha": sources_sha(),
             "k_accumulate<9>": analyse(path, "k_accumulateILi9ELi0", True, quiet=True),
             "k_accumulate<14>": analyse(path, "k_accumulateILi14ELi0", True, quiet=True),
             # round 6: BLS12-377 over BOUND bases -- affine records, 7 products per gathered point (k_accumulate<14, 1>)
             "k_accumulate<14,affine>": analyse(path, "k_accumulateILi14ELi1", True, quiet=True)}
        json.dump(j, open(outp, "w"), indent=1)
        print(json.dumps(j, indent=1))
        return
    analyse(sys.argv[1], sys.argv[2], "--loop" in sys.argv)


if __name__ == "__main__":
    main()
