#!/usr/bin/env python3
"""Peephole pass over the gfx950 assembly of te_msm.hip (csrc/Makefile runs it between `hipcc -S` and the assembler).

The field products keep their multiply-accumulate chains in source order with EMPTY inline-asm markers (fp.hpp, chain()).
LLVM's hazard recognizer treats any inline asm as a possible dst_sel writer (gfx940+ "DstSel forwarding" hazard, one wait
state) and puts an `s_nop 0` between a marker and the next instruction that reads the marker's register.  The marker is
empty, so there is nothing to wait for -- but the nops take issue slots: 72 per accumulated point in k_accumulate<9>
(4.5 % of the loop, profiles/r02_isa_hist_k_accumulate.txt).

Removed here: `s_nop 0` directly behind an empty ;;#ASMSTART / ;;#ASMEND pair, and only when the last real instruction
before the marker is a plain VALU/SALU integer instruction from the list below (no SDWA / DPP / op_sel forms, no
transcendental, no lane-access or memory instruction) -- so a wait state that a REAL hazard of that instruction needs is
never dropped.  Anything else is left as the compiler wrote it.

usage: peephole.py in.s out.s   (prints the number of nops removed per kernel family to stderr)
"""
import re
import sys

SAFE_PRODUCERS = {
    "v_mad_u64_u32", "v_lshrrev_b64", "v_lshlrev_b64", "v_lshl_add_u64", "v_add_u32", "v_sub_u32", "v_subrev_u32",
    "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_bitop3_b32", "v_mov_b32", "v_mov_b64", "v_alignbit_b32",
    "v_lshrrev_b32", "v_lshlrev_b32", "v_and_or_b32", "v_lshl_or_b32", "v_lshl_add_u32", "v_add3_u32", "v_bfe_u32",
    "v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32", "v_cndmask_b32", "v_mul_lo_u32", "v_mul_hi_u32",
    "v_accvgpr_write_b32", "v_accvgpr_read_b32",
    "s_mov_b32", "s_mov_b64", "s_add_u32", "s_addc_u32", "s_and_b32", "s_and_b64", "s_or_b64", "s_lshl_b32", "s_lshr_b32",
}
BAD_TEXT = ("sdwa", "dpp", "op_sel", "dst_sel", "row_", "quad_perm", "wave_")


def is_code(line):
    s = line.strip()
    return bool(s) and not s.startswith(";") and not s.startswith(".") and not s.endswith(":")


def main():
    src, dst = sys.argv[1], sys.argv[2]
    lines = open(src).read().split("\n")
    out = []
    removed = kept = 0
    last_real = None                     # last real instruction emitted (markers and comments do not count)
    i = 0
    while i < len(lines):
        l = lines[i]
        s = l.strip()
        if s.endswith(":") and not s.startswith(";"):
            last_real = None             # a label: control flow may arrive from elsewhere
        if (s == "s_nop 0" and len(out) >= 2 and out[-1].strip() == ";;#ASMEND" and out[-2].strip() == ";;#ASMSTART"):
            ok = False
            if last_real is not None:
                mnem = re.split(r"[\s]", last_real, 1)[0]
                mnem = re.sub(r"_e(32|64)$", "", mnem)
                ok = mnem in SAFE_PRODUCERS and not any(b in last_real for b in BAD_TEXT)
            if ok:
                removed += 1
                i += 1
                continue
            kept += 1
        if is_code(l) and not s.startswith(";;#"):
            last_real = s.split(";")[0].strip()
        out.append(l)
        i += 1
    open(dst, "w").write("\n".join(out))
    sys.stderr.write("peephole: %d marker nops removed, %d kept\n" % (removed, kept))


if __name__ == "__main__":
    main()
