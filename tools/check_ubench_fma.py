"""bigint check of the SAMPLE lines printed by tools/ubench_fma (both product forms): r * R == a * b (mod p)"""
import sys
P = 8444461749428370424248824938781546531375899335154063827935233455917409239041
ok = bad = 0
for line in sys.stdin:
    if not line.startswith("SAMPLE"):
        if line.startswith("RESULT"):
            print(line.rstrip())
        continue
    f = line.split()
    kind, bitsz, nl, rbits = (f[1], 52, 5, 260) if f[1] == "fma" else (f[1], 29, 9, 261)
    ia, ib, ir = f.index("a"), f.index("b"), f.index("r")
    val = lambda xs: sum(int(v) << (bitsz * i) for i, v in enumerate(xs))
    a, b, r = val(f[ia + 1:ib]), val(f[ib + 1:ir]), val(f[ir + 1:ir + 1 + nl])
    good = (r << rbits) % P == (a * b) % P and r < 4 * P
    ok += good; bad += not good
    if not good:
        print("MISMATCH", kind, a, b, r)
print("product samples checked against bigints: %d right, %d wrong" % (ok, bad))
sys.exit(1 if bad else 0)
