"""examples/msm_example.c: the C-ABI from plain C (no Python, no torch in the process; the library's own RUNPATH finds the ROCm
runtime).  Without a GPU the program must fail loudly at te_msm_init (no CPU fallback); on the MI355X box it runs te_msm_run, tickets
in flight on one and on four "devices" and the scalar-range error, and checks that every path returned the same 64 bytes."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "webgpu-msm-twisted-edwards_amd")


def _build(tmp_path):
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    exe = str(tmp_path / "msm_example")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "msm_example.c"),
                           "-L", PKG, "-ltemsm", "-Wl,-rpath," + PKG, "-o", exe])
    return exe


def test_c_example_builds_and_fails_loudly_without_a_gpu(pkg, tmp_path):
    import torch
    exe = _build(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: see the gpu-marked test")
    r = subprocess.run([exe, "10"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode == 1 and b"no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_c_example_on_the_gpu(pkg, ora, tmp_path):
    exe = _build(tmp_path)
    for args in (["16"], ["17", "0,0,0,0"], ["12", "0,0"]):
        r = subprocess.run([exe] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        out = r.stdout.decode()
        assert r.returncode == 0, (args, out, r.stderr.decode()[-800:])
        assert "all results equal: yes" in out and "code -3" in out
        lg = int(args[0])
        pts, sc = pkg.synth_inputs(0x5EED0000 + lg, 1 << lg)
        want = ora.msm(pts, sc, threads=8)
        assert "x = 0x%064x" % int.from_bytes(want[:32], "little") in out and "y = 0x%064x" % int.from_bytes(want[32:], "little") in out
