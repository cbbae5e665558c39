"""The stream-sharing recipe of include/te_msm.h (te_msm_workset_stream + torch.cuda.ExternalStream) and its lifetime rule:
a handle handed out stays a valid hipStream_t until the process exits -- te_msm_destroy parks exported streams.

Round 5's tools/exp_batch_small.py was the only program in the tree that ordered torch work on the engine's own streams; it
printed its results and then "dumped core": its pinned host tensors had been the target of non-blocking copies on
ExternalStream(workset_stream) and outlived the context.  PyTorch's pinned-memory allocator records an event on every stream a
pinned block was used on WHEN THE BLOCK IS RELEASED (here: at interpreter exit); te_msm_destroy had destroyed the stream;
hipEventRecord failed inside a deleter and the process aborted (profiles/r06_batch_small_abort.txt).

The recipe runs in a CHILD process (the failure mode is an abort of the interpreter, at its exit) and the test asserts exit
code 0.  Reference side: none -- the reference has one device and one queue (implementation/cuzk/gpu.ts:14-25).
Nothing here reads /root/reference."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# what the child runs: the recipe, a context close with a pinned tensor still alive, a second context that takes the parked
# streams over, and pinned tensors that are only released at interpreter exit
CHILD = r'''
import importlib, os, sys
sys.path.insert(0, %(root)r)
pkg = importlib.import_module("webgpu-msm-twisted-edwards_amd")
import torch
n = 20000
pts, sc = pkg.synth_inputs(0xABCD, n, fixed_point="chain")
dp = torch.frombuffer(bytearray(pts), dtype=torch.uint8).cuda()
ds = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
torch.cuda.synchronize()
keep = []
handles = []
all8 = []
for round_ in range(2):
    with pkg.MsmContext((0,)) as c:
        cb, W = c.plan(n)
        all8.append({c.workset_stream(k)[0] for k in range(pkg.WORKSETS)})
        want = c.run_device(dp.data_ptr(), ds.data_ptr(), n)
        rows = torch.zeros(W * c.row_bytes, dtype=torch.uint8, device="cuda")
        host = torch.zeros(W * c.row_bytes, dtype=torch.uint8).pin_memory()
        for k in (0, 1):
            c.set_option("workset", k)
            st, _ = c.workset_stream(k)
            handles.append(st)
            c.partial_device(dp.data_ptr(), ds.data_ptr(), n, rows.data_ptr())
            with torch.cuda.stream(torch.cuda.ExternalStream(st)):
                host.copy_(rows, non_blocking=True)              # the pinned block now remembers the engine's stream
            c.partial_wait(k)
            torch.cuda.ExternalStream(st).synchronize()
            got = pkg.finalize_host(host.numpy().tobytes(), cb, W)
            assert got == want, "rows copied on the exported stream"
        if os.environ.get("TE_CHILD_RELEASE_EARLY") == "1":      # (tools/diag_exported_streams.py: the control -- nothing outlives the context)
            del host
            tmp = torch.zeros(1 << 20, dtype=torch.uint8).pin_memory()
            del tmp
            torch.cuda.synchronize()
            continue
        keep.append(host)                                        # outlives the context: released after te_msm_destroy
    if round_ == 0:
        del host
        keep.clear()                                             # released between two contexts: event record on a parked stream
        tmp = torch.zeros(1 << 20, dtype=torch.uint8).pin_memory()   # makes the allocator process its pending events
        del tmp
if os.environ.get("TE_MSM_PARK_STREAMS", "1") != "0":
    # the second context took the first one's parked streams over (the same eight handles, in some order): the pool does not grow
    assert len(all8[0]) == pkg.WORKSETS and all8[0] == all8[1], all8
    s = torch.cuda.ExternalStream(handles[-1])
    s.synchronize()                                              # still a valid stream after the context is gone
print("child ok", len(keep), flush=True)
# `keep` holds a pinned tensor used on an exported stream: it is released at interpreter exit
'''


def _run_child(env_extra=None, timeout=300):
    env = dict(os.environ)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], capture_output=True, text=True, timeout=timeout, env=env)


def test_exported_streams_outlive_nothing(pkg):
    """ExternalStream + pinned non-blocking copy + close + tensor release (between contexts and at interpreter exit): exit code 0"""
    r = _run_child()
    assert "child ok" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])


def test_a_context_that_exported_nothing_destroys_its_streams(pkg):
    """the parking is tied to the export: contexts that never hand a stream out leave nothing behind (created and closed in a row,
    each creates its eight streams anew; 'device_bytes' of a fresh context is zero)"""
    for _ in range(3):
        with pkg.MsmContext((0,)) as c:
            assert c.get_option("device_bytes") == 0
            assert c.get_option("streams_final") in (0, 1)
