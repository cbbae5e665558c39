"""Diagnosis of round 5's abort (profiles/r05_batch_small_experiment.txt: "timeout: the monitored command dumped core"):
the stream-sharing recipe of tests/test_gpu_stream_export.py in a child process, once with the old behaviour
(TE_MSM_PARK_STREAMS=0: te_msm_destroy destroys exported streams) and once with the fix (parked).  Prints exit code and the
tail of stderr of both -- the old form names its own cause.  One run each; nothing is repeated.
python tools/diag_exported_streams.py > profiles/r06_batch_small_abort_raw.txt"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_stream_export import _run_child

for label, env in (("old behaviour: TE_MSM_PARK_STREAMS=0 (exported streams destroyed by te_msm_destroy), a pinned tensor used on one outlives the context", {"TE_MSM_PARK_STREAMS": "0"}),
                   ("control: TE_MSM_PARK_STREAMS=0, but every pinned tensor is released BEFORE the context closes", {"TE_MSM_PARK_STREAMS": "0", "TE_CHILD_RELEASE_EARLY": "1"}),
                   ("fix: exported streams parked, a pinned tensor outlives the context", {})):
    r = _run_child(env)
    print("==== %s" % label)
    print("exit code %d%s" % (r.returncode, " (killed by signal %d)" % -r.returncode if r.returncode < 0 else ""))
    print("stdout tail:", r.stdout[-300:].strip())
    print("stderr tail:")
    print("\n".join(r.stderr.strip().splitlines()[-25:]))
    sys.stdout.flush()
