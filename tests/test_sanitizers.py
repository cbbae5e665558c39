"""Sanitizer builds of the host-side code inside the CPU suite (SURVEY.md section 5 "race detection / sanitizers"; the
reference has jest only).  tests/sanitize/Makefile holds the recipes:
  asan-quick  -fsanitize=address,undefined: the device arithmetic headers compiled for the host, every form of the host tails,
              the input synthesis and the two C oracles, each result compared with the oracle (tests/csrc/san_te.cpp, san_377.cpp)
  tsan        -fsanitize=thread: worker_t, the ticket bookkeeping (csrc/host_sched.hpp), the N-API addon's lock protocol
              (js/promise_protocol.hpp) and the multi-thread row merge, with a stand-in device (tests/csrc/sched_harness.cpp)
A sanitizer report aborts the program (halt_on_error): the make target fails.  `make -C tests/sanitize` also runs the two timing
programs under tools/ the same way (minutes; logs of one such run are profiles/r05_sanitize_*.log)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make(target, timeout):
    if not shutil.which("g++") or not shutil.which("make"):
        pytest.skip("no g++ / make")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "sanitize"), target], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
    out = r.stdout.decode(errors="replace")
    if r.returncode != 0 and ("unexpected memory mapping" in out or "ThreadSanitizer: unsupported" in out):
        pytest.skip("ThreadSanitizer cannot map its shadow memory on this kernel")
    assert r.returncode == 0, out[-4000:]
    assert "Sanitizer" not in out, out[-4000:]
    return out


def test_address_and_undefined_behaviour_sanitizers():
    out = _make("asan-quick", 900)
    assert "san_te: all checks passed" in out and "san_377: all checks passed" in out


def test_thread_sanitizer_harness():
    out = _make("tsan", 900)
    assert "sched_harness: all checks passed" in out
