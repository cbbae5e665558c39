"""bench.py's bookkeeping (no GPU): the algorithmic byte counts of SURVEY.md 8d, the staleness rule of the committed counter
profile and ISA listing, and that the committed evidence files parse."""
import json
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_algorithmic_bytes_match_the_survey():
    # A(2^20) = 1 375 731 776 B, A(2^16) = 211 812 416 B (signed 16-bit windows); unsigned: + 134 217 728
    whole, acc = bench.algorithmic_bytes(1 << 20, 16, 1 << 15)
    assert whole == 1375731776 and acc == 16 * (1 << 20) * 68 + 16 * (1 << 15) * 128 == 1207959552
    assert bench.algorithmic_bytes(1 << 16, 16, 1 << 15)[0] == 211812416
    assert bench.algorithmic_bytes(1 << 20, 16, 1 << 16)[0] == 1375731776 + 134217728
    assert bench.algorithmic_bytes(1 << 20, 16, 1 << 15, bls=True)[0] == 1979711584


def test_sources_hash_ignores_comments_and_white_space(tmp_path, monkeypatch):
    sha = bench.kernel_sources_sha()
    assert len(sha) == 16 and sha == bench.kernel_sources_sha()
    # the same code with another comment hashes the same; another token does not
    src = os.path.join(ROOT, bench.PKG, "csrc")
    fake = tmp_path / bench.PKG / "csrc"
    fake.mkdir(parents=True)
    for f in bench.KERNEL_SOURCES:
        text = open(os.path.join(src, f)).read()
        (fake / f).write_text("// a new comment\n" + text.replace("\n", "\n  ", 3))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.kernel_sources_sha() == sha
    (fake / bench.KERNEL_SOURCES[0]).write_text((fake / bench.KERNEL_SOURCES[0]).read_text() + "\nint x;\n")
    assert bench.kernel_sources_sha() != sha


def test_traffic_figure_is_withheld_when_the_sources_changed(tmp_path, monkeypatch):
    j = json.load(open(bench.TRAFFIC_JSON))
    assert {"kernel_sources_sha", "kernels", "correction"} <= set(j) and "k_accumulate" in j["kernels"]
    val, info = bench.measured_traffic(20, 16, 1)
    if info.get("stale"):
        warnings.warn("profiles/pmc_traffic.json was taken from other kernel sources: re-run tools/final_profiles.sh")
        assert val is None and info["stale_value"] > 1e9
    else:
        assert 1.2e9 < val < 4e9                       # between the algorithmic bytes and 3x of them
    assert bench.measured_traffic(19, 16, 1)[0] is None     # another workload: no figure
    stale = dict(j, kernel_sources_sha="0" * 16)
    p = tmp_path / "t.json"
    p.write_text(json.dumps(stale))
    monkeypatch.setattr(bench, "TRAFFIC_JSON", str(p))
    val, info = bench.measured_traffic(20, 16, 1)
    assert val is None and info["stale"] is True


def test_isa_listing_figures():
    te, bls = bench.isa_cycles(False), bench.isa_cycles(True)
    assert 5000 < te["cycles"] < 8000 and 12000 < bls["cycles"] < 20000
    j = json.load(open(bench.ISA_JSON))
    assert j["k_accumulate<9>"]["mads"] == 7 * 153     # seven field products of 153 multiply-accumulates per accumulated point


def test_roofline_follows_the_accumulated_entries_not_w_times_n():
    """A prover's witness (a quarter zeros, a quarter ones, the rest uniform) has ~8.25 non-zero digits of 16 per scalar: the
    roofline is priced with the entries the engine counted (option "entries_accumulated"), so neither fraction can exceed what
    the kernel did.  Round 4 charged W * n and reported the kernel at 1.49 of its own instruction floor."""
    n, W, B = 1 << 20, 16, 1 << 15
    entries = int(n * (0.25 * 0 + 0.25 * 1 + 0.5 * 16 * (1 - 2.0 ** -16)))        # 8 650 k of the 16 777 k of a uniform set
    whole_u, acc_u = bench.algorithmic_bytes(n, W, B)
    whole_w, acc_w = bench.algorithmic_bytes(n, W, B, entries=entries)
    assert acc_w == entries * 68 + W * B * 128 and whole_w == 96 * n + entries * 68 + 2 * W * B * 128 + 64
    assert 0.5 < acc_w / acc_u < 0.56
    # the driver's round-4 figures for this config: the kernel alone 0.440 ms at 2.0 GHz
    alone = {"accumulate": 0.440, "accumulate_core_clock_ghz": 2.0}
    r = bench.roofline_block(acc_w, alone, None, False, entries / 64.0, entries=entries)
    assert r["entries_accumulated_per_launch"] == entries and r["algorithmic_bytes_per_launch"] == acc_w
    assert 0.15 < r["frac"] < 0.22                                                  # HBM: ~0.19, not 0.34
    v = r["binding_roofline"]
    assert v["frac"] < 1.0 and 0.8 < v["frac_at_measured_clock"] < 1.0             # VALU issue: ~0.9, not 1.49 / 1.79
    wrong = bench.roofline_block(acc_u, alone, None, False, W * n / 64.0)
    assert wrong["binding_roofline"]["frac_at_measured_clock"] > 1.5               # what charging W * n gives
    # a uniform set: the count changes nothing beyond 2^-16
    assert abs(bench.algorithmic_bytes(n, W, B, entries=int(W * n * (1 - 2.0 ** -16)))[1] / acc_u - 1) < 1e-4


# ------------------------------------------------------------------ `python bench.py --gpus N` without a launcher
_STUB = r'''
import os, sys, time
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["LOCAL_RANK"] == str(rank) and os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
mode = sys.argv[1]
if mode == "ok":
    time.sleep(0.2 * rank)
    print('{"rank": %d, "world": %d}' % (rank, world), flush=True)
elif mode == "fail1":
    if rank == 1:
        time.sleep(0.3)
        sys.exit(7)
    print("rank %d waits in its collective" % rank, flush=True)
    time.sleep(600)                      # the other ranks would sit in a collective until their own timeout
elif mode == "hang":
    time.sleep(600)
'''


def _stub(tmp_path):
    p = tmp_path / "rank_stub.py"
    p.write_text(_STUB)
    return str(p)


def test_launch_ranks_relays_rank0_and_returns_zero(tmp_path, capfd):
    import time
    t0 = time.time()
    assert bench.launch_ranks(3, [sys.executable, _stub(tmp_path), "ok"], timeout=60) == 0
    out, err = capfd.readouterr()
    assert out.strip() == '{"rank": 0, "world": 3}'                  # ONE line on stdout: rank 0's
    assert '[rank 1] {"rank": 1, "world": 3}' in err and '[rank 2] {"rank": 2, "world": 3}' in err
    assert time.time() - t0 < 30


def test_launch_ranks_stops_everybody_when_one_rank_fails(tmp_path, capfd):
    """a rank that exits non-zero ends the job with its code within seconds; the ranks left in their 'collective' are killed"""
    import time
    t0 = time.time()
    rc = bench.launch_ranks(2, [sys.executable, _stub(tmp_path), "fail1"], timeout=120)
    took = time.time() - t0
    assert rc == 7 and took < 30, (rc, took)
    out, err = capfd.readouterr()
    assert "rank 1 exited with 7" in err and "rank 0 waits" in out


def test_launch_ranks_times_out(tmp_path):
    import time
    t0 = time.time()
    assert bench.launch_ranks(2, [sys.executable, _stub(tmp_path), "hang"], timeout=1.5) == 124
    assert time.time() - t0 < 30


def test_bench_without_a_launcher_starts_its_ranks_before_torch(tmp_path):
    """`python bench.py --gpus 2` with no WORLD_SIZE: the parent never imports torch (it would touch HIP on a GPU box) -- checked by
    making `import torch` fatal in the PARENT only; the children fail at once on this box (no GPU) and the parent reports it"""
    import subprocess
    site = tmp_path / "site"
    site.mkdir()
    (site / "torch.py").write_text("import os, sys\nif 'WORLD_SIZE' not in os.environ:\n    sys.exit(99)\nraise SystemExit(5)\n")
    env = dict(os.environ, PYTHONPATH=str(site) + os.pathsep + os.environ.get("PYTHONPATH", ""), TE_BENCH_LAUNCH_TIMEOUT="60")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 5, (r.returncode, r.stderr[-800:])       # the children's code, not the parent's 99
    assert "stopping the other ranks" in r.stderr
