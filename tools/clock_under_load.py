"""What holds the shader clock at 1.97-2.18 GHz under k_accumulate -- the power cap or the DPM state?  (round-5 verdict, item 6)

Samples the GPU's sysfs sensors at 10 Hz -- shader clock (hwmon freq1_input / pp_dpm_sclk), socket power (power1_average or
power1_input) and cap (power1_cap), temperatures, busy percent -- through five phases on ONE box:
  idle | MSMs in flight (the engine's pipelined loop: k_accumulate runs ~87 % of the time) | idle |
  tools/clock_load fma32 (full-rate FP32 FMAs) | tools/clock_load mad (the engine's field product alone) at 4 / 3 / 2 waves per SIMD
and takes one `rocm-smi --showpower --showclocks --showperflevel --showmaxpower` snapshot idle and one under the MSM load.
The kernels report the mean core clock their own waves saw (shader-clock / wall-clock ticks): the sysfs figure and that one
should agree.  python tools/clock_under_load.py > profiles/r06_clock_under_load_raw.txt   (needs gpurun_out/clock_load built)"""
import ctypes, glob, importlib, os, subprocess, sys, threading, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("webgpu-msm-twisted-edwards_amd")
import torch


def pci_bus_id():
    try:
        return torch.cuda.get_device_properties(0).pci_bus_id, torch.cuda.get_device_properties(0).pci_device_id
    except Exception:
        return None, None


def find_card():
    """sysfs directory of HIP device 0 (by PCI bus id when torch reports one; else the only / first amdgpu card)"""
    cards = []
    for d in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        try:
            if open(os.path.join(d, "vendor")).read().strip() != "0x1002":
                continue
        except OSError:
            continue
        cards.append(d)
    want = None
    try:
        buf = ctypes.create_string_buffer(64)
        hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
        if hip.hipDeviceGetPCIBusId(buf, 64, 0) == 0:
            want = buf.value.decode().lower()
    except Exception:
        pass
    for d in cards:
        if want and os.path.realpath(d).lower().endswith(want):
            return d, want, cards
    return (cards[0] if cards else None), want, cards


def rd(path):
    try:
        return open(path).read().strip()
    except OSError:
        return None


class Sampler(threading.Thread):
    def __init__(self, card):
        super().__init__(daemon=True)
        self.card, self.rows, self.stop_flag, self.phase = card, [], False, "start"
        self.hw = sorted(glob.glob(os.path.join(card, "hwmon", "hwmon*"))) if card else []
        self.files = {}
        for h in self.hw:
            for name in ("freq1_input", "freq2_input", "power1_average", "power1_input", "power1_cap", "temp1_input", "temp2_input", "temp3_input"):
                p = os.path.join(h, name)
                if os.path.exists(p):
                    self.files[name] = p
        for name in ("gpu_busy_percent", "pp_dpm_sclk", "pp_dpm_mclk", "power_dpm_force_performance_level"):
            p = os.path.join(card, name) if card else None
            if p and os.path.exists(p):
                self.files[name] = p

    def run(self):
        t0 = time.perf_counter()
        while not self.stop_flag:
            row = {"t": time.perf_counter() - t0, "phase": self.phase}
            for k, p in self.files.items():
                v = rd(p)
                if v is None:
                    continue
                if k.startswith("pp_dpm"):
                    act = [ln for ln in v.splitlines() if ln.strip().endswith("*")]
                    row[k] = act[0].strip() if act else v.replace("\n", " | ")
                else:
                    row[k] = v
            self.rows.append(row)
            time.sleep(0.1)


def smi(label):
    for cmd in (["rocm-smi", "--showpower", "--showclocks", "--showperflevel", "--showmaxpower"], ["amd-smi", "metric", "-p", "-c"]):
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=60)
            print("---- %s: %s (rc %d)" % (label, " ".join(cmd), r.returncode))
            print("\n".join(ln for ln in (r.stdout + r.stderr).splitlines() if ln.strip())[-3000:])
        except Exception as e:          # the tool is missing or not permitted: say so and go on
            print("---- %s: %s failed: %r" % (label, cmd[0], e))
    sys.stdout.flush()


def main():
    card, bus, cards = find_card()
    print("HIP device 0: PCI %s; amdgpu cards in sysfs: %d; sampling %s" % (bus, len(cards), card))
    s = Sampler(card)
    print("sensors:", {k: v for k, v in s.files.items()})
    s.start()
    results = []
    smi("idle")
    s.phase = "idle"; time.sleep(1.5)
    # ---- the engine: MSMs in flight for ~3 s
    n = 1 << 20
    pts, sc = pkg.synth_inputs(0xC10C, n, fixed_point="chain")
    dp = torch.frombuffer(bytearray(pts), dtype=torch.uint8).cuda(); ds = torch.frombuffer(bytearray(sc), dtype=torch.uint8).cuda()
    torch.cuda.synchronize()
    with pkg.MsmContext((0,)) as c:
        c.set_option("profile", 1)
        for _ in range(8):
            c.run_device(dp.data_ptr(), ds.data_ptr(), n)
        smi_thread = threading.Thread(target=smi, args=("under the MSM load",))
        s.phase = "msm_in_flight"
        t0 = time.perf_counter(); tk = []; done = 0; clocks = []; started_smi = False
        while time.perf_counter() - t0 < 4.0:
            tk.append(c.submit_device(dp.data_ptr(), ds.data_ptr(), n))
            if len(tk) >= 4:
                c.collect(tk.pop(0)); done += 1
                if done % 50 == 0:
                    clocks.append(c.stage_ms().get("accumulate_core_clock_ghz"))
            if not started_smi and time.perf_counter() - t0 > 1.0:
                smi_thread.start(); started_smi = True
        while tk:
            c.collect(tk.pop(0)); done += 1
        el = time.perf_counter() - t0
        results.append("msm_in_flight: %d MSMs in %.2f s = %.1f MSM/s; k_accumulate's own core clock (GHz, every 50th MSM): %s"
                       % (done, el, done / el, " ".join("%.3f" % x for x in clocks if x)))
        # one MSM at a time as well: the accumulation alone
        s.phase = "msm_one_at_a_time"
        t0 = time.perf_counter(); acc = []; clk1 = []
        while time.perf_counter() - t0 < 2.0:
            c.run_device(dp.data_ptr(), ds.data_ptr(), n)
            st = c.stage_ms(); acc.append(st.get("accumulate")); clk1.append(st.get("accumulate_core_clock_ghz"))
        results.append("msm_one_at_a_time: accumulate %.3f ms mean (min %.3f), its core clock %.3f GHz mean" % (sum(acc) / len(acc), min(acc), sum(clk1) / len(clk1)))
        if started_smi:
            smi_thread.join()
    s.phase = "idle2"; time.sleep(1.5)
    exe = os.path.join(ROOT, "gpurun_out", "clock_load")
    for kind, wps in (("fma32", 4), ("mad", 4), ("mad", 3), ("mad", 2)):
        s.phase = "%s_w%d" % (kind, wps)
        r = subprocess.run([exe, kind, "3.0", str(wps)], capture_output=True, text=True, timeout=120)
        results.append((r.stdout.strip() or r.stderr.strip()[-300:]))
        s.phase = "gap"; time.sleep(0.7)
    s.stop_flag = True; s.join()
    print("\n==== workloads")
    for r in results:
        print(r)
    # per-phase summary of the samples
    print("\n==== sysfs samples by phase (10 Hz): n, shader clock MHz (min / mean / max), power W (min / mean / max), cap W, busy %, temp")
    def num(v, scale=1.0):
        try:
            return float(v) / scale
        except (TypeError, ValueError):
            return None
    phases = []
    for r in s.rows:
        if r["phase"] not in phases:
            phases.append(r["phase"])
    for ph in phases:
        rows = [r for r in s.rows if r["phase"] == ph]
        f = [x for x in (num(r.get("freq1_input"), 1e6) for r in rows) if x is not None]
        pw = [x for x in (num(r.get("power1_average") or r.get("power1_input"), 1e6) for r in rows) if x is not None]
        cap = num(rows[0].get("power1_cap"), 1e6)
        busy = [x for x in (num(r.get("gpu_busy_percent")) for r in rows) if x is not None]
        tmp = [x for x in (num(r.get("temp2_input") or r.get("temp1_input"), 1e3) for r in rows) if x is not None]
        fmt = lambda a: ("%.0f / %.0f / %.0f" % (min(a), sum(a) / len(a), max(a))) if a else "-"
        print("%-20s n=%3d  sclk %s  power %s  cap %s  busy %s  temp %s  dpm %s" % (ph, len(rows), fmt(f), fmt(pw), cap, fmt(busy), fmt(tmp), rows[len(rows) // 2].get("pp_dpm_sclk")))
    print("\n==== raw samples (t, phase, sclk MHz, power W, busy, dpm)")
    for r in s.rows:
        print("%.2f %s %s %s %s %s" % (r["t"], r["phase"], num(r.get("freq1_input"), 1e6), num(r.get("power1_average") or r.get("power1_input"), 1e6), r.get("gpu_busy_percent"), r.get("pp_dpm_sclk")))


if __name__ == "__main__":
    main()
