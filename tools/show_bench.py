"""prints the figures of a bench.py line that a builder looks at first (the whole line is ~30 KB): python tools/show_bench.py <file>"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
keys = ("value", "ms_per_step", "latency_ms", "host_buffers_ms", "host_buffers_in_flight_ms", "bases_resident_in_flight_ms", "bases_resident_latency_ms",
        "bases_resident_device_scalars_ms", "parity", "n_gpus", "inputs", "input_distribution_ms", "input_distribution_error", "rccl_ranks", "backend")
print({k: d.get(k) for k in keys if d.get(k) is not None})
print("bases_resident:", d.get("bases_resident"))
r = d.get("roofline", {})
print("roofline:", {k: r.get(k) for k in ("achieved", "frac", "traffic", "kernel_ms")}, {k: r.get("binding_roofline", {}).get(k) for k in ("frac", "core_clock_ghz", "frac_at_measured_clock")})
for k, v in sorted((d.get("sizes") or {}).items()):
    print("size", k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items()})
for k, v in (d.get("configs") or {}).items():
    print("config", k, {a: v.get(a) for a in ("value", "ms_per_step", "latency_ms", "parity")}, "bases:", {a: (round(b, 4) if isinstance(b, float) else b) for a, b in (v.get("bases_resident") or {}).items() if a != "note"})
