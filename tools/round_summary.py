#!/usr/bin/env python3
"""One screen of what tools/profile_round.sh left in gpurun_out/round/ (before the files are copied to profiles/)."""
import json
from pathlib import Path

R = Path(__file__).resolve().parents[1] / "gpurun_out" / "round"
for f in ("bench", "bench_serial", "bench_config2", "bench_config4", "bench_config5"):
    d = json.load(open(R / f"{f}.json"))
    r = d["roofline"]
    print(f, d["value"], d["steps"], d["ms_per_step"], r["frac"], r.get("kernel_ms_avg_each"), r.get("traffic"),
          d.get("exact_dp_fraction"), d.get("refiltered_fraction"), d.get("full_size_launch_equals_piecewise_launches"))
d = json.load(open(R / "bench.json"))
v = d.get("roofline_valu") or {}
print({k: v.get(k) for k in ("frac", "frac_at_mix_ceiling", "valu_insts_per_64_pair_tile", "clock_hz",
                             "scan_wait_any_over_wave_cycles", "scan_wait_inst_any_over_wave_cycles", "valu_insts_per_launch")})
print({k: (t.get("M_pairs_per_s"), t.get("first_run_M_pairs_per_s")) for k, t in d.get("tiers", {}).items()
       if isinstance(t, dict) and "M_pairs_per_s" in t})
print(d["cpu_baseline"]["value"], r.get("copy_measured_GBps"), d["roofline"].get("frac_of_copy_measured"), d["roofline"]["frac_scan_kernel"])
print(json.load(open(R / "e2e.json")))
print(open(R / "kernel_stats.csv").read().split("\n")[1][:170])
print(open(R / "kernel_stats.csv").read().split("\n")[2][:170])
