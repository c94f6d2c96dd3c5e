#!/usr/bin/env python3
"""One-line summary of bench.py JSON files:  python tools/bench_line.py file.json [...]"""
import json
import sys

for path in sys.argv[1:]:
    try:
        d = json.load(open(path))
    except Exception as exc:  # noqa: BLE001
        print(path, "unreadable:", exc)
        continue
    cb = d.get("cpu_baseline", {})
    print(f"{path}: {d['value']} {d['unit']}  step {d['ms_per_step']} ms  kernels {d['roofline']['kernel_ms_avg_each']}  "
          f"dp {d['exact_dp_fraction']} refl {d['refiltered_fraction']}  identical {cb.get('gpu_results_identical_on_sample')}")
