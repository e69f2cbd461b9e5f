#!/usr/bin/env python3
"""The figures of a bench line a reader looks at first:  python scripts/show_line.py <bench_line.json>"""
import json
import sys

d = json.load(open(sys.argv[1]))
r = d["roofline"]
print(f"value {d['value']:.3f} {d['unit']}  ms_per_step {d['ms_per_step']:.2f}")
for key in ("roofline", "roofline_adjoint", "roofline_marginals", "roofline_other_solves"):
    o = d.get(key)
    if o:
        print(f"{key:24s} {o['ms_per_step']:7.2f} ms/step  frac {o['frac']:.4f}  traffic/alg {o.get('traffic_over_alg')}  "
              f"fma_frac {o.get('fp64_fma_frac')}  valu/64 states {o.get('valu_wave_insts_per_64_states')}  waves/SIMD {o.get('waves_per_simd')}")
print("eval_traffic", d.get("eval_traffic"))
print("eval_floor", {k: v for k, v in d.get("eval_floor", {}).items() if k != "B_pat_equivalent"})
kv = d.get("kronvec", {})
for k, v in kv.items():
    print(f"{k:12s} {v['ms_per_launch']:.4f} ms  frac_of_peak {v['frac_of_peak']:.3f}  live {v['frac_of_peak_live']:.3f}  moved {v.get('frac_of_peak_moved')}")
print("small_cohort", d.get("small_cohort"))
print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline_optimised", {}).get("value"), "max_rel_err", d.get("max_rel_err"))
print("stream", d.get("measured_stream"))
