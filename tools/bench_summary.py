#!/usr/bin/env python3
"""readable summary of a bench.py JSON line (file argument or stdin)"""
import json, sys
d = json.loads((open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin).read().strip().splitlines()[-1])
def stages(st, ind="  "):
    for k, v in st.items():
        print(f"{ind}{k:14s} {v['kernel'][:34]:34s} ms {v['ms_per_step']:.4f}  GB {v['necessary_gb_per_step'] or 0:.3f}  hbm {v['frac_of_hbm_peak'] or 0:.3f}  f32 {v['frac_of_f32_peak'] or 0:.3f}")
print(f"value {d['value']/1e6:.1f} M frames/s  ms/step {d['ms_per_step']:.4f}  host issue {d['host_issue_ms_per_step']:.4f}  device {d['device_ms_per_step']:.4f}")
r = d["roofline"]
print(f"roofline: {r['kernel']} frac {r['frac']:.3f} ({r['achieved']:.0f} GB/s) flops_frac {r['flops_frac']}")
stages(d["stages"])
if d.get("parity"):
    p = d["parity"]
    if "timed_step_rms_vs_f64" in p: print("timed step vs f64:", p["timed_step_rms_vs_f64"]["rms_abs"], "rel", p["timed_step_rms_vs_f64"]["rms_relative_to_bus"])
    if "rms_abs" in p: print("vs oracle:", p["rms_abs"], "rel", p["rms_relative_to_bus"])
if d.get("cpu_baseline"):
    print("cpu 1 thread", d["cpu_baseline"]["value"], "all cores", d["cpu_baseline"]["all_cores"]["value"], "speedups", d.get("speedup_vs_cpu_1thread"), d.get("speedup_vs_cpu_all_cores"))
for n, v in d.get("variants", {}).items():
    t = v.get("timed_step_rms_vs_f64")
    print(f"{n}: ms/step {v['ms_per_step']:.4f}  {v['frames_per_s']/1e6:.1f} M frames/s  host {v['host_issue_ms_per_step']:.3f} dev {v['device_ms_per_step']:.3f}  f64 err {t['rms_abs'] if t else None}")
    r = v["roofline"]
    print(f"   dominant {r['kernel']} ms {r['avg_launch_ms']:.4f} hbm {r['frac']:.3f} f32 {r['flops_frac']}")
    stages(v["stages"], "     ")
