#!/usr/bin/env python3
"""one-line summary of a bench.py JSON line read from stdin: ms per step and per-stage times"""
import json, sys
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print("ms/step %.3f device %.3f" % (r["ms_per_step"], r["device_ms_per_step"]),
      {n: round(s["ms_per_step"], 3) for n, s in r["stages"].items()})
