#!/bin/bash
# effective shader clock per kernel: GRBM_GUI_ACTIVE / 8 XCDs / kernel duration (MI355X_MICROARCH.md, DVFS give-back)
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/prof_clk
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/prof_clk -o c -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2> gpurun_out/${TAG}_clk.err
python3 - "$TAG" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
f = glob.glob("/tmp/prof_clk/**/*counter_collection.csv", recursive=True)[0]
k = glob.glob("/tmp/prof_clk/**/*kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(k)):
    dur[r["Dispatch_Id"]] = (r["Kernel_Name"][:50], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
per = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] in dur:
        name, ns = dur[r["Dispatch_Id"]]
        if ns > 50000 and "ga::" in name:
            per[name].append(float(r["Counter_Value"]) / 8.0 / ns)
with open(f"gpurun_out/{tag}_clk.txt", "w") as o:
    for n, v in per.items():
        line = f"{n:52s} launches {len(v):3d} effective clock {sum(v)/len(v):.2f} GHz"
        print(line); o.write(line + "\n")
PY
