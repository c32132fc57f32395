#!/bin/bash
# SQ counters of the bench's kernels (one --pmc pass, 8 SQ slots): tools/prof_sq.sh TAG [bench args...]
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/prof_sq
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS \
  --output-format csv -d /tmp/prof_sq -o sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2> gpurun_out/${TAG}_sq.err
python3 - "$TAG" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
f = glob.glob("/tmp/prof_sq/**/*counter_collection.csv", recursive=True)[0]
per = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    per[r["Kernel_Name"][:60]][r["Counter_Name"].replace("SQ_", "")].append(float(r["Counter_Value"]))
with open(f"gpurun_out/{tag}_sq.txt", "w") as o:
    for k, c in per.items():
        if "ga::" not in k: continue
        m = {n: sum(v) / len(v) for n, v in c.items()}
        wc = m.get("WAVE_CYCLES", 1)
        line = f"{k:60s} launches {len(c['WAVE_CYCLES']):3d} wave_cycles {wc:.3e} " + " ".join(f"{n} {v / wc:.2f}" for n, v in m.items() if n != "WAVE_CYCLES")
        print(line); o.write(line + "\n")
PY
