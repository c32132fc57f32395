#!/bin/bash
# Collects the rocprofv3 evidence bench.py cites (run on the GPU box from the repo root):
#   1. --kernel-trace --stats summary of the default bench
#   2. FETCH_SIZE and WRITE_SIZE in separate --pmc passes (MI355X_MICROARCH.md, HBM section)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r01_v4}
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o st -- python3 bench.py > gpurun_out/${TAG}_bench_under_rocprof.json 2> gpurun_out/${TAG}_stats.err
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_f -o f -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/${TAG}_f.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_w -o w -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/${TAG}_w.err
python3 - "$TAG" <<'PY'
import csv, glob, json, sys, collections
tag = sys.argv[1]
out = {"kernels": {}}
for name, d in (("FETCH_SIZE", "/tmp/prof_f"), ("WRITE_SIZE", "/tmp/prof_w")):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            per[r["Kernel_Name"].split("(")[0]].append(round(float(r["Counter_Value"]), 1))
    for k, v in per.items():
        if max(v) > 1e5:
            out["kernels"].setdefault(k, {})[name + "_KB_per_launch"] = v
json.dump(out, open(f"gpurun_out/{tag}_pmc_hbm_raw.json", "w"), indent=1)
print(json.dumps(out)[:1500])
PY
