#!/bin/bash
# per-kernel times of the default bench under rocprofv3 (run on the GPU box from the repo root): tools/kstats.sh TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-ks}
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o st -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}.err
cp $(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_kernel_stats.csv
python3 - gpurun_out/${TAG}_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["TotalDurationNs"]) > 2e5: print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e6:8.3f} ms')
PY
