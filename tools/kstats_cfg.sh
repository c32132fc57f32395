#!/bin/bash
# per-kernel times of one of the other configurations under rocprofv3 (GPU box, repo root): tools/kstats_cfg.sh TAG <run_configs args>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
rm -rf /tmp/prof_$TAG
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o st -- python3 tools/run_configs.py "$@" > gpurun_out/${TAG}_run.log 2> gpurun_out/${TAG}.err
cp $(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_kernel_stats.csv
tail -4 gpurun_out/${TAG}_run.log
python3 - gpurun_out/${TAG}_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["TotalDurationNs"]) > 1e5: print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e6:8.3f} ms total {float(r["TotalDurationNs"])/1e6:8.2f} ms')
PY
