#!/bin/bash
# rocprofv3 kernel statistics of config 4 at full size on one GPU (tools/run_configs.py 4): which kernels the 12-16 ms per 2.5 s are
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/prof_cfg4
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_cfg4 -o st -- python3 tools/run_configs.py 4 ${1:-10} ${2:-4096} > gpurun_out/prof_cfg4.log 2>&1
cp $(find /tmp/prof_cfg4 -name "*kernel_stats.csv" | head -1) gpurun_out/cfg4_kernel_stats.csv
head -12 gpurun_out/cfg4_kernel_stats.csv
