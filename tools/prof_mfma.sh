#!/bin/bash
# matrix-core and memory counters of one bench variant: tools/prof_mfma.sh TAG VARIANT
set -e
TAG=$1; V=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/prof_mf /tmp/prof_mf2
rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 \
  --output-format csv -d /tmp/prof_mf -o m -- python3 bench.py --only-variant $V --no-check --variant-steps 2 > /dev/null 2> gpurun_out/${TAG}_mf.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_mf2 -o f -- python3 bench.py --only-variant $V --no-check --variant-steps 2 > /dev/null 2>> gpurun_out/${TAG}_mf.err
python3 - "$TAG" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
with open(f"gpurun_out/{tag}_mf.txt", "w") as o:
    for d in ("/tmp/prof_mf", "/tmp/prof_mf2"):
        f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
        per = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            per[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, c in per.items():
            if "coarse" not in k: continue
            line = f"{k:48s} " + " ".join(f"{n} {sum(v) / len(v):.4g}" for n, v in c.items())
            print(line); o.write(line + "\n")
PY
