#!/bin/bash
# timeline of one bench run: kernels + memory copies (no counters), gaps between consecutive device activities
# tools/trace_gaps.sh TAG [bench args...]
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/trace_gaps
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/trace_gaps -o t -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" > /dev/null 2> gpurun_out/${TAG}_trace.err
python3 - "$TAG" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
ev = []
for f in glob.glob("/tmp/trace_gaps/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"][:40]))
for f in glob.glob("/tmp/trace_gaps/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", r.get("Kind", "?")) ))
ev.sort()
t0 = ev[0][0]
with open(f"gpurun_out/{tag}_trace.txt", "w") as o:
    last_end = None
    for s, e, n in ev[-90:]:
        gap = (s - last_end) / 1e3 if last_end else 0.0
        line = f"{(s - t0) / 1e6:10.3f} ms  dur {(e - s) / 1e3:9.1f} us  gap {gap:8.1f} us  {n}"
        print(line); o.write(line + "\n")
        last_end = max(last_end or 0, e)
PY
