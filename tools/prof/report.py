#!/usr/bin/env python3
"""tools/prof/report.py SAMPLES [LIB]: per-function histograms (innermost library frame, and inclusive) of a sigprof run"""
import collections, subprocess, sys
samples = [l.split() for l in open(sys.argv[1]) if not l.startswith("#")]
lib = sys.argv[2] if len(sys.argv) > 2 else "graphaudio_amd/libgraphaudio_hip.so"
offs = sorted({o for s in samples for o in s if o != "-"})
if not offs:
    sys.exit("no samples inside the library")
import os
tool = "/opt/rocm/lib/llvm/bin/llvm-addr2line" if os.path.exists("/opt/rocm/lib/llvm/bin/llvm-addr2line") else "addr2line"
out = subprocess.run([tool, "-f", "-C", "-e", lib] + ["0x" + o for o in offs], capture_output=True, text=True).stdout.splitlines()
name = {o: out[2 * i][:90] for i, o in enumerate(offs)}
line = {o: out[2 * i + 1].split("/")[-1].split(" ")[0] for i, o in enumerate(offs)}   # file:line of the innermost (inlined) code
leaf, incl = collections.Counter(), collections.Counter()
inlib = 0
for s in samples:
    if s == ["-"]:
        continue
    inlib += 1
    leaf[name[s[0]]] += 1
    for fn in {name[o] for o in s}:
        incl[fn] += 1
print(f"{len(samples)} samples, {inlib} inside the library")
print("-- inclusive --")
for fn, c in incl.most_common(30):
    print(f"{c:6d} {100.0 * c / max(inlib, 1):5.1f}%  {fn}")
print("-- innermost library frame --")
for fn, c in leaf.most_common(30):
    print(f"{c:6d} {100.0 * c / max(inlib, 1):5.1f}%  {fn}")
print("-- innermost source lines --")
lines = collections.Counter()
for s_ in samples:
    if s_ != ["-"]:
        lines[line[s_[0]] + "  in " + name[s_[0]][:60]] += 1
for ln, c in lines.most_common(45):
    print(f"{c:6d} {100.0 * c / max(inlib, 1):5.1f}%  {ln}")
