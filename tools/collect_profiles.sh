#!/bin/bash
# Collects the rocprofv3 evidence bench.py cites (run on the GPU box from the repo root): tools/collect_profiles.sh TAG [bench args]
#   1. --kernel-trace --stats summary of the default bench command (per-kernel average durations)
#   2. FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (MI355X_MICROARCH.md, HBM section: FETCH_SIZE costs 3 TCC slots,
#      WRITE_SIZE 2; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads -> x2)
#   3. a per-kernel table: necessary bytes (planner, from the bench line) vs PMC bytes
set -e
TAG=${1:-r02}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/prof_stats /tmp/prof_f /tmp/prof_w
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o st -- python3 bench.py --no-cpu-baseline --no-variants "$@" > gpurun_out/${TAG}_bench_under_rocprof.json 2> gpurun_out/${TAG}_stats.err
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_f -o f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants "$@" > gpurun_out/${TAG}_bench_pmc_pass.json 2> gpurun_out/${TAG}_f.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_w -o w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-variants "$@" > /dev/null 2> gpurun_out/${TAG}_w.err
python3 - "$TAG" "$@" <<'PY'
import csv, glob, json, sys, collections
tag = sys.argv[1]
bench = json.loads(open(f"gpurun_out/{tag}_bench_under_rocprof.json").read().strip().splitlines()[-1])
stage_of = {"coarse_premix_kernel": "coarse_premix", "coarse_fwd_kernel": "coarse_fwd", "coarse_mac_kernel": "coarse_mac", "coarse_mfma16_kernel": "coarse_mac", "coarse_sum_kernel": "coarse_mac", "coarse_inv_kernel": "coarse_inv", "coarse_hist_kernel": "coarse_hist",
            "mix_kernel": "mix", "rfft_fwd_b_kernel": "rfft_fwd", "hist_copy_b_kernel": "rfft_fwd", "tconv16_kernel": "mac", "irfft_ola_b_kernel": "rfft_inv"}
raw = {}
for name, d in (("FETCH_SIZE", "/tmp/prof_f"), ("WRITE_SIZE", "/tmp/prof_w")):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name:
            per[r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].replace("ga::", "")].append(float(r["Counter_Value"]))
    for k, v in per.items():
        raw.setdefault(k, {})[name] = v
stats = {}
for r in csv.DictReader(open(f"gpurun_out/{tag}_kernel_stats.csv")):
    k = r["Name"].split("(")[0].replace("void ", "").split("<")[0].replace("ga::", "")
    s = stats.setdefault(k, {"calls": 0, "total_ns": 0.0})
    s["calls"] += int(r["Calls"]); s["total_ns"] += float(r["TotalDurationNs"])
out = {"command": "python3 bench.py --no-cpu-baseline --no-variants " + " ".join(sys.argv[2:]), "units": "FETCH_SIZE / WRITE_SIZE as rocprofv3 reports them: kilobytes (x1024 B) per dispatch",
       "fetch_correction": "MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE reports exactly half the bytes of a wide coalesced streaming read (16 B/lane) -> x2; "
                           "other widths are uncalibrated (coarse_hist_kernel reads 4 B/lane, coarse_fwd_kernel 8 B/lane: see calibration column)",
       "kernels": {}}
for k, st in stats.items():
    if k not in raw and k not in stage_of:
        continue
    if st["total_ns"] < 1e5:
        continue
    e = {"calls": st["calls"], "avg_ms": st["total_ns"] / st["calls"] / 1e6}
    fv = raw.get(k, {}).get("FETCH_SIZE", []); fv = [x for x in fv if x > 0.5 * max(fv)]   # (setup launches of the same kernel are tiny)
    wv = raw.get(k, {}).get("WRITE_SIZE", []); wv = [x for x in wv if x > 0.5 * max(wv)]
    if fv: e["fetch_raw_gb_per_launch"] = sum(fv) / len(fv) * 1024 / 1e9
    if fv: e["fetch_x2_gb_per_launch"] = 2 * e["fetch_raw_gb_per_launch"]
    if wv: e["write_gb_per_launch"] = sum(wv) / len(wv) * 1024 / 1e9
    stg = stage_of.get(k)
    if stg and stg in bench.get("stages", {}):
        s = bench["stages"][stg]
        if s.get("necessary_gb_per_step"):
            e["necessary_gb_per_launch (stage, planner)"] = s["necessary_gb_per_step"] / s["launches_per_step"]
            if "fetch_x2_gb_per_launch" in e and "write_gb_per_launch" in e:
                tot = e["fetch_x2_gb_per_launch"] + e["write_gb_per_launch"]
                e["pmc_total_x2_gb"] = tot
                e["pmc_over_necessary"] = tot / e["necessary_gb_per_launch (stage, planner)"]
                e["tb_per_s_on_pmc_bytes"] = tot / e["avg_ms"]
                e["tb_per_s_on_necessary_bytes"] = e["necessary_gb_per_launch (stage, planner)"] / e["avg_ms"]
    out["kernels"][k] = e
json.dump(out, open(f"gpurun_out/{tag}_pmc_hbm_traffic.json", "w"), indent=1)
for k, e in out["kernels"].items():
    print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in e.items()})
PY
