#!/bin/bash
# Short bench lines for a set of workloads (run on the GPU box from the repository root):
#   tools/ab_bench.sh <tag> [workload ...]    -> gpurun_out/r3/<tag>_<workload>.json and a one-line summary each
TAG=$1; shift
WLS=${@:-"1080p_ra_m3 1080p_ra_m8 4k_ra_m8 720p_ld_m12"}
mkdir -p gpurun_out/r3
for w in $WLS; do
    timeout -k 10 200 python3 bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r3/${TAG}_$w.json 2> gpurun_out/r3/${TAG}_$w.err || { tail -5 gpurun_out/r3/${TAG}_$w.err; exit 1; }
    python3 - gpurun_out/r3/${TAG}_$w.json $w <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
print(sys.argv[2], round(d["value"]), "verified", d["verified"], r["kernel"], round(r["frac"], 3), "all", round(r["all_kernels"]["frac"], 3),
      {k[:-7]: round(v["avg_ms"] * 1e3, 1) for k, v in d["kernels"].items()})
PY
done
