#!/bin/bash
# Instruction-cache counters of one bench workload (run on the GPU box from the repository root):
#   tools/pmc_icache.sh <workload> <out dir>
set -e
WL=${1:-1080p_ra_m8}
OUT=${2:-gpurun_out/icache_$WL}
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVES --output-format csv -d "$OUT/ic" -o ic -- python3 bench.py --workload "$WL" --steps 4 --warmup 2 --no-cpu-baseline --kernel-steps 0 --search-streams 1 > "$OUT/ic.log" 2>&1
python3 tools/pmc_sum.py "$OUT/ic" "$OUT/pmc_icache.csv" > /dev/null
grep -E "svtme" "$OUT/pmc_icache.csv"
