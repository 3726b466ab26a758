#!/bin/bash
# Collects the profiles committed under profiles/ for one bench workload (run on the GPU box from the repository root):
#   tools/collect_profiles.sh <workload> <out dir>
# kernel trace with one and with three search streams, then one --pmc pass per counter group (each in a run of its own).
set -e
WL=${1:-1080p_ra_m3}
OUT=${2:-gpurun_out/prof_$WL}
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--workload $WL --steps 4 --warmup 2 --no-cpu-baseline --kernel-steps 0"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/one" -o one -- python3 bench.py --workload "$WL" --steps 20 --warmup 4 --no-cpu-baseline --kernel-steps 0 --search-streams 1 > "$OUT/one.log" 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/three" -o three -- python3 bench.py --workload "$WL" --steps 20 --warmup 4 --no-cpu-baseline --kernel-steps 0 --search-streams 3 > "$OUT/three.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o fetch -- python3 bench.py $ARGS --search-streams 1 > "$OUT/fetch.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o write -- python3 bench.py $ARGS --search-streams 1 > "$OUT/write.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU --output-format csv -d "$OUT/sq_wave" -o sq_wave -- python3 bench.py $ARGS --search-streams 1 > "$OUT/sq_wave.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq_insts" -o sq_insts -- python3 bench.py $ARGS --search-streams 1 > "$OUT/sq_insts.log" 2>&1
for d in fetch write sq_wave sq_insts; do python3 tools/pmc_sum.py "$OUT/$d" "$OUT/pmc_$d.csv" > /dev/null; done
BATCH=$(grep '^{' "$OUT/fetch.log" | tail -1 | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['pictures_per_submit'])")
python3 tools/make_traffic.py "$WL" "$OUT/pmc_fetch.csv" "$OUT/pmc_write.csv" "$OUT/traffic_$WL.json" "$BATCH"
cp "$OUT"/one/*kernel_stats.csv "$OUT/kernel_stats_one_stream.csv"
cp "$OUT"/three/*kernel_stats.csv "$OUT/kernel_stats_three_streams.csv"
echo collected
