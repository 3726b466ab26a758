#!/bin/bash
# Re-runs bench.py for every workload and writes the JSON lines under profiles/ (run on the GPU box from the repository root).
set -e
R=${1:-r03}
ONLY=${2:-} # optional: a regular expression, only the runs whose name matches it
run() { # name, args...
    local name=$1; shift
    if [ -n "$ONLY" ] && ! [[ "$name" =~ $ONLY ]]; then return 0; fi
    timeout -k 10 600 python3 bench.py "$@" > "profiles/${R}_bench_${name}.json.tmp" 2> "gpurun_out/refresh_${name}.err"
    tail -1 "profiles/${R}_bench_${name}.json.tmp" > "profiles/${R}_bench_${name}.json"
    rm "profiles/${R}_bench_${name}.json.tmp"
    python3 -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d['value']), d['roofline']['kernel'], round(d['roofline']['frac'],3), round(d['roofline']['all_kernels']['achieved'],1), (d.get('cpu_baseline') or {}).get('value'))" "profiles/${R}_bench_${name}.json" "$name"
}
mkdir -p gpurun_out
run 1080p_ra_m3
run 1080p_ra_m3_tail1 --device-tail
run 1080p_ra_m3_tail2 --device-tail-only
run 1080p_ra_m8 --workload 1080p_ra_m8
run 720p_ld_m12 --workload 720p_ld_m12
run 4k_ra_m8 --workload 4k_ra_m8
run 1080p_tf_l0 --workload 1080p_tf_l0
run 1080p_tf_l2 --workload 1080p_tf_l2
run 1080p_first_pass --workload 1080p_first_pass
run 1080p_first_pass_fast --workload 1080p_first_pass_fast
run 8k_ra_m2 --workload 8k_ra_m2 --steps 10 --warmup 2 --kernel-steps 2
# the same with every source picture put into the device store inside the timed region (zero-copy puts from page-locked
# buffers with the encoder's stride); these lines carry no CPU baseline
run 1080p_ra_m3_with_upload --with-upload --no-cpu-baseline
run 1080p_ra_m8_with_upload --workload 1080p_ra_m8 --with-upload --no-cpu-baseline
run 4k_ra_m8_with_upload --workload 4k_ra_m8 --with-upload --no-cpu-baseline
run 720p_ld_m12_with_upload --workload 720p_ld_m12 --with-upload --no-cpu-baseline
# batches alternating over two search streams: the kernels of one batch fill the tails of the other's (higher rates; the
# kernels' own durations then come from a one-stream pass after the timed region)
run 1080p_ra_m3_two_streams --search-streams 2 --no-cpu-baseline
run 4k_ra_m8_two_streams --workload 4k_ra_m8 --search-streams 2 --no-cpu-baseline
run 1080p_ra_m8_two_streams --workload 1080p_ra_m8 --search-streams 2 --no-cpu-baseline
run 720p_ld_m12_one_stream --workload 720p_ld_m12 --search-streams 1 --no-cpu-baseline # (its default is two)
run 1080p_tf_l0_two_streams --workload 1080p_tf_l0 --search-streams 2 --no-cpu-baseline
run 1080p_first_pass_two_streams --workload 1080p_first_pass --search-streams 2 --no-cpu-baseline
# one process, two engines on this GPU behind svtme_group (a rehearsal of the N-GPU group mode: batched submits, uploads
# and peer copies inside the timed region)
SVTME_BENCH_REHEARSAL=1 run 1080p_ra_m8_group2_one_gpu --workload 1080p_ra_m8 --group --gpus 2 --no-cpu-baseline
mkdir -p gpurun_out/profiles && cp profiles/${R}_bench_*.json gpurun_out/profiles/
