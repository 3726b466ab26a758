#!/bin/bash
# Diagnostic: a launch-bound workload on one and two search streams, with the fused plan / search / finish kernel and with
# the three kernels (SVTME_ME_UNFUSED). Usage (GPU box, repository root): tools/streams_probe.sh [workload]
mkdir -p gpurun_out/r3
W=${1:-1080p_first_pass_fast}
run() { # tag, streams, env...
  local tag=$1 s=$2; shift 2
  env "$@" timeout -k 10 200 python3 bench.py --workload $W --steps 60 --warmup 10 --no-cpu-baseline --search-streams $s --kernel-steps 0 > gpurun_out/r3/fpf_$tag.json 2> gpurun_out/r3/fpf_$tag.err || exit 1
  python3 -c "
import json,sys
d=json.load(open(sys.argv[1])); print('$W', sys.argv[2], 'streams', d['search_streams'], 'batch', d['pictures_per_submit'], round(d['value']), 'pictures/s', round(d['ms_per_step']*1000/d['config']['pictures_per_step'],2), 'us/pic; kernels of a batch, first start to last end', round(d['picture_kernels_ms_in_timed_region']*1000,2), 'us/pic')" gpurun_out/r3/fpf_$tag.json $tag
}
run fused 1 A=1
run unfused 1 SVTME_ME_UNFUSED=1
run fused 2 A=1
run unfused 2 SVTME_ME_UNFUSED=1
