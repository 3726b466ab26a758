#!/bin/bash
# Batch-size comparison (run on the GPU box from the repository root): [EXTRA='--with-upload'] tools/batch_ab.sh <workload> <batch> [<batch> ...]
W=$1; shift
mkdir -p gpurun_out/r3
for b in "$@"; do
  timeout -k 10 200 python3 bench.py --workload $W --steps 40 --warmup 5 --no-cpu-baseline $EXTRA --batch $b > gpurun_out/r3/batch_${W}_$b.json 2> gpurun_out/r3/batch_${W}_$b.err || exit 1
  python3 -c "
import json,sys
d=json.load(open('gpurun_out/r3/batch_${W}_$b.json')); r=d['roofline']
print('$W', $b, round(d['value']), d['verified'], r['kernel'], round(r['frac'],4), round(r['all_kernels']['frac'],3), {k[:-7]: round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()})"
done
