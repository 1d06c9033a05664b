#!/bin/bash
# lab: sweeps the tile-order column-group width of the persistent GEMM (VIT_AMD_GEMM_GC) over alternating bench.py runs
for rep in 1 2; do
for gc in 0 6 4 3; do
  VIT_AMD_GEMM_GC=$gc timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > /tmp/gc_$gc.json 2>/dev/null || exit 1
  python - $gc <<'PY'
import json,sys
d=json.loads(open(f'/tmp/gc_{sys.argv[1]}.json').read().strip().splitlines()[-1])
k=d['kernels']
print('GC', sys.argv[1], 'ms/step %.3f' % d['ms_per_step'], ' '.join('%s %.1f' % (n, k[n]['us_per_launch']) for n in ('gemm_qkv','gemm_o_proj_res','gemm_fc1_gelu','gemm_fc2_res')))
PY
done; done
