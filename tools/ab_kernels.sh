#!/bin/bash
# lab: two builds of the library, per-kernel live timings of bench.py side by side, alternating on one box
# usage: tools/ab_kernels.sh libA.so libB.so [rounds]
A=${1:-vit_amd/libvit_amd.so}; B=${2:-vit_amd/libvit_base.so}; R=${3:-2}
for i in $(seq $R); do for L in $A $B; do
VIT_AMD_LIB=$PWD/$L timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('$L', '%.3f ms' % d['ms_per_step'], ' '.join('%s %.1f' % (n[5:], k[n]['us_per_launch']) for n in ('gemm_qkv','gemm_o_proj_res','gemm_fc1_gelu','gemm_fc2_res')))" || exit 1
done; done
