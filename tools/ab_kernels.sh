for i in 1 2; do for L in libvit_amd.so libvit_amd_base.so; do
VIT_AMD_LIB=$PWD/vit_amd/$L timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('$L', '%.3f ms' % d['ms_per_step'], ' '.join('%s %.1f' % (n[5:], k[n]['us_per_launch']) for n in ('gemm_qkv','gemm_o_proj_res','gemm_fc1_gelu','gemm_fc2_res')))"
done; done
