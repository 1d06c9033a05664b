# A/B of the few-tile planner (default) against 256-row tiles only (VIT_AMD_GEMM_MI=4), alternating, one box.
set -e
O=gpurun_out/r3d; mkdir -p $O; rm -f $O/*.jsonl
for mi in 4 0; do VIT_AMD_GEMM_MI=$mi timeout -k 10 120 python tools/time_fewtile.py >> $O/fewtile.jsonl; done
for rep in 1 2; do
  for mi in 4 0; do
    for cfg in "b16 256" "l14 16" "l14 128" "h14 64" "h14 8" "b16 32"; do
      set -- $cfg
      VIT_AMD_GEMM_MI=$mi timeout -k 10 200 python bench.py --config $1 --batch $2 --scaling weak --steps 10 --warmup 3 --no-cpu-baseline --no-kernels --no-other-configs | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'mi': '$mi', 'cfg': '$1', 'batch': $2, 'ms': round(d['ms_per_step'],3), 'img_s': round(d['value'],1)}))" >> $O/ab.jsonl
    done
  done
done
cat $O/fewtile.jsonl | cut -c1-400; cat $O/ab.jsonl
