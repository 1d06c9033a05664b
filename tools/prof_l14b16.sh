# Per-kernel durations of CLIP-L/14-336 at 16 images (the per-GPU shard of BASELINE.json configs[2] on 8 GPUs) and at 128.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r03_l14; rm -rf $O; mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b16 -- python3 bench.py --config l14 --batch 16 --steps 5 --warmup 2 --no-cpu-baseline --no-kernels > $O/b16.log 2>&1
VIT_AMD_STREAMS=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b128 -- python3 bench.py --config l14 --batch 128 --steps 3 --warmup 1 --no-cpu-baseline --no-kernels > $O/b128.log 2>&1
python3 tools/prof_filter.py $O/b16/*/*_kernel_stats.csv > $O/l14_b16_kernel_stats.csv
python3 tools/prof_filter.py $O/b128/*/*_kernel_stats.csv > $O/l14_b128_single_stream_kernel_stats.csv
find $O/b16 $O/b128 -name "*.csv" -delete
grep -h '"metric"' $O/b16.log $O/b128.log | cut -c1-200
