set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r02c; rm -rf $O; mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/h14 -- python3 bench.py --config h14 --steps 3 --warmup 1 --no-cpu-baseline --no-kernels > $O/h14.log 2>&1
VIT_AMD_STREAMS=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/h14s1 -- python3 bench.py --config h14 --steps 3 --warmup 1 --no-cpu-baseline --no-kernels > $O/h14s1.log 2>&1
python3 tools/prof_filter.py $O/h14/*/*_kernel_stats.csv > $O/h14_kernel_stats.csv
python3 tools/prof_filter.py $O/h14s1/*/*_kernel_stats.csv > $O/h14_single_stream_kernel_stats.csv
find $O/h14 $O/h14s1 -name "*.csv" -delete
grep -h '"metric"' $O/h14.log $O/h14s1.log | cut -c1-200
