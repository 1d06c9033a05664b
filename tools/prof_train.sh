# the training-step profiles of a round: default (two streams / side stream) and with every overlap off
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_train_final; rm -rf $O; mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 bench.py --workload train --steps 4 --warmup 2 --no-cpu-baseline > $O/train.log 2>&1
python3 tools/prof_filter.py $O/t/*/*_kernel_stats.csv > $O/train_kernel_stats.csv
export VIT_AMD_STREAMS=1 VIT_AMD_BWD_SIDE_STREAM=0
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s -- python3 bench.py --workload train --steps 4 --warmup 2 --no-cpu-baseline > $O/train_serial.log 2>&1
python3 tools/prof_filter.py $O/s/*/*_kernel_stats.csv > $O/train_serial_kernel_stats.csv
find $O/t $O/s -name "*.csv" -delete
grep -h '"metric"' $O/train.log $O/train_serial.log | cut -c1-220
