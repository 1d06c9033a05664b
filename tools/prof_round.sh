set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_${PROF_TAG:-r03c}
rm -rf $O && mkdir -p $O
export VIT_AMD_STREAMS=1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s1 -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs > $O/s1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernels --no-other-configs > $O/fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernels --no-other-configs > $O/write.log 2>&1
unset VIT_AMD_STREAMS
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s2 -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernels --no-other-configs > $O/s2.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/h14 -- python3 bench.py --config h14 --steps 3 --warmup 1 --no-cpu-baseline --no-kernels --no-other-configs > $O/h14.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -- python3 bench.py --workload train --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs > $O/train.log 2>&1
python3 tools/pmc_traffic.py $O/fetch $O/write > $O/traffic.json
for d in s1 s2 h14 train; do python3 tools/prof_filter.py $O/$d/*/*_kernel_stats.csv > $O/${d}_kernel_stats.csv; done
find $O -name "*kernel_stats.csv" | head; find $O -name "*.csv" -size +20M -delete; find $O -name "*_kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; du -sh $O
tail -2 $O/s1.log | cut -c1-400
