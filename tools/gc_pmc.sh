set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export VIT_AMD_STREAMS=1
for gc in 6 4; do
O=gpurun_out/pmc_gc$gc; rm -rf $O; mkdir -p $O
VIT_AMD_GEMM_GC=$gc timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernels > $O/fetch.log 2>&1
VIT_AMD_GEMM_GC=$gc timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernels > $O/write.log 2>&1
python3 tools/pmc_traffic.py $O/fetch $O/write > $O/traffic.json
find $O -name "*.csv" -delete
python3 -c "
import json
d=json.load(open('$O/traffic.json'))['per_launch_kb']
print('GC $gc', {k:(round(v['FETCH_SIZE']*2/1024,1), round(v['WRITE_SIZE']/1024,1)) for k,v in d.items() if k.startswith('gemm')})
"
done
