#!/bin/bash
# lab: SQ counters of the forward's kernels (single stream, full-batch launches), two passes of 8 SQ counters
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_gemm; rm -rf $O; mkdir -p $O
export VIT_AMD_STREAMS=1
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_MFMA --output-format csv -d $O/p1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernels > $O/p1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $O/p2 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernels > $O/p2.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections
for p in ("p1", "p2"):
    f = glob.glob(f"gpurun_out/pmc_gemm/{p}/*/*counter_collection.csv")[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm8p" in k or "attention" in k:
            acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(p, k)
        for c, v in sorted(d.items()):
            print("    %-28s %.4g  (launches %d)" % (c, sum(v) / len(v), len(v)))
PY
find $O -name "*counter_collection.csv" -delete
