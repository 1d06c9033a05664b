#!/bin/bash
# lab: SQ counters of the attention backward kernels at ViT-B/16 batch 256 (one pass of 8 SQ counters)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_attn_bwd; rm -rf $O; mkdir -p $O
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/p1 -- python3 tools/time_attn_bwd.py > $O/p1.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_attn_bwd/p1/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "attn_" in k:
        acc[k[28:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print("    %-28s %.4g  (launches %d)" % (c, sum(v) / len(v), len(v)))
PY
find $O -name "*counter_collection.csv" -delete
