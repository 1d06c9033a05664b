#!/bin/bash
# lab: SQ counters of an attention kernel (default: dh64 at ViT-B/16 batch 256; `PMC_KERNEL=attention_ring tools/pmc_attn.sh 0` = head_dim 80 at ViT-H/14 batch 32) (three passes of 8 SQ counters, each its own run)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_attn; rm -rf $O; mkdir -p $O
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d $O/p1 -- python3 tools/time_attn.py ${1:-2} > $O/p1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/p2 -- python3 tools/time_attn.py ${1:-2} > $O/p2.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_WAVES --output-format csv -d $O/p3 -- python3 tools/time_attn.py ${1:-2} > $O/p3.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections, os
for p in ("p1", "p2", "p3"):
    f = glob.glob(f"gpurun_out/pmc_attn/{p}/*/*counter_collection.csv")[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if os.environ.get("PMC_KERNEL", "attention_dh64") in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(p, k, "per launch: %.4g" % (sum(v) / len(v)), "launches", len(v))
PY
find $O -name "*counter_collection.csv" -delete
