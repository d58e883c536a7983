#!/bin/bash
# usage: tools/pmc_kernel.sh <shape> <tagdir>   -> prints averaged SQ counters of the gemm kernels for that shape
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
shape=$1; out=gpurun_out/pmc_$2; rm -rf $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $out/a -- python tools/one_kernel.py $shape > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $out/b -- python tools/one_kernel.py $shape > /dev/null 2>&1
python - <<PY
import csv, glob, collections
for sub in ("a","b"):
    for f in glob.glob("$out/%s/*/*counter_collection.csv" % sub):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "gemm" in r["Kernel_Name"] or "conv3x3" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print(f"{k:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
