#!/bin/bash
# SQ counters of the ROIPool kernels (tools/probes/roi_bench.py), averaged per launch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_roi; rm -rf $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d $out/a -- python tools/probes/roi_bench.py > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $out/b -- python tools/probes/roi_bench.py > /dev/null 2>&1
python - <<PY
import csv, glob, collections
for sub in ("a","b"):
    for f in glob.glob("$out/%s/*/*counter_collection.csv" % sub):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "roi_pool" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"].split("<")[0].split("::")[-1], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print(f"{k[0]:28s} {k[1]:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
