#!/bin/bash
# SQ counters of the kernels whose name contains FILTER, for any python script:   tools/pmc_cmd.sh FILTER TAG script.py [args...]
# (two rocprofv3 --pmc passes, 8 SQ counters each; averages per dispatch; the program after `--` is python itself: no exec hop)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
filter=$1; tag=$2; shift 2
out=gpurun_out/pmc_$tag; rm -rf $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $out/a -- python "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $out/b -- python "$@" > /dev/null 2>&1
python - <<PY
import csv, glob, collections
for sub in ("a","b"):
    for f in glob.glob("$out/%s/*/*counter_collection.csv" % sub):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "$filter" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print(f"{k:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
rm -rf $out
