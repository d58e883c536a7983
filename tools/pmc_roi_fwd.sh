#!/bin/bash
# SQ counters of the ROIPool forward kernels (tools/roi_fwd_forms.py one H W R), averaged per launch.  usage: pmc_roi_fwd.sh H W R [sparse 0|1]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
H=${1:-63}; W=${2:-63}; R=${3:-4000}; export SW_ROI_FWD_SPARSE=${4:-1}
out=gpurun_out/pmc_roi_fwd_$H_$W_$SW_ROI_FWD_SPARSE; rm -rf $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d $out/a -- python tools/roi_fwd_forms.py one $H $W $R > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $out/b -- python tools/roi_fwd_forms.py one $H $W $R > /dev/null 2>&1
python - <<PY
import csv, glob, collections
print("map $H x $W, $R ROIs, SW_ROI_FWD_SPARSE=$SW_ROI_FWD_SPARSE")
for sub in ("a","b"):
    for f in glob.glob("$out/%s/*/*counter_collection.csv" % sub):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "roi_pool" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"].split("<")[0].split("::")[-1], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in sorted(acc.items()):
            print(f"{k[0]:28s} {k[1]:28s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
