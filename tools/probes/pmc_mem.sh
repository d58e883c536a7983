#!/bin/bash
# (TA_* counters are left out on purpose: a pass with them hung rocprofv3 on this pool)
# usage: tools/pmc_mem.sh <shape>   -> memory-system counters (TCC/TCP/TA) of the gemm kernels of that shape, averaged per launch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
shape=$1; out=gpurun_out/pmcm_$shape; rm -rf $out
i=0
for set in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum" "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum TCC_EA0_RDREQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$i -- python tools/one_kernel.py $shape 3 > $out.$i.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$out/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Kernel_Name"] or "conv3x3" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k:36s} {sum(v)/len(v):18.0f}  (n={len(v)})")
PY
