#!/bin/bash
# usage: tools/pmc_fetch.sh <shape>...  -> HBM-side FETCH_SIZE / WRITE_SIZE (KB, raw) of the gemm kernels of each shape
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for shape in "$@"; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmcf_${shape}_$c
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmcf_${shape}_$c -- python tools/one_kernel.py $shape 3 > /dev/null 2>&1
  done
  python - <<PY
import csv, glob
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = []
    for f in glob.glob(f"gpurun_out/pmcf_${shape}_{c}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "gemm2_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c: vals.append(float(r["Counter_Value"]))
    print(f"${shape} {c} raw KB avg {sum(vals)/max(len(vals),1):.0f} (n={len(vals)})")
PY
done
