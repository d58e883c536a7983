#!/bin/bash
# HBM-side bytes of the kernels whose name contains FILTER, for any python script:   tools/pmc_mem_cmd.sh FILTER script.py [args...]
# (rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes, as MI355X_MICROARCH.md prescribes; KiB units; FETCH_SIZE x 2 on gfx950 for
#  16-byte-per-lane streaming reads is NOT applied here: raw counters)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
filter=$1; shift
for c in FETCH_SIZE WRITE_SIZE; do
  out=gpurun_out/pmcm_$c; rm -rf $out
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out -- python "$@" > /dev/null 2>&1
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$out/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "$filter" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("<")[0].split("(")[0]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"$c {k:40s} {sum(v) / len(v) * 1024 / 1e6:10.1f} MB per launch (n={len(v)})")
PY
  rm -rf $out
done
