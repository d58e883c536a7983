#!/bin/bash
# usage: tools/roi_tasks_trace.sh H W R   -> per-kernel average duration of one roi_tasks_forms.py run (rocprofv3 --kernel-trace)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/roi_tk; rm -rf $out
rocprofv3 --kernel-trace --output-format csv -d $out -- python tools/roi_tasks_forms.py one $1 $2 $3 > $out.log 2>&1
python - <<PY
import csv, glob, collections
d = collections.defaultdict(list)
for f in glob.glob("$out/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0][-60:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if "roi" in k: print(f"{k:62s} n={len(v):4d} avg {sum(v[3:]) / max(1, len(v[3:])):8.1f} us")
PY
rm -rf $out
