#!/bin/bash
# HBM-side traffic of the fc6 GEMMs (separate --pmc passes, as MI355X_MICROARCH.md prescribes) -> profiles/pmc_traffic.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for shape in fc6_fwd fc6_dgrad fc6_wgrad; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmct_${shape}_$c
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmct_${shape}_$c -- python tools/one_kernel.py $shape 3 > /dev/null 2>&1
  done
done
python - <<PY
import csv, glob, json
out = {"_method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/one_kernel.py <shape> 3 (gemm2 256x256x2 kernel; operands laid out as in the step: padded pitches, dgrad as NT on fc1.weight^T); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of 16-B/lane streaming reads; Infinity-Cache hits are counted); WRITE_SIZE taken as is"}
M, D0, D1 = 8000, 25088, 4096
alg = {"fc6_fwd": 2*(M*D0 + D1*D0 + M*D1), "fc6_dgrad": 2*(M*D1 + D1*D0 + M*D0), "fc6_wgrad": 2*(M*D1 + M*D0) + 4*D1*D0}
for shape in ("fc6_fwd", "fc6_dgrad", "fc6_wgrad"):
    v = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = []
        for f in glob.glob(f"gpurun_out/pmct_{shape}_{c}/*/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if "gemm2_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c: vals.append(float(r["Counter_Value"]))
        v[c] = sum(vals) / 3.0          # per CALL (one_kernel.py runs the shape 3 times; the peeled wgrad is two launches per call)
    out[shape] = {"hbm_bytes_per_launch": round((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024), "fetch_size_kb_raw": round(v["FETCH_SIZE"]),
                  "write_size_kb_raw": round(v["WRITE_SIZE"]), "algorithmic_bytes": alg[shape]}
json.dump(out, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
