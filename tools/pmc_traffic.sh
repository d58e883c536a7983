#!/bin/bash
# HBM-side traffic of the fc6 GEMMs (separate --pmc passes, as MI355X_MICROARCH.md prescribes) -> profiles/pmc_traffic.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for shape in fc6_fwd fc6_dgrad fc6_wgrad roi_fwd roi_bwd wgrad_grouped conv5_3; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmct_${shape}_$c
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmct_${shape}_$c -- python tools/one_kernel.py $shape 3 > /dev/null 2>&1
  done
done
python - <<PY
import csv, glob, json
out = {"_method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/one_kernel.py <shape> 3 (gemm2 256x256x2 kernel; operands laid out as in the step: padded pitches, dgrad as NT on fc1.weight^T, wgrad = transpose(dZ) + NN GEMM + peeled tail; roi_* = one ROIPool call of 2 x 2000 ROIs; wgrad_grouped = all conv weight gradients of a backward pass: conv_wgrad_direct_kernel, slabs per the direct kernel's split plan); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of 16-B/lane streaming reads; Infinity-Cache hits are counted); WRITE_SIZE taken as is"}
M, D0, D1 = 8000, 25088, 4096
R2, FM = 4000, 2*63*63*512*2        # one ROIPool call: 2 x 2000 ROIs, bf16 values + u16 argmax; the 2-image bf16 feature map
LW = [(2*128*128, 128, 256), (2*128*128, 256, 256), (2*128*128, 256, 256), (2*64*64, 256, 512), (2*64*64, 512, 512), (2*64*64, 512, 512),
      (2*63*63, 512, 512), (2*63*63, 512, 512), (2*63*63, 512, 512)]
alg = {"fc6_fwd": 2*(M*D0 + D1*D0 + M*D1), "fc6_dgrad": 2*(M*D1 + D1*D0 + M*D0), "fc6_wgrad": 2*(M*D1 + M*D0) + 4*D1*D0,
       "roi_fwd": R2*25088*4 + FM, "roi_bwd": R2*25088*4 + 2*FM,
       "wgrad_grouped": sum(2 * (2*p*(ci + co)) + 2 * 4*co*9*ci for p, ci, co in LW),     # 2 view batches: operands once, one slab each
       "conv5_3": 2 * (2*63*63*512) * 2 + 512*9*512*2}                                    # bf16 map in + out (batch 2), weights once
name = {"roi_fwd": "roi_pool_fwd", "roi_bwd": "roi_pool_bwd", "wgrad_grouped": "conv_wgrad_direct", "conv5_3": "conv3x3_direct"}
for shape in ("fc6_fwd", "fc6_dgrad", "fc6_wgrad", "roi_fwd", "roi_bwd", "wgrad_grouped", "conv5_3"):
    v = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = []
        for f in glob.glob(f"gpurun_out/pmct_{shape}_{c}/*/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if (name.get(shape, "gemm2_kernel") in r["Kernel_Name"] or (shape == "fc6_wgrad" and "transpose_2d" in r["Kernel_Name"])) \
                        and r["Counter_Name"] == c: vals.append(float(r["Counter_Value"]))
        v[c] = sum(vals) / (4.0 if shape == "roi_fwd" else 3.0)   # per CALL (one_kernel.py runs the shape 3 times [+1 set-up call for roi]; the peeled wgrad is transpose + two launches per call)
    out[shape] = {"hbm_bytes_per_launch": round((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024), "fetch_size_kb_raw": round(v["FETCH_SIZE"]),
                  "write_size_kb_raw": round(v["WRITE_SIZE"]), "algorithmic_bytes": alg[shape]}
json.dump(out, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
