#!/bin/bash
# usage: tools/pmc_mfma.sh <shape ...>  -> gpurun_out/mfma_busy.json: counter-based MFMA utilisation per kernel
#   util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles).  kernel cycles = SQ_BUSY_CYCLES / 32 (the counter is the
#   sum over the 32 shader engines; on the 1.4 ms fc6 launches it is 97.8 % of GRBM_GUI_ACTIVE / 8, the guide's clock
#   formula, which reads high on dispatches shorter than ~0.3 ms — the 39 us conv5_3 launch would come out at 2.46 GHz).
#   All counters in ONE pass so they describe the same launches.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for shape in "$@"; do
  out=gpurun_out/pmc_mfma_$shape; rm -rf $out
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $out -- python tools/one_kernel.py $shape 12 > /dev/null 2>&1
done
python - "$@" <<'PY'
import csv, glob, collections, json, sys
res = {}
for shape in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/pmc_mfma_{shape}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"] + " grid " + r.get("Grid_Size", "?")
            if "gemm2" in k or "conv3x3" in k or "conv_wgrad_direct" in k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmc_mfma_{shape}/*/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            gs = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            dur[r["Kernel_Name"] + " grid " + str(gs)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, c in acc.items():
        n = len(c["GRBM_GUI_ACTIVE"])
        skip = 2 if n > 4 else 0                     # warm-up launches
        m = lambda name: sum(c[name][skip:]) / max(1, len(c[name][skip:]))
        cyc = m("SQ_BUSY_CYCLES") / 32.0
        d = dur[k][skip:]
        us = sum(d) / max(1, len(d))
        kk = k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0] + " grid " + k.rsplit(" grid ", 1)[1]
        res.setdefault(shape, {})[kk] = {
            "launches": n - skip, "avg_us_profiled": round(us, 2), "kernel_cycles": round(cyc),
            "grbm_gui_active_div8": round(m("GRBM_GUI_ACTIVE") / 8.0),
            "effective_clock_GHz": round(cyc / us / 1e3, 3) if us else None,
            "SQ_VALU_MFMA_BUSY_CYCLES": round(m("SQ_VALU_MFMA_BUSY_CYCLES")),
            "mfma_busy_frac": round(m("SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * cyc), 4),
            "wave_cycles_quad": round(m("SQ_WAVE_CYCLES")), "wait_any_quad": round(m("SQ_WAIT_ANY")),
            "wait_inst_any_quad": round(m("SQ_WAIT_INST_ANY")), "active_inst_any_quad": round(m("SQ_ACTIVE_INST_ANY")),
        }
json.dump(res, open("gpurun_out/mfma_busy.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
