# kernel stats of tools/stage3_step.py (bf16): gpurun_out/s3_kernel_stats.csv (+ head), and the slowest single launches of the trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_s3
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_s3 -- python tools/stage3_step.py bf16 > gpurun_out/s3_prof_run.log 2>&1
f=$(ls gpurun_out/prof_s3/*/*kernel_stats.csv | head -1)
t=$(ls gpurun_out/prof_s3/*/*kernel_trace.csv | head -1)
head -40 $f | cut -c1-200 > gpurun_out/s3_kernel_stats_head.csv
cp $f gpurun_out/s3_kernel_stats.csv
cp $t gpurun_out/s3_kernel_trace.csv
python - <<PY
import csv
rows = list(csv.DictReader(open("$t")))
for r in rows: r["d"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
rows.sort(key=lambda r: -r["d"])
print("slowest launches:")
for r in rows[:40]:
    print(f'{r["d"]/1e3:9.1f} us  grid {r.get("Grid_Size_X", "?")}x{r.get("Grid_Size_Y", "?")}x{r.get("Grid_Size_Z", "?")} wg {r.get("Workgroup_Size_X", "?")} lds {r.get("LDS_Block_Size", "?")}  {r["Kernel_Name"][:110]}')
# idle gaps of the last iteration-sized window: which kernel the GPU waited for (the host was syncing or issuing)
rs = sorted(rows, key=lambda r: int(r["Start_Timestamp"]))
tail = [r for r in rs if int(r["Start_Timestamp"]) > int(rs[-1]["End_Timestamp"]) - 110e6]
gaps = []
end = int(tail[0]["End_Timestamp"])
for a, b in zip(tail[:-1], tail[1:]):
    end = max(end, int(a["End_Timestamp"]))
    g = int(b["Start_Timestamp"]) - end
    if g > 0: gaps.append((g, a["Kernel_Name"][:70], b["Kernel_Name"][:70]))
busy = sum(r["d"] for r in tail)
print(f"last 110 ms: {len(tail)} launches, kernel time {busy/1e6:.1f} ms, idle {sum(g for g, _, _ in gaps)/1e6:.1f} ms in {len(gaps)} gaps; gaps > 100 us:")
import collections
hist = collections.Counter()
for g, a, b in gaps:
    hist["<5us" if g < 5e3 else "<20us" if g < 20e3 else "<100us" if g < 100e3 else ">=100us"] += g
print({k: round(v / 1e6, 2) for k, v in hist.items()})
for g, a, b in sorted(gaps, reverse=True)[:25]:
    print(f"  {g/1e3:8.1f} us  after {a}  before {b}")
t0 = min(int(r["Start_Timestamp"]) for r in rows); t1 = max(int(r["End_Timestamp"]) for r in rows)
print("trace span ms", (t1 - t0) / 1e6, "kernel sum ms", sum(r["d"] for r in rows) / 1e6, "launches", len(rows))
PY
rm -rf gpurun_out/prof_s3
cat gpurun_out/s3_kernel_stats_head.csv | cut -c1-160
tail -2 gpurun_out/s3_prof_run.log
