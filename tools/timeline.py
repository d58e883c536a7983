"""Per-step kernel timeline from a rocprofv3 --kernel-trace CSV: start, duration, queue, and the GPU-idle gap before
each kernel (time since the latest end of any earlier kernel).  usage: timeline.py kernel_trace.csv [step_from_end]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "preprocess" in r["Kernel_Name"]]
multi = any("preprocess_multi" in rows[i]["Kernel_Name"] for i in starts)
steps = starts[::2 if multi else 4]                  # one launch per view batch (2 per step) / per view (4, older builds)
lo, hi = steps[-back - 1], steps[-back]
t0 = int(rows[lo]["Start_Timestamp"])
last_end = t0
idle = 0
busy_union = 0
print(f"{'t_us':>9} {'dur_us':>8} {'gap_us':>7} q  kernel")
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - last_end) / 1e3
    if gap > 0:
        idle += gap
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"\(.*", "", n) if "<" not in n else n[:n.index(">") + 1][:80]
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap:7.1f} {r['Queue_Id']}  {n}  grid={r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}")
    last_end = max(last_end, e)
print(f"# step span {(int(rows[hi]['Start_Timestamp']) - t0) / 1e3:.1f} us, idle (no kernel running) {idle:.1f} us")
