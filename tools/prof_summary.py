"""Condense a rocprofv3 --kernel-trace --stats run (…_kernel_stats.csv) into a per-step table for profiles/."""
import csv
import sys

stats, steps, out = sys.argv[1], float(sys.argv[2]), sys.argv[3]
rows = list(csv.DictReader(open(stats)))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
with open(out, "w") as f:
    f.write(f"# source: {stats}  (rocprofv3 --kernel-trace --stats; {steps:g} bench steps incl. warm-up)\n")
    f.write(f"# total GPU time per step: {tot / steps / 1e6:.3f} ms\n")
    f.write("# rows with < 1 call per step (at::native fills / random init, conv_weight_prep, ...) run once, before the first step:\n")
    f.write("# model construction, input synthesis, the first staging of the compute-dtype weights\n")
    f.write("kernel,calls_per_step,ms_per_step,avg_us,percent\n")
    for r in rows:
        n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        n = n.split("(")[0] if "<" not in n else n[: n.index(">") + 1] if n.index(">") < 90 else n[:90]
        f.write(f"\"{n}\",{int(r['Calls']) / steps:.1f},{int(r['TotalDurationNs']) / steps / 1e6:.4f},"
                f"{float(r['AverageNs']) / 1e3:.1f},{r['Percentage']}\n")
print(open(out).read()[:2500])
