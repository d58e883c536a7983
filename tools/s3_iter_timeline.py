"""One steady-state Stage-3 iteration out of a rocprofv3 kernel trace (tools/profile_stage3.sh keeps gpurun_out/s3_kernel_trace.csv):
iterations are cut at the teacher EMA's first launch; prints the iteration's span, kernel-time sum, GPU-idle time and its largest
gaps (what ran before / after: a host sync or a slow host stretch), and the per-kernel totals of that iteration.
usage: s3_iter_timeline.py trace.csv [iteration_from_end=2] [--list]"""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 2
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return (re.sub(r"\(.*", "", n) if "<" not in n else n[:n.index(">") + 1])[:90]
marks = [i for i, r in enumerate(rows) if "ema_multi" in r["Kernel_Name"] and (i == 0 or "ema_multi" not in rows[i - 1]["Kernel_Name"])]
lo, hi = marks[-back - 1], marks[-back]
it = rows[lo:hi]
t0 = it[0]["s"]; span = rows[hi]["s"] - t0
ksum = sum(r["e"] - r["s"] for r in it)
end = it[0]["e"]; idle = 0; gaps = []
for a, b in zip(it[:-1], it[1:]):
    end = max(end, a["e"])
    g = b["s"] - end
    if g > 0:
        idle += g; gaps.append((g, short(a["Kernel_Name"]), short(b["Kernel_Name"]), (b["s"] - t0) / 1e3))
print(f"iteration: span {span/1e3:.1f} us, {len(it)} launches, kernel sum {ksum/1e3:.1f} us, idle {idle/1e3:.1f} us in {len(gaps)} gaps")
h = collections.Counter()
for g, *_ in gaps:
    h["<2us" if g < 2e3 else "<5us" if g < 5e3 else "<20us" if g < 20e3 else "<100us" if g < 100e3 else ">=100us"] += g
print("idle by gap size (us):", {k: round(v / 1e3, 1) for k, v in h.items()})
for g, a, b, t in sorted(gaps, reverse=True)[:30]:
    print(f"  {g/1e3:7.1f} us at t={t:8.1f}  after {a[:50]:50s} before {b[:50]}")
tot = collections.defaultdict(lambda: [0, 0])
for r in it:
    k = short(r["Kernel_Name"]); tot[k][0] += 1; tot[k][1] += r["e"] - r["s"]
print("per kernel (calls, us):")
for k, (c, d) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:70]:
    print(f"  {c:4d} {d/1e3:8.1f}  {k}")
if "--list" in sys.argv:
    for r in it:
        print(f"{(r['s']-t0)/1e3:9.1f} {(r['e']-r['s'])/1e3:7.1f} q{r['Queue_Id']} {short(r['Kernel_Name'])} grid={r['Grid_Size_X']}")
