# which kernels run right before / after the memcpy (copyBuffer) and fill launches of a Stage-3 iteration (kernel trace of tools/stage3_step.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_s3n
ITERS=4 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_s3n -- python tools/stage3_step.py bf16 > gpurun_out/s3n_run.log 2>&1
t=$(ls gpurun_out/prof_s3n/*/*kernel_trace.csv | head -1)
python - <<PY
import csv, collections
rows = sorted(csv.DictReader(open("$t")), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n[:70]
last_stem = max(i for i, n in enumerate(names) if "stem_conv7" in n)
# the last iteration = from the third-last stem launch on
stems = [i for i, n in enumerate(names) if "stem_conv7" in n]
lo = stems[-3]
cnt = collections.Counter()
for i in range(lo, len(names)):
    if "copyBuffer" in names[i] or "fillBuffer" in names[i]:
        j = i - 1
        while j > 0 and ("copyBuffer" in names[j] or "fillBuffer" in names[j]): j -= 1
        k = i + 1
        while k < len(names) - 1 and ("copyBuffer" in names[k] or "fillBuffer" in names[k]): k += 1
        cnt[("copy" if "copyBuffer" in names[i] else "fill", short(names[j]), short(names[min(k, len(names) - 1)]))] += 1
print("launches in the last iteration:", len(names) - lo)
for (kind, a, b), n in cnt.most_common(45):
    print(f"{n:3d} {kind}  after {a}  |  before {b}")
PY
rm -rf gpurun_out/prof_s3n
