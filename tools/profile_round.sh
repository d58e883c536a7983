#!/bin/bash
# The round's profile set, written under gpurun_out/ (copy the summaries into profiles/ afterwards):
#   kernel stats of the benchmark step (rocprofv3 --kernel-trace --stats), PMC traffic (tools/pmc_traffic.sh), MFMA busy (tools/pmc_mfma.sh)
# usage on the GPU box:  bash tools/profile_round.sh [stats|traffic|mfma ...]   (default: stats traffic)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
what="${@:-stats traffic}"
for w in $what; do
  case $w in
    stats)
      rm -rf gpurun_out/prof_round
      rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_round -- python bench.py --no-fp32-line --no-cpu-baseline --no-extra-shapes --steps 30 --warmup 6 > gpurun_out/round_bench_under_rocprof.json 2> gpurun_out/round_prof_err.log
      f=$(ls gpurun_out/prof_round/*/*kernel_stats.csv | head -1)
      n=$(python - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
print(max(int(r["Calls"]) for r in rows if "mine_label" in r["Name"]))
PY
)
      cp $f gpurun_out/round_kernel_stats_raw.csv
      python tools/prof_summary.py $f $n gpurun_out/round_kernel_stats.csv | head -45
      rm -rf gpurun_out/prof_round ;;
    traffic) bash tools/pmc_traffic.sh | tail -40 ;;
    mfma) bash tools/pmc_mfma.sh fc6_fwd fc6_dgrad fc6_wgrad conv5_3 wgrad_grouped | tail -60 ;;
  esac
done
