#!/bin/bash
# Where the data-parallel step's overhead over the single-GPU replay goes (world-1 RCCL group, stage graphs): kernel trace of
# tools/ddp_world1_step.py with and without ddp, one step's timeline each (tools/timeline.py) -> gpurun_out/<tag>_ddp_world1_timeline.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-r06}; out=gpurun_out/${tag}_ddp_world1_timeline.txt; : > $out
for ddp in 1 0; do
  rm -rf gpurun_out/prof_ddp
  USE_GRAPH=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_ddp -- python tools/ddp_world1_step.py 12 $ddp > /dev/null 2> gpurun_out/ddp_tl_err.log
  f=$(ls gpurun_out/prof_ddp/*/*kernel_trace.csv | head -1)
  echo "=== ddp=$ddp (USE_GRAPH=1): $(grep 'ms/step' gpurun_out/ddp_tl_err.log | tail -1)" >> $out
  python tools/timeline.py $f 3 > gpurun_out/ddp_tl_one.txt
  tail -1 gpurun_out/ddp_tl_one.txt >> $out
  # gaps > 3 us and every non-library kernel (RCCL, copies) of the step
  awk '$3 > 3.0 || /nccl|rccl|Kernel_Generic|copyBuffer|fillBuffer/' gpurun_out/ddp_tl_one.txt | head -60 >> $out
done
rm -rf gpurun_out/prof_ddp
cat $out | head -90
