#!/bin/bash
# rocprofv3 kernel stats of the training step at the recipe's mean scale and at the COCO shape (tools/shape_step.py), condensed per step
# into gpurun_out/<tag>_{recipe,coco}_kernel_stats.csv   (copy into profiles/ afterwards).   usage: bash tools/profile_shapes.sh <tag> [recipe coco headline]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-r06}; shift
for which in ${@:-recipe coco}; do
  rm -rf gpurun_out/prof_shape
  n=10
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_shape -- python tools/shape_step.py $which $n > gpurun_out/${tag}_${which}_step.txt 2> gpurun_out/${tag}_${which}_err.log
  f=$(ls gpurun_out/prof_shape/*/*kernel_stats.csv | head -1)
  cp $f gpurun_out/${tag}_${which}_kernel_stats_raw.csv
  python tools/prof_summary.py $f $((n + 6)) gpurun_out/${tag}_${which}_kernel_stats.csv | head -40
  cat gpurun_out/${tag}_${which}_step.txt
  rm -rf gpurun_out/prof_shape
done
