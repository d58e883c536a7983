#!/bin/bash
# kernel timeline of one step of tools/shape_step.py <which> (tools/timeline.py) -> gpurun_out/<tag>_<which>_step_timeline.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-r06}; which=${2:-recipe}
rm -rf gpurun_out/prof_tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_tl -- python tools/shape_step.py $which 8 > gpurun_out/tl_shape.log 2>&1
f=$(ls gpurun_out/prof_tl/*/*kernel_trace.csv | head -1)
python tools/timeline.py $f 3 > gpurun_out/${tag}_${which}_step_timeline.txt
rm -rf gpurun_out/prof_tl
tail -2 gpurun_out/${tag}_${which}_step_timeline.txt; cat gpurun_out/tl_shape.log | tail -1
