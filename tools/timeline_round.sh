#!/bin/bash
# kernel timeline of one replayed step (tools/timeline.py) -> gpurun_out/step_timeline.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_tl -- python bench.py --no-fp32-line --no-cpu-baseline --no-extra-shapes --steps 12 --warmup 4 > gpurun_out/tl_bench.json 2> gpurun_out/tl_err.log
f=$(ls gpurun_out/prof_tl/*/*kernel_trace.csv | head -1)
python tools/timeline.py $f ${1:-14} > gpurun_out/step_timeline.txt
rm -rf gpurun_out/prof_tl
tail -3 gpurun_out/step_timeline.txt
