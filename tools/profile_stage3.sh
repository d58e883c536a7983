cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_s3
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_s3 -- python tools/stage3_step.py bf16 > gpurun_out/s3_prof_run.log 2>&1
f=$(ls gpurun_out/prof_s3/*/*kernel_stats.csv | head -1)
head -40 $f | cut -c1-200 > gpurun_out/s3_kernel_stats_head.csv
cp $f gpurun_out/s3_kernel_stats.csv
rm -rf gpurun_out/prof_s3
cat gpurun_out/s3_kernel_stats_head.csv
