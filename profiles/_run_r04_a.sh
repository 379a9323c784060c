cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
( time python bench.py ) > gpurun_out/r04_a_bench_default.log 2>&1
bash profiles/measure_r03.sh r04_a prof:B prof:C:bf16 > gpurun_out/r04_a_measure.log 2>&1
tail -3 gpurun_out/r04_a_bench_default.log | cut -c1-600
