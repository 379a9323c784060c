R=$GRAFT_REPO_ROOT
python bench.py --dtype bf16 --no-cpu-baseline > $R/gpurun_out/r01_z_benchB_bf16.log 2>&1
python bench.py --workload C --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r01_z_benchC_bf16.log 2>&1
python bench.py --workload E --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r01_z_benchE_bf16.log 2>&1
