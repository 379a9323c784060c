R=$GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline > $R/gpurun_out/q_B.log 2>&1
python bench.py --workload C --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/q_C.log 2>&1
