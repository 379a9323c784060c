R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $R/gpurun_out/pytest_gpu.log
python bench.py --no-cpu-baseline > $R/gpurun_out/q_B.log 2>&1
python bench.py --no-cpu-baseline --no-roofline > $R/gpurun_out/q_B2.log 2>&1
python bench.py --workload C --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/q_C.log 2>&1
python bench.py --workload E --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $R/gpurun_out/q_E.log 2>&1
