R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $R/gpurun_out/pytest_gpu.log
python profiles/bench_losses.py > $R/gpurun_out/bench_losses.log 2>&1
