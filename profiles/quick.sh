R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $R/gpurun_out/pytest_gpu.log
for F in 0 1 2 4 8; do
EQD_WGRAD_FLUSH=$F python bench.py --no-cpu-baseline --no-roofline > $R/gpurun_out/q_B_f$F.log 2>&1
done
EQD_WGRAD_FLUSH=2 python bench.py --graph --no-cpu-baseline --no-roofline > $R/gpurun_out/q_B_graph.log 2>&1
for F in 0 2 8; do
EQD_WGRAD_FLUSH=$F python bench.py --workload C --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $R/gpurun_out/q_C_f$F.log 2>&1
done
