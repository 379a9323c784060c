R=$GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline --no-roofline > $R/gpurun_out/q_B.log 2>&1
python bench.py --eager --no-cpu-baseline --no-roofline > $R/gpurun_out/q_Be.log 2>&1
python bench.py --workload C --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $R/gpurun_out/q_C.log 2>&1
python bench.py --workload E --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $R/gpurun_out/q_E.log 2>&1
EQD_BENCH_ONE_DEVICE=1 EQD_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $R/gpurun_out/q_N2.log 2>&1; echo "rc=$?" >> $R/gpurun_out/q_N2.log
