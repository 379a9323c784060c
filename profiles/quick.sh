R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $R/gpurun_out/pytest_gpu.log
python bench.py --no-cpu-baseline --no-roofline > $R/gpurun_out/q_B.log 2>&1
python bench.py --no-cpu-baseline --no-roofline > $R/gpurun_out/q_B2.log 2>&1
python bench.py --workload C --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $R/gpurun_out/q_C.log 2>&1
python bench.py --workload E --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $R/gpurun_out/q_E.log 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof -o h -- python $R/bench.py --eager --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $R/gpurun_out/q_prof.log 2>&1
DB=$(find /tmp/prof -name "*.db" | head -1)
python $R/profiles/summarize.py $DB $R/gpurun_out/q.md "quick" "quick" > $R/gpurun_out/q.txt 2>&1
