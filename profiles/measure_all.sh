TAG=${1:-r01_x}
R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $R/gpurun_out/pytest_gpu.log
python bench.py > $R/gpurun_out/${TAG}_bench.log 2>&1
python bench.py --eager --no-cpu-baseline --no-roofline > $R/gpurun_out/${TAG}_bench_eager.log 2>&1
python bench.py --workload C --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/${TAG}_benchC.log 2>&1
python bench.py --workload E --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/${TAG}_benchE.log 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof -o h -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $R/gpurun_out/bench_prof13.log 2>&1
DB=$(find /tmp/prof -name "*.db" | head -1)
python $R/profiles/summarize.py $DB $R/gpurun_out/${TAG}.md "round 1 (${TAG}): end-of-round state, workload B (8 pairs x 200/200, 8 layers, fp32)" "rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline" > $R/gpurun_out/${TAG}.txt 2>&1
python $R/profiles/timeline.py $DB > $R/gpurun_out/${TAG}_tl.txt 2>&1
for W in B C; do for CNT in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc; rocprofv3 --kernel-trace --pmc $CNT -d /tmp/pmc -o p -- python $R/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /tmp/pmc_$W_$CNT.log 2>&1
  python $R/profiles/pmcstats.py $(find /tmp/pmc -name "*.db" | head -1) k_edge > $R/gpurun_out/pmc_${W}_${CNT}.json 2>&1
done; done
