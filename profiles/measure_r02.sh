# Round-2 measurement driver (runs on the GPU box through gpurun; writes under gpurun_out/).
#   bash profiles/measure_r02.sh TAG [tests] [bench] [prof] [pmc]
TAG=${1:-r02_a}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
for what in "$@"; do case $what in
tests)
  python -m pytest tests -m gpu -q -s --tb=line 2>&1 | grep -v Warning > $O/${TAG}_pytest_gpu.log ;;
testsfast)   # everything except the big-workload oracle comparisons
  python -m pytest tests -m gpu -q -s --tb=line -k "not config_c and not config_e and not config_b" 2>&1 | grep -v Warning > $O/${TAG}_pytest_gpu_fast.log ;;
testsbig)
  python -m pytest tests -m gpu -q -s --tb=line -k "config_c or config_e or config_b or ragged" 2>&1 | grep -v Warning > $O/${TAG}_pytest_gpu_big.log ;;
bench)
  python bench.py > $O/${TAG}_bench.log 2>&1
  python bench.py --eager --no-cpu-baseline --no-roofline > $O/${TAG}_bench_eager.log 2>&1
  python bench.py --workload C --steps 20 --warmup 5 > $O/${TAG}_benchC.log 2>&1
  python bench.py --workload C --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $O/${TAG}_benchC_bf16.log 2>&1
  python bench.py --workload E --steps 20 --warmup 5 > $O/${TAG}_benchE.log 2>&1 ;;
benchq)      # the four bench lines without the CPU baseline
  python bench.py --no-cpu-baseline > $O/${TAG}_bench.log 2>&1
  python bench.py --workload C --steps 20 --warmup 5 --no-cpu-baseline > $O/${TAG}_benchC.log 2>&1
  python bench.py --workload C --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $O/${TAG}_benchC_bf16.log 2>&1
  python bench.py --workload E --steps 20 --warmup 5 --no-cpu-baseline > $O/${TAG}_benchE.log 2>&1 ;;
pmctraffic)
  cd /tmp; export TMPDIR=/tmp
  for W in B C E; do for CNT in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc2; rocprofv3 --kernel-trace --pmc $CNT -d /tmp/pmc2 -o p -- python $R/bench.py --eager --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /tmp/pmc2_${W}_$CNT.log 2>&1
    python $R/profiles/pmcstats.py $(find /tmp/pmc2 -name "*.db" | head -1) k_edge k_attn k_rowres k_rowwave k_rowchain k_atb k_linear k_node k_layer > $O/${TAG}_pmc_${W}_${CNT}.json 2>&1
  done; done
  cd $R ;;
benchB)
  python bench.py --no-cpu-baseline > $O/${TAG}_bench.log 2>&1 ;;
prof)
  cd /tmp; export TMPDIR=/tmp
  for W in B C E; do
    rm -rf /tmp/prof_$W
    rocprofv3 --kernel-trace --stats -d /tmp/prof_$W -o h -- python $R/bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/${TAG}_prof_$W.log 2>&1
    DB=$(find /tmp/prof_$W -name "*.db" | head -1)
    python $R/profiles/summarize.py $DB $O/${TAG}_kernels_$W.md "round 2 (${TAG}): workload $W, fp32" "rocprofv3 --kernel-trace --stats -- python bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline --no-roofline" > $O/${TAG}_kernels_$W.txt 2>&1
    python $R/profiles/timeline.py $DB 130 > $O/${TAG}_timeline_$W.txt 2>&1
  done
  cd $R ;;
profbf16)
  cd /tmp; export TMPDIR=/tmp
  rm -rf /tmp/prof_Cb
  rocprofv3 --kernel-trace --stats -d /tmp/prof_Cb -o h -- python $R/bench.py --workload C --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/${TAG}_prof_Cbf16.log 2>&1
  DB=$(find /tmp/prof_Cb -name "*.db" | head -1)
  python $R/profiles/summarize.py $DB $O/${TAG}_kernels_C_bf16.md "round 2 (${TAG}): workload C, bf16" "rocprofv3 --kernel-trace --stats -- python bench.py --workload C --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline" > $O/${TAG}_kernels_C_bf16.txt 2>&1
  cd $R ;;
dprehearsal)   # N = 2 control flow of bench.py on ONE GPU (gloo; real runs: one GPU per rank over RCCL)
  EQD_BENCH_ONE_DEVICE=1 EQD_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 3 > $O/${TAG}_bench_dp2_rehearsal.log 2>&1
  EQD_BENCH_ONE_DEVICE=1 EQD_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --workload D --steps 5 --warmup 2 > $O/${TAG}_bench_dp2_rehearsal_D.log 2>&1 ;;
collate)
  python profiles/bench_collate.py > $O/${TAG}_collate.txt 2>&1 ;;
pmc)
  cd /tmp; export TMPDIR=/tmp
  for W in B C E; do for CNT in MfmaUtil SQ_INSTS_VALU_MFMA_MOPS_F32 FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc2; rocprofv3 --kernel-trace --pmc $CNT -d /tmp/pmc2 -o p -- python $R/bench.py --eager --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /tmp/pmc2_${W}_$CNT.log 2>&1
    python $R/profiles/pmcstats.py $(find /tmp/pmc2 -name "*.db" | head -1) k_edge k_attn k_rowres k_rowwave k_rowchain k_atb k_linear k_node k_layer > $O/${TAG}_pmc_${W}_${CNT}.json 2>&1
  done; done
  cd $R ;;
esac; done
ls -la $O | tail -30
