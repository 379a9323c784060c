# Round-6 measurement driver (runs on the GPU box through gpurun; writes under gpurun_out/).
#   bash profiles/measure_r06.sh TAG step [step ...]
# steps:  tests | testsfast | testsbig | test:<pytest -k expression>
#         bench:<W>[:bf16][:full]     bench.py line of workload W (":full" keeps the CPU baseline and the roofline part)
#         eager:<W>[:bf16]            the same line with --eager
#         drop:<W>[:bf16][:torchpack|:lib]  the line with --dropout 0.25 (training-mode masks drawn every step;
#                                     lib = --dropout-masks library, torchpack = mask packing in torch operators)
#         prof:<W>[:bf16]             rocprofv3 --kernel-trace --stats table + timeline
#         pmc:<W>[:bf16]              MfmaUtil, MFMA ops, FETCH_SIZE, WRITE_SIZE (one counter per rocprofv3 pass)
#         traffic:<W>[:bf16]          FETCH_SIZE, WRITE_SIZE only
#         sq:<W>[:bf16]               SQ issue / stall counters per kernel (two passes of 8)
#         ab:<W>[:bf16]               A/B of the 0 / 1 switch named in $AB_VAR (bench line per run -> <TAG>_ab.txt)
#         collate | dprehearsal | frows
# (bench.py with an explicit --workload runs that workload ALONE: no secondary / inference blocks in a profiled command)
TAG=${1:-r06_a}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
# the identity of the kernel sources these numbers are measured on (bench.py flags counters of another code state as stale)
python equidock_public_amd/build.py --digest > $O/${TAG}_csrc_digest.txt 2>/dev/null
for what in "$@"; do
  IFS=: read -r cmd W opt1 opt2 <<< "$what"
  DT=""; SUF=""; FULL="--no-cpu-baseline --no-roofline"
  for o in "$opt1" "$opt2"; do
    [ "$o" = "bf16" ] && DT="--dtype bf16" && SUF="_bf16"
    [ "$o" = "full" ] && FULL=""
  done
  ST="--steps 20 --warmup 5"; [ "$W" = "B" ] && ST=""; [ "$W" = "A" ] && ST=""
  case $cmd in
  tests)
    EQD_PARITY_DIAGNOSTICS=1 python -m pytest tests -m gpu -q -s --tb=short --durations=15 2>&1 | grep -v Warning > $O/${TAG}_pytest_gpu.log ;;
  testsfast)   # everything except the big-workload oracle comparisons
    python -m pytest tests -m gpu -q -s --tb=short -k "not config_c and not config_e and not config_b and not workload_r" 2>&1 | grep -v Warning > $O/${TAG}_pytest_gpu_fast.log ;;
  testsbig)
    python -m pytest tests -m gpu -q -s --tb=short -k "config_ or workload_r or ragged" 2>&1 | grep -v Warning > $O/${TAG}_pytest_gpu_big.log ;;
  test)
    python -m pytest tests -m gpu -q -s --tb=short -k "$W" 2>&1 | grep -v Warning > $O/${TAG}_pytest_gpu_sel.log ;;
  bench)
    python bench.py --workload $W $DT $ST $FULL > $O/${TAG}_bench_${W}${SUF}.log 2>&1 ;;
  drop)      # training with dropout 0.25 (masks drawn per step inside the graph); drop:<W>:torchpack = packing in torch ops
    PK=""; MS="torch"; [ "$opt1" = "torchpack" -o "$opt2" = "torchpack" ] && PK="torch"
    [ "$opt1" = "lib" -o "$opt2" = "lib" ] && MS="library" && PK="lib"
    EQD_BENCH_DROPOUT_PACK=${PK/lib/} python bench.py --workload $W $DT $ST --dropout 0.25 --dropout-masks $MS > $O/${TAG}_bench_${W}${SUF}_dropout${PK}.log 2>&1 ;;
  eager)
    python bench.py --eager --workload $W $DT $ST --no-cpu-baseline --no-roofline > $O/${TAG}_bench_${W}${SUF}_eager.log 2>&1 ;;
  prof)
    cd /tmp; export TMPDIR=/tmp
    rm -rf /tmp/prof_$W$SUF
    rocprofv3 --kernel-trace --stats -d /tmp/prof_$W$SUF -o h -- python $R/bench.py --workload $W $DT --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/${TAG}_prof_$W$SUF.log 2>&1
    DB=$(find /tmp/prof_$W$SUF -name "*.db" | head -1)
    python $R/profiles/summarize.py $DB $O/${TAG}_kernels_$W$SUF.md "round 6 (${TAG}): workload $W${SUF}" "rocprofv3 --kernel-trace --stats -- python bench.py --workload $W $DT --steps 20 --warmup 5 --no-cpu-baseline --no-roofline" > $O/${TAG}_kernels_$W$SUF.txt 2>&1
    python $R/profiles/timeline.py $DB 130 > $O/${TAG}_timeline_$W$SUF.txt 2>&1
    cd $R ;;
  proftrain)     # the reference's training step (TrainStep: three hipGraphs around the host OT solve): kernel table + timeline
    cd /tmp; export TMPDIR=/tmp
    rm -rf /tmp/proft_$W$SUF
    EQD_BENCH_TRAIN_EAGER_TAIL=0 rocprofv3 --kernel-trace --stats -d /tmp/proft_$W$SUF -o h -- python $R/bench.py --workload $W $DT --train-step --steps 20 --warmup 5 > $O/${TAG}_proftrain_$W$SUF.log 2>&1
    DB=$(find /tmp/proft_$W$SUF -name "*.db" | head -1)
    python $R/profiles/timeline.py $DB 160 > $O/${TAG}_timeline_train_step_$W$SUF.txt 2>&1
    cd $R ;;
  trainstep)     # the training step's bench block alone
    python bench.py --workload $W $DT --train-step --steps 20 --warmup 5 > $O/${TAG}_train_step_$W$SUF.log 2>&1 ;;
  profdrop)      # the kernel table of a training step with dropout 0.25; profdrop:<W>[:bf16][:lib]
    MS="torch"; [ "$opt1" = "lib" -o "$opt2" = "lib" ] && MS="library"
    cd /tmp; export TMPDIR=/tmp
    rm -rf /tmp/profd_$W$SUF
    rocprofv3 --kernel-trace --stats -d /tmp/profd_$W$SUF -o h -- python $R/bench.py --workload $W $DT --steps 20 --warmup 5 --dropout 0.25 --dropout-masks $MS > $O/${TAG}_profdrop_$W$SUF.log 2>&1
    DB=$(find /tmp/profd_$W$SUF -name "*.db" | head -1)
    python $R/profiles/summarize.py $DB $O/${TAG}_kernels_${W}${SUF}_dropout_$MS.md "round 6 (${TAG}): workload $W${SUF}, dropout 0.25, masks: $MS" "rocprofv3 --kernel-trace --stats -- python bench.py --workload $W $DT --steps 20 --warmup 5 --dropout 0.25 --dropout-masks $MS" > /dev/null 2>&1
    cd $R ;;
  pmc|traffic)
    cd /tmp; export TMPDIR=/tmp
    CNTS="MfmaUtil SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 FETCH_SIZE WRITE_SIZE"
    [ "$cmd" = "traffic" ] && CNTS="FETCH_SIZE WRITE_SIZE"
    for CNT in $CNTS; do
      rm -rf /tmp/pmc2; rocprofv3 --kernel-trace --pmc $CNT -d /tmp/pmc2 -o p -- python $R/bench.py --eager --workload $W $DT --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /tmp/pmc2_${W}_$CNT.log 2>&1
      python $R/profiles/pmcstats.py $(find /tmp/pmc2 -name "*.db" | head -1) k_edge k_attn k_rowres k_rowwave k_rowchain k_atb k_linear k_node k_layer k_head k_keypoint > $O/${TAG}_pmc_${W}${SUF}_${CNT}.json 2>&1
    done
    cd $R ;;
  sq)      # issue / stall breakdown of every kernel: two passes of 8 SQ counters (quad-cycles; see MI355X_MICROARCH.md)
    cd /tmp; export TMPDIR=/tmp
    P=0
    for CNT in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU"; do
      P=$((P+1))
      rm -rf /tmp/pmc3; rocprofv3 --kernel-trace --pmc $CNT -d /tmp/pmc3 -o p -- python $R/bench.py --eager --workload $W $DT --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /tmp/pmc3_${W}_$P.log 2>&1
      python $R/profiles/pmcstats.py $(find /tmp/pmc3 -name "*.db" | head -1) k_edge k_attn k_rowres k_rowchain k_atb k_linear k_node > $O/${TAG}_sq_${W}${SUF}_pass$P.json 2>&1
    done
    cd $R ;;
  ab)        # ab:<W>[:bf16]  A/B of the switch named in $AB_VAR (0 / 1), two alternating repeats, one line per run
    for rep in 1 2; do for val in 0 1; do
      env $AB_VAR=$val python bench.py --workload $W $DT --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | \
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W$SUF $AB_VAR=$val', d['value'], 'pairs/s', d['ms_per_step'], 'ms')" >> $O/${TAG}_ab.txt 2>&1
    done; done ;;
  dprehearsal)   # the driver's N > 1 launch line with one rank on RCCL (world of one)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/${TAG}_bench_rccl1_B.log 2>&1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --workload D --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $O/${TAG}_bench_rccl1_D.log 2>&1 ;;
  collate)
    python profiles/bench_collate.py > $O/${TAG}_collate.txt 2>&1 ;;
  frows)
    python profiles/bench_frows.py > $O/${TAG}_frows.txt 2>&1 ;;
  esac
done
ls -la $O | tail -30
