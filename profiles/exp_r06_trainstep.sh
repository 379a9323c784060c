# Round-6 experiment: where does the training step's time beyond (model step + exposed OT solve) go?  Variants of TrainStep:
#   graphs (default) | the D -> H copy of the cost matrices outside graph F (a switch that existed for this experiment only) |
#   every launch enqueued from the host (no graphs).  Result (profiles/r06_zx_trainstep_variants.txt): all within the repeat spread;
#   step - model step - exposed solve = 0.15 ms at B, 0.35-0.55 ms at C bf16 = the loss kernels themselves.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', d['value'], 'pairs/s', d['ms_per_step'], 'ms; exposed', d['ot_exposed_ms'], 'solve', d['host_solve_ms'], 'model only', d['model_only_ms_per_step'])"; }
for rep in 1 2; do
  for W in "B f32" "C bf16"; do
    set -- $W
    python bench.py --workload $1 --dtype $2 --train-step --steps 20 --warmup 5 2>/dev/null | tail -1 | line "graphs $1 $2" >> $O/${TAG:-r06_zx}_trainstep_variants.txt
    EQD_TRAINSTEP_COPY_OUTSIDE=1 python bench.py --workload $1 --dtype $2 --train-step --steps 20 --warmup 5 2>/dev/null | tail -1 | line "copy-outside $1 $2" >> $O/${TAG:-r06_zx}_trainstep_variants.txt
    EQD_TRAINSTEP_NO_GRAPHS=1 python bench.py --workload $1 --dtype $2 --train-step --steps 20 --warmup 5 2>/dev/null | tail -1 | line "no-graphs $1 $2" >> $O/${TAG:-r06_zx}_trainstep_variants.txt
  done
done
cat $O/${TAG:-r06_zx}_trainstep_variants.txt
