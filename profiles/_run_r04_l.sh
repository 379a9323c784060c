cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
python -m pytest tests -m gpu -q -k "attention or bf16 or dropout" 2>&1 | tail -4 | cut -c1-200
for ds in 0 1; do for w in C R E; do
  EQD_ATT_DS=$ds python bench.py --workload $w --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/r04_l_bench_${w}_bf16_ds$ds.log 2>&1
done; done
python - <<PY
import json,glob
for f in sorted(glob.glob('gpurun_out/r04_l_bench_*.log')):
    for ln in open(f):
        if ln.startswith('{'):
            d=json.loads(ln); sp=d.get('step_profile',{}).get('us_per_step_by_kernel',{})
            print(f[21:], d['value'], d['ms_per_step'], {k:round(v) for k,v in sp.items() if 'attn_bwd' in k or 'gather' in k})
PY
