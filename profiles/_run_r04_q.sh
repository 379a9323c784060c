cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
python -m pytest tests -m gpu -q -k "bf16 or ds_handoff or gather or attention" 2>&1 | tail -4 | cut -c1-200
for w in "C bf16" "R bf16" "E bf16"; do set -- $w
  python bench.py --workload $1 --dtype $2 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/r04_q_bench_$1_$2.log 2>&1
done
python bench.py --dtype bf16 --no-cpu-baseline --no-secondary > gpurun_out/r04_q_bench_B_bf16.log 2>&1
python - <<PY
import json,glob
for f in sorted(glob.glob('gpurun_out/r04_q_bench_*.log')):
    for ln in open(f):
        if ln.startswith('{'):
            d=json.loads(ln); sp=d.get('step_profile',{}).get('us_per_step_by_kernel',{})
            print(f[21:], d['value'], d['ms_per_step'], {k:round(v) for k,v in sp.items() if 'attn' in k})
PY
