cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
python -m pytest tests -m gpu -q -k "edge or bf16 or dropout" 2>&1 | tail -4 | cut -c1-200
for w in "C bf16" "R bf16"; do set -- $w
  python bench.py --workload $1 --dtype $2 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/r04_r_bench_$1_$2.log 2>&1
done
python - <<PY
import json,glob
for f in sorted(glob.glob('gpurun_out/r04_r_bench_*.log')):
    for ln in open(f):
        if ln.startswith('{'):
            d=json.loads(ln); sp=d.get('step_profile',{}).get('us_per_step_by_kernel',{})
            print(f[21:], d['value'], d['ms_per_step'], {k:round(v) for k,v in sp.items() if 'edge' in k}, d['roofline']['kernel'], d['roofline']['avg_launch_us'])
PY
