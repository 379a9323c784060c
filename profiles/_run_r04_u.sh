cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
python -m pytest tests -m gpu -q -k "edge or attention or linear or golden or dropout or head or kabsch or keypoint" 2>&1 | tail -3 | cut -c1-200
python bench.py --no-cpu-baseline --no-secondary > gpurun_out/r04_u_bench_B.log 2>&1
for w in "C f32" "C bf16" "E f32"; do set -- $w
  python bench.py --workload $1 --dtype $2 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/r04_u_bench_$1_$2.log 2>&1
done
python - <<PY
import json,glob
for f in sorted(glob.glob('gpurun_out/r04_u_bench_*.log')):
    for ln in open(f):
        if ln.startswith('{'):
            d=json.loads(ln); sp=d.get('step_profile',{}).get('us_per_step_by_kernel',{})
            print(f[21:], d['value'], d['ms_per_step'], {k:round(v) for k,v in list(sp.items())[:7]}, d.get('roofline',{}).get('avg_launch_us'), d.get('roofline',{}).get('frac'))
PY
