# key / query maps as matrix products (k_head_u_mm, k_head_u_bwd_mm): parity + B / C lines against the first kernels of this stage
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "keypoint or head_backward or golden" 2>&1 | grep -v Warning | tail -4 | cut -c1-200
for w in "B f32" "C bf16" "C f32"; do set -- $w
  python bench.py --workload $1 --dtype $2 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/r04_kd_bench_$1_$2.log 2>&1
done
python - <<PY
import json,glob
for f in sorted(glob.glob('gpurun_out/r04_kd_bench_*.log')):
    for ln in open(f):
        if ln.startswith('{'):
            d=json.loads(ln); ra=d.get('roofline_all') or {}; sp=d.get('step_profile',{}).get('us_per_step_by_kernel',{})
            print(f[25:], d['value'], d['ms_per_step'], d.get('step_profile',{}).get('library_launches_per_step'), {k:v for k,v in sp.items() if 'keypoint' in k or 'head' in k})
PY
