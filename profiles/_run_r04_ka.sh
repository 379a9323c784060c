# keypoint pooling as matrix products: parity on the GPU, then bench lines with the product forms / the first kernels
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "keypoint or head_backward or golden" 2>&1 | grep -v Warning | tail -6 | cut -c1-200
for w in "B f32" "C f32" "C bf16" "E f32"; do set -- $w
  for m in auto 0; do
    if [ $m = auto ]; then unset EQD_KEYPOINT_MM; else export EQD_KEYPOINT_MM=$m; fi
    python bench.py --workload $1 --dtype $2 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/r04_ka_bench_$1_$2_mm$m.log 2>&1
  done
done
unset EQD_KEYPOINT_MM
EQD_KEYPOINT_MM=1 python bench.py --workload B --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/r04_ka_bench_B_f32_mm1.log 2>&1
EQD_KEYPOINT_MM=1 EQD_KEYPOINT_NC=3 python bench.py --workload B --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/r04_ka_bench_B_f32_mm1nc3.log 2>&1
python - <<PY
import json,glob
for f in sorted(glob.glob('gpurun_out/r04_ka_bench_*.log')):
    for ln in open(f):
        if ln.startswith('{'):
            d=json.loads(ln); ra=d.get('roofline_all') or {}
            print(f[25:], d['value'], d['ms_per_step'], {k:(v.get('us_per_step'), v.get('hbm_frac_algorithmic')) for k,v in ra.items() if 'keypoint' in k}, (d.get('inference') or {}).get('value'))
PY
