cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
python -m pytest tests -m gpu -q -x -k "bf16 or dropout or head_backward" > gpurun_out/r04_e_pytest_bf16.log 2>&1; tail -3 gpurun_out/r04_e_pytest_bf16.log
for w in "C bf16" "E bf16" "R bf16"; do set -- $w
  python bench.py --workload $1 --dtype $2 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/r04_e_bench_$1_$2.log 2>&1
done
python - <<PY
import json,glob
for f in sorted(glob.glob('gpurun_out/r04_e_bench_*.log')):
    for ln in open(f):
        if ln.startswith('{'):
            d=json.loads(ln); print(f, d['value'], d['ms_per_step']); print({k:round(v,1) for k,v in list(d.get('step_profile',{}).get('us_per_step_by_kernel',{}).items())[:12]})
PY
