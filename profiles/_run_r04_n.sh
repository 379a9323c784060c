cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
python -m pytest tests -m gpu -q -k "workspace or gather or attention or golden or toggles or smoke" 2>&1 | tail -4 | cut -c1-200
python bench.py --no-cpu-baseline > gpurun_out/r04_n_bench_default.log 2>&1
python - <<PY
import json
for ln in open('gpurun_out/r04_n_bench_default.log'):
    if ln.startswith('{'):
        d = json.loads(ln)
        print('B', d['value'], d['ms_per_step'], d['step_profile']['library_launches_per_step'], d['step_profile']['share_of_kernel_time_in_roofline_all'])
        print({k: round(v, 1) for k, v in d['step_profile']['us_per_step_by_kernel'].items()})
        for k, v in d.get('secondary', {}).items(): print(k, v.get('value'), v.get('ms_per_step'), v.get('error'))
        print(d.get('inference'))
PY
tail -3 gpurun_out/r04_n_bench_default.log | grep -v "^{" | cut -c1-300
