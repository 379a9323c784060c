# end-of-backward as two graph branches (d h[0] -> embedding gradient beside the weight-gradient GEMMs): B / C with and without
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or forked or graph" 2>&1 | grep -v Warning | tail -3 | cut -c1-200
for m in 1 0; do
  EQD_TAIL_FORK=$m python bench.py --workload B --steps 50 --warmup 10 --no-cpu-baseline --no-secondary > gpurun_out/r04_tf_bench_B_fork$m.log 2>&1
  EQD_TAIL_FORK=$m python bench.py --workload C --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/r04_tf_bench_C_bf16_fork$m.log 2>&1
done
EQD_TAIL_FORK=1 python bench.py --workload B --steps 50 --warmup 10 --no-cpu-baseline --no-secondary > gpurun_out/r04_tf_bench_B_fork1b.log 2>&1
EQD_TAIL_FORK=0 python bench.py --workload B --steps 50 --warmup 10 --no-cpu-baseline --no-secondary > gpurun_out/r04_tf_bench_B_fork0b.log 2>&1
python - <<PY
import json,glob
for f in sorted(glob.glob('gpurun_out/r04_tf_bench_*.log')):
    ok=False
    for ln in open(f):
        if ln.startswith('{'):
            d=json.loads(ln); ok=True
            print(f[25:], d['value'], d['ms_per_step'], d.get('step_profile',{}).get('library_launches_per_step'))
    if not ok: print(f, 'NO LINE', open(f).read()[-400:])
PY
