# final round-4 set: every GPU test (+ parity report with diagnostics), the default bench line, single bench lines of the other configs
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
rm -f gpurun_out/parity_report.txt
EQD_PARITY_DIAGNOSTICS=1 python -m pytest tests -m gpu -q --durations=6 2>&1 | grep -v Warning > gpurun_out/r04_z_pytest_gpu.log; tail -10 gpurun_out/r04_z_pytest_gpu.log | cut -c1-200
cp gpurun_out/parity_report.txt gpurun_out/r04_z_parity_report.txt 2>/dev/null
python bench.py > gpurun_out/r04_z_bench_default.log 2>&1
for w in "C f32" "C bf16" "E f32" "E bf16" "R f32" "R bf16" "A f32"; do set -- $w
  python bench.py --workload $1 --dtype $2 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/r04_z_bench_$1_$2.log 2>&1
done
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04_z_bench_rccl1_default.log 2>&1
python - <<PY
import json,glob
for f in sorted(glob.glob('gpurun_out/r04_z_bench_*.log')):
    for ln in open(f):
        if ln.startswith('{'):
            d=json.loads(ln); print(f[21:], d['value'], d['ms_per_step'], d.get('step_profile',{}).get('library_launches_per_step'), d.get('roofline',{}).get('frac'), {k:(v.get('value'),v.get('error')) for k,v in d.get('secondary',{}).items()})
PY
