cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
bash profiles/_run_quick.sh r04_d
for occ in 1 2; do
  EQD_ROWCHAIN_OCC=$occ python bench.py --workload C --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/r04_d_bench_Cbf16_occ$occ.log 2>&1
  EQD_ROWCHAIN_OCC=$occ python bench.py --workload C --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline > gpurun_out/r04_d_bench_C_occ$occ.log 2>&1
done
python - <<PY
import json,glob
for f in sorted(glob.glob('gpurun_out/r04_d_bench_C*.log')):
    for ln in open(f):
        if ln.startswith('{'):
            d=json.loads(ln); print(f, d['value'], d['ms_per_step'], d.get('step_profile',{}).get('us_per_step_by_kernel',{}).get('k_rowchain'))
PY
