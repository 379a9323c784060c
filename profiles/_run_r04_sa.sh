# split body (eqd_linsplit_inl.h): linear parity on the GPU, then B with and without it
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "linear or row or golden or ops" 2>&1 | grep -v Warning | tail -6 | cut -c1-200
python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r04_sa_bench_B_split1.log 2>&1
EQD_ROWSPLIT=0 python bench.py --no-secondary --no-cpu-baseline > gpurun_out/r04_sa_bench_B_split0.log 2>&1
python - <<PY
import json,glob
for f in sorted(glob.glob('gpurun_out/r04_sa_bench_*.log')):
    for ln in open(f):
        if ln.startswith('{'):
            d=json.loads(ln); print(f, d['value'], d['ms_per_step'], d.get('step_profile',{}).get('library_launches_per_step'))
            for k in d.get('roofline_all',[]):
                if k['kernel'] in ('k_rowchain','k_linear'): print('   ', k['kernel'], k.get('us_per_step'), k.get('launches_per_step'))
PY
