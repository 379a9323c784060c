# full check on the GPU box: every -m gpu test, then the default bench line (B + the secondary block)    usage: bash profiles/_run_full.sh TAG
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=$1
rm -f gpurun_out/parity_report.txt
EQD_PARITY_DIAGNOSTICS=0 python -m pytest tests -m gpu -q --durations=8 2>&1 | grep -v Warning > gpurun_out/${TAG}_pytest_gpu.log; tail -14 gpurun_out/${TAG}_pytest_gpu.log | cut -c1-200
cp gpurun_out/parity_report.txt gpurun_out/${TAG}_parity_report.txt 2>/dev/null
python bench.py > gpurun_out/${TAG}_bench_default.log 2>&1
python - <<PY
import json
for ln in open('gpurun_out/${TAG}_bench_default.log'):
    if ln.startswith('{'):
        d = json.loads(ln)
        print('B', d['value'], d['ms_per_step'], d['step_profile']['library_launches_per_step'], 'roofline', d['roofline']['frac'])
        for k, v in d.get('secondary', {}).items(): print(k, v.get('value'), v.get('ms_per_step'), v.get('error'))
PY
