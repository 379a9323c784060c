# quick A/B line: bench B with the step profile, no CPU baseline, no secondary   (usage: bash profiles/_run_quick.sh TAG [extra bench args])
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
TAG=$1; shift
python bench.py --no-cpu-baseline --no-secondary "$@" > gpurun_out/${TAG}_bench_B.log 2>&1
python - <<PY
import json
for ln in open('gpurun_out/${TAG}_bench_B.log'):
    if ln.startswith('{'):
        d = json.loads(ln)
        print(d['value'], d['ms_per_step'], d['step_profile']['library_launches_per_step'])
        print(d['step_profile']['us_per_step_by_kernel'])
        print({k: v['avg_launch_us'] for k, v in d['roofline_all'].items()})
        print('roofline', d['roofline']['kernel'], d['roofline']['avg_launch_us'], d['roofline']['frac'])
PY
tail -3 gpurun_out/${TAG}_bench_B.log | grep -v "^{" | cut -c1-300
