# final set on HEAD: every GPU test, bench lines, kernel tables and PMC counters for B, C, C bf16, E, R
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
rm -f gpurun_out/parity_report.txt
EQD_PARITY_DIAGNOSTICS=1 python -m pytest tests -m gpu -q --durations=6 2>&1 | grep -v Warning > gpurun_out/${TAG:-r04_y9}_pytest_gpu.log; tail -8 gpurun_out/${TAG:-r04_y9}_pytest_gpu.log | cut -c1-200
cp gpurun_out/parity_report.txt gpurun_out/${TAG:-r04_y9}_parity_report.txt 2>/dev/null
python bench.py > gpurun_out/${TAG:-r04_y9}_bench_default.log 2>&1
for w in "C f32" "C bf16" "E f32" "R f32" "R bf16"; do set -- $w
  python bench.py --workload $1 --dtype $2 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/${TAG:-r04_y9}_bench_$1_$2.log 2>&1
done
python bench.py --dropout 0.25 --no-secondary > gpurun_out/${TAG:-r04_y9}_bench_B_dropout.log 2>&1
bash profiles/measure_r04.sh ${TAG:-r04_y9} prof:B prof:C prof:C:bf16 prof:E prof:R > gpurun_out/${TAG:-r04_y9}_measure_prof.log 2>&1
bash profiles/measure_r04.sh ${TAG:-r04_y9} pmc:B pmc:C pmc:C:bf16 pmc:E pmc:R > gpurun_out/${TAG:-r04_y9}_measure_pmc.log 2>&1
python - <<PY
import json,glob
for f in sorted(glob.glob('gpurun_out/${TAG:-r04_y9}_bench_*.log')):
    for ln in open(f):
        if ln.startswith('{'):
            d=json.loads(ln); print(f[22:], d['value'], d['ms_per_step'], d.get('step_profile',{}).get('library_launches_per_step'), d.get('roofline',{}).get('frac'), {k:(v.get('value'),v.get('error')) for k,v in d.get('secondary',{}).items()}, (d.get('inference') or {}).get('value'))
PY
