cd $GRAFT_REPO_ROOT
for t in 0 8 5 16 12; do
  if [ $t = 0 ]; then unset EQD_ROWRES_TPS; else export EQD_ROWRES_TPS=$t; fi
  python bench.py --workload C --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('EQD_ROWRES_TPS=$t', d['value'], d['ms_per_step'])"
done > gpurun_out/r05_w_tps.txt 2>&1
cat gpurun_out/r05_w_tps.txt
