# Knock-out experiment: k_edge_bwd at workload B with and without its weight-gradient partial writes
# (45 KB per workgroup, 250 workgroups = 11.3 MB per launch).  Timing of the standalone launches (bench.py "roofline").
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for ko in 0 1 0 1; do
  EQD_EXP_EDGE_NO_PARTIALS=$ko python bench.py --workload B --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline_all']['k_edge_bwd']['standalone']
print('EQD_EXP_EDGE_NO_PARTIALS=$ko k_edge_bwd standalone avg_launch_us', r['avg_launch_us'])"
done > $O/${1:-r05_e}_partials_knockout.txt 2>&1
cat $O/${1:-r05_e}_partials_knockout.txt
