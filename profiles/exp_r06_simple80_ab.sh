# A/B of EQD_LINEAR_SIMPLE80 (0 = the first layer's projection group on k_linear's general body, 1 = on k_linear_simple80 where
# k_linear would run one row tile per workgroup, 2 = at every size), alternating on one box:
#   bash profiles/exp_r06_simple80_ab.sh TAG "B f32" "C bf16" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out; mkdir -p $O
TAG=$1; shift
WLS=("$@")      # (set -- below replaces the positional parameters)
for rep in 1 2; do
  for W in "${WLS[@]}"; do
    set -- $W
    for val in 0 1 2; do
      EQD_LINEAR_SIMPLE80=$val python bench.py --workload $1 --dtype $2 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | \
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2 EQD_LINEAR_SIMPLE80=$val', d['value'], 'pairs/s', d['ms_per_step'], 'ms', 'inference', (d.get('inference') or {}).get('value'))" >> $O/${TAG}_simple80_ab.txt 2>&1
    done
  done
done
cat $O/${TAG}_simple80_ab.txt
