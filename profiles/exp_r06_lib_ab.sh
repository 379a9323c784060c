# A/B of two builds of the library on one box: the shipped libequidock_hip.so against $1 (a .so under profiles/_exp/), workloads $2..
#   bash profiles/exp_r06_lib_ab.sh profiles/_exp/libequidock_hip_prev.so TAG "C bf16" "E f32" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out; mkdir -p $O
OTHER=$R/$1; TAG=$2; shift; shift
WLS=("$@")      # (set -- below replaces the positional parameters: without the copy the second repeat ran on garbage)
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', d['value'], 'pairs/s', d['ms_per_step'], 'ms')"; }
for rep in 1 2; do
  for W in "${WLS[@]}"; do
    set -- $W
    python bench.py --workload $1 --dtype $2 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | line "shipped $1 $2" >> $O/${TAG}_lib_ab.txt 2>&1
    EQD_EXP_LIBRARY=$OTHER python bench.py --workload $1 --dtype $2 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | line "other   $1 $2" >> $O/${TAG}_lib_ab.txt 2>&1
  done
done
cat $O/${TAG}_lib_ab.txt
