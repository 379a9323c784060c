# Round-6 knock-out builds ("price with knock-out builds, then build what pays").  Timing only -
# the knock-out libraries compute WRONG weight gradients.  Two steps:
#   bash profiles/exp_r06_knockouts.sh build      here (hipcc cross-compiles): profiles/_exp/libequidock_hip_noslab.so =
#                                                 the shipped objects with eqd_edge_kernels.hip rebuilt under -DEQD_EXP_NO_SLABS
#                                                 (k_edge_bwd without its three weight-gradient slab GEMMs, their LDS stores, barriers)
#   bash profiles/exp_r06_knockouts.sh run TAG    on the GPU box: bench lines of C bf16, C fp32, B with the shipped library and
#                                                 the knock-out one, alternating -> gpurun_out/<TAG>_knockouts.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function"
if [ "$1" = "build" ]; then
  python -m equidock_public_amd.build > /dev/null
  mkdir -p profiles/_exp
  /opt/rocm/bin/hipcc $FLAGS -DEQD_EXP_NO_SLABS -c equidock_public_amd/csrc/eqd_edge_kernels.hip -o profiles/_exp/eqd_edge_kernels_noslab.o
  OBJS=$(ls equidock_public_amd/csrc/build/*.o | grep -v eqd_edge_kernels.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o profiles/_exp/libequidock_hip_noslab.so $OBJS profiles/_exp/eqd_edge_kernels_noslab.o
  ls -la profiles/_exp/libequidock_hip_noslab.so
  exit 0
fi
TAG=${2:-r06_c}; O=$R/gpurun_out; mkdir -p $O
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$1', d['value'], 'pairs/s', d['ms_per_step'], 'ms')"; }
for rep in 1 2; do
  for lib in shipped noslab; do
    L=""; [ "$lib" = "noslab" ] && L="$R/profiles/_exp/libequidock_hip_noslab.so"
    for W in "C bf16" "C f32" "B f32"; do
      set -- $W
      EQD_EXP_LIBRARY=$L python bench.py --workload $1 --dtype $2 --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | line "$lib $1 $2" >> $O/${TAG}_knockouts.txt 2>&1
    done
  done
done
cat $O/${TAG}_knockouts.txt
