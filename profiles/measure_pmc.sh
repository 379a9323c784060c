# MFMA utilisation, MFMA f32 op count and LDS bank conflicts of the main kernels (separate --pmc passes, --kernel-trace only)
TAG=${1:-r01_z}
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for W in B C; do for CNT in MfmaUtil SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do
  rm -rf /tmp/pmc2; rocprofv3 --kernel-trace --pmc $CNT -d /tmp/pmc2 -o p -- python $R/bench.py --eager --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /tmp/pmc2_$W_$CNT.log 2>&1
  python $R/profiles/pmcstats.py $(find /tmp/pmc2 -name "*.db" | head -1) k_edge k_attn k_rowchain k_atb > $R/gpurun_out/${TAG}_pmc_${W}_${CNT}.json 2>&1
done; done
