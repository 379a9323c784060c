cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
K="edge_message or cross_attention or linear or kabsch or keypoint or golden_case"
echo "--- both (DPP + permlane swaps)"
python -m pytest tests -m gpu -q --tb=line -k "$K" 2>&1 | grep -v Warning | tail -25 | cut -c1-230
cp equidock_public_amd/libequidock_hip.so /tmp/lib_both.so
for v in NO_DPP NO_PLSWAP; do
  cp profiles/_exp/lib_$v.so equidock_public_amd/libequidock_hip.so
  echo "--- variant $v"
  python -m pytest tests -m gpu -q --tb=line -k "$K" 2>&1 | grep -v Warning | tail -6 | cut -c1-230
done
cp /tmp/lib_both.so equidock_public_amd/libequidock_hip.so
