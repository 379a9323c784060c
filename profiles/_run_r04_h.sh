cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
./profiles/_exp/mfma32b
cp equidock_public_amd/libequidock_hip.so /tmp/lib32.so
cp profiles/_exp/libKEEP.so equidock_public_amd/libequidock_hip.so
echo "--- K=32 library, operands kept live past the MFMA (edge kernels)"
python -m pytest tests -m gpu -q -k "edge_message_with_dropout" 2>&1 | tail -3 | cut -c1-200
cp /tmp/lib32.so equidock_public_amd/libequidock_hip.so
