cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
./profiles/_exp/mfma32
echo "--- K=32 library"
python -m pytest tests -m gpu -q -k "edge_message_with_dropout" 2>&1 | tail -3 | cut -c1-200
cp equidock_public_amd/libequidock_hip.so /tmp/lib32.so
cp profiles/_exp/libK16.so equidock_public_amd/libequidock_hip.so
echo "--- K=16 fallback library (same sources, -DEQD_MFMA_K16)"
python -m pytest tests -m gpu -q -k "edge_message_with_dropout" 2>&1 | tail -3 | cut -c1-200
python bench.py --workload C --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-roofline 2>&1 | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('K16 C bf16', d['value'], d['ms_per_step'])"
cp /tmp/lib32.so equidock_public_amd/libequidock_hip.so
