cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
./profiles/_exp/mfma32
python -m pytest tests -m gpu -q -k "bf16 or dropout or head_backward" 2>&1 | tail -25 | cut -c1-250 > gpurun_out/r04_f_pytest_bf16.log; cat gpurun_out/r04_f_pytest_bf16.log
