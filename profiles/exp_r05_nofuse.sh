cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rm -rf /tmp/pnf; EQD_FUSE_GATHER=0 rocprofv3 --kernel-trace --stats -d /tmp/pnf -o h -- python $R/bench.py --workload C --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $R/gpurun_out/r05_o_nofuse.log 2>&1
DB=$(find /tmp/pnf -name "*.db" | head -1)
python $R/profiles/summarize.py $DB $R/gpurun_out/r05_o_kernels_C_bf16_nofuse.md "round 5 (r05_o): C bf16, EQD_FUSE_GATHER=0" "EQD_FUSE_GATHER=0 rocprofv3 ... bench.py --workload C --dtype bf16" > $R/gpurun_out/r05_o_kernels_nofuse.txt 2>&1
