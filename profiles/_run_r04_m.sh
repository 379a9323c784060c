# round-4 measurement set on HEAD: kernel tables + timelines, then PMC counters (one counter per pass) for B, C, C bf16, E, R
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0
bash profiles/measure_r04.sh r04_m prof:B prof:C prof:C:bf16 prof:E prof:R > gpurun_out/r04_m_measure_prof.log 2>&1
bash profiles/measure_r04.sh r04_m pmc:B pmc:C pmc:C:bf16 pmc:E pmc:R > gpurun_out/r04_m_measure_pmc.log 2>&1
ls gpurun_out | grep r04_m | wc -l
