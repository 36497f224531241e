# Round-4 evidence (run on the GPU box: bash tools/profile_round4.sh):
#  a) cfg4 headline workload, windows one after the other (RTD_NO_PIPELINE=1): rocprofv3 --stats per-kernel averages (they must agree
#     with the HIP-event pass bench.py reports in roofline.kernel_ms_per_launch) + PMC passes for traffic and SQ counters
#  b) cfg5 (64 streams) with the lean two-wavefront boundary-condition kernel: stats + PMC passes of one 128-column window,
#     and the same with RTD_BC_TILE_V1=1 (the one-wavefront kernel of rounds 2-3)
#  c) the bench line itself
cd $GRAFT_REPO_ROOT
export RTD_NO_PIPELINE=1
bash tools/profile_pmc.sh prof_r4_cfg4_serial python3 bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 --total-columns 16384 > gpurun_out/prof_r4_cfg4_serial.txt 2>&1
unset RTD_NO_PIPELINE
RTD_EXTRA_PMC="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" bash tools/profile_pmc.sh prof_r4_cfg5 python3 tools/profile_config.py cfg5 128 0 2 > gpurun_out/prof_r4_cfg5.txt 2>&1
RTD_BC_TILE_V1=1 bash tools/profile_pmc.sh prof_r4_cfg5_v1 python3 tools/profile_config.py cfg5 128 0 2 > gpurun_out/prof_r4_cfg5_v1.txt 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_prof.json 2> gpurun_out/r04_bench_prof.err
