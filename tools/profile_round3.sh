# Round-3 evidence (run on the GPU box: bash tools/profile_round3.sh):
#  a) cfg4 headline workload, windows one after the other (RTD_NO_PIPELINE=1): the per-kernel averages of rocprofv3 --stats agree
#     with the HIP-event pass bench.py reports in roofline.kernel_ms_per_launch; PMC passes for traffic and SQ counters
#  b) the same workload as bench.py runs it by default (window pipeline on two streams): kernel durations overlap
#  c) cfg5 (64 streams): stats + PMC passes of one 96-column window
cd $GRAFT_REPO_ROOT
export RTD_NO_PIPELINE=1
bash tools/profile_pmc.sh prof_r3_cfg4_serial python3 bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 --total-columns 16384 > gpurun_out/prof_r3_cfg4_serial.txt 2>&1
unset RTD_NO_PIPELINE
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_r3_cfg4_pipe
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r3_cfg4_pipe/stats -- python3 bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 --total-columns 16384 > gpurun_out/prof_r3_cfg4_pipe/stats.log 2>&1
find gpurun_out/prof_r3_cfg4_pipe/stats -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} gpurun_out/prof_r3_cfg4_pipe/kernel_stats.csv
rm -rf gpurun_out/prof_r3_cfg4_pipe/stats
RTD_EXTRA_PMC="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" bash tools/profile_pmc.sh prof_r3_cfg5 python3 tools/profile_config.py cfg5 96 0 2 > gpurun_out/prof_r3_cfg5.txt 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_prof.json 2> gpurun_out/r03_bench_prof.err
