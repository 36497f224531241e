#!/usr/bin/env bash
# Round-4 evidence (run on the GPU box: bash tools/profile_round4.sh):
#  a) cfg4 headline workload, windows one after the other (RTD_NO_PIPELINE=1): rocprofv3 --stats per-kernel averages (they must agree
#     with the HIP-event pass bench.py reports in roofline.kernel_ms_per_launch) + PMC passes for traffic and SQ counters
#  b) cfg5 (64 streams) with the lean two-wavefront boundary-condition kernel: stats + PMC passes of one 128-column window,
#     and the same with RTD_BC_TILE_V1=1 (the one-wavefront kernel of rounds 2-3)
#  c) the bench line itself
set -euo pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
mkdir -p gpurun_out
export RTD_NO_PIPELINE=1
bash tools/profile_pmc.sh prof_r4_cfg4_serial python3 bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 --total-columns 16384 > gpurun_out/prof_r4_cfg4_serial.txt 2>&1
unset RTD_NO_PIPELINE
RTD_EXTRA_PMC="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" bash tools/profile_pmc.sh prof_r4_cfg5 python3 tools/profile_config.py cfg5 128 0 2 > gpurun_out/prof_r4_cfg5.txt 2>&1
RTD_BC_TILE_V1=1 bash tools/profile_pmc.sh prof_r4_cfg5_v1 python3 tools/profile_config.py cfg5 128 0 2 > gpurun_out/prof_r4_cfg5_v1.txt 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_prof.json 2> gpurun_out/r04_bench_prof.err
#  d) the 66 ... 128-stream path (profiles/archive/r04_many_streams.json, r04_many_stream_kernel_stats.csv, r04_pmc_many_streams.txt):
#     stage times with the round's kernels and with the row-per-lane BC kernels, kernel stats and counters of 24 columns of 128 x 50 x 64
python tools/many_stream_timing.py 24 > gpurun_out/many_stream_new.txt 2>&1
RTD_BC_WIDE_V1=1 python tools/many_stream_timing.py 24 > gpurun_out/many_stream_v1.txt 2>&1
RTD_EXTRA_PMC="SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA" bash tools/profile_pmc.sh prof_r4_many_streams python3 tools/many_stream_timing.py 24 > gpurun_out/prof_r4_many_streams.txt 2>&1
