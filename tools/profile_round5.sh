#!/usr/bin/env bash
# Round-5 evidence (run on the GPU box: bash tools/profile_round5.sh; the files it leaves under gpurun_out/r05/ are what
# profiles/r05_* were copied from):
#  a) cfg4 headline workload, windows one after the other (RTD_NO_PIPELINE=1): rocprofv3 --kernel-trace --stats per-kernel averages
#     (they must agree with the HIP-event pass of the bench line, roofline.kernel_ms_per_launch) + the --pmc passes: FETCH_SIZE,
#     WRITE_SIZE, SQ busy / wait counters and -- new in round 5 -- the FP64 instruction counters (executed FLOPs)
#  b) cfg5 (64 streams): the same for one window of 128 columns (the eigen kernel with the rows of L spread over the lanes)
#  c) s_memtime stamps per phase of rtd_eigen_kernel<32, 2> (librtd_stamps.so: python tools/build_variant.py stamps -DRTD_EIG_STAMPS)
#  d) one-column latency, its breakdown, the bench line itself
set -euo pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out=gpurun_out/r05
mkdir -p $out
export RTD_NO_PIPELINE=1
bash tools/profile_pmc.sh r05/cfg4_serial python3 bench.py --no-cpu-baseline --no-extras --no-live-traffic --steps 2 --warmup 1 --total-columns 16384 > $out/cfg4_serial.txt 2>&1 || true
unset RTD_NO_PIPELINE
bash tools/profile_pmc.sh r05/cfg5 python3 tools/profile_config.py cfg5 128 0 2 > $out/cfg5.txt 2>&1 || true
if [ -f pythonic-disort_amd/pydisort_amd/librtd_stamps.so ]; then
  RTD_LIB=$PWD/pythonic-disort_amd/pydisort_amd/librtd_stamps.so python tools/profile_config.py cfg5 128 0 1 2>&1 | python tools/eig_phase_cycles.py > $out/eigen32_phases.txt || true
fi
python tools/single_column_latency.py > $out/single_column_latency.txt 2>&1 || true
python tools/latency_breakdown.py >> $out/single_column_latency.txt 2>&1 || true
python tools/many_stream_timing.py 24 > $out/many_streams.txt 2>&1 || true
python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err || true
tail -2 $out/bench.err
