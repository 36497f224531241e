#!/usr/bin/env bash
# Round-6 evidence (run on the GPU box: bash tools/profile_round6.sh; the files it leaves under gpurun_out/r06/ are what
# profiles/r06_* are copied from).  Every step's exit code goes to gpurun_out/r06/status.txt and the script exits non-zero when a
# bench or profiler leg failed (round-5 advice: `|| true` behind every step let stale files through unnoticed).
#  a) cfg4 headline workload, windows one after the other (RTD_NO_PIPELINE=1): rocprofv3 --kernel-trace --stats per-kernel
#     averages (they must agree with the HIP-event pass of the bench line) + the --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ busy /
#     wait counters, FP64 instruction counters)
#  b) cfg5 (64 streams): the same for the bench line's own workload -- 10^4 columns in 79 windows of 128, three passes: 237 launches
#     per kernel, long enough for the chip to settle at the clock it sustains -- so that the committed summary reproduces the bench
#     line's figure (round-5 verdict: 3 launches incl. the cold one were not a summary; a 24-launch run from a cool chip read 4 % fast)
#  c) the bench line itself
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out=gpurun_out/r06
mkdir -p $out
: > $out/status.txt
fail=0
step() {  # step <name> <critical 0|1> <command ...>: runs it, records the exit code
  local name=$1 crit=$2; shift 2
  "$@"
  local rc=$?
  echo "$name rc=$rc" >> $out/status.txt
  if [ $rc -ne 0 ] && [ $crit -eq 1 ]; then fail=1; fi
  return 0
}
pmc() {  # pmc <outdir under gpurun_out> <log> <program ...>
  local dir=$1 log=$2; shift 2
  bash tools/profile_pmc.sh $dir "$@" > $log 2>&1 && [ -s gpurun_out/$dir/kernel_stats.csv ] && [ -s gpurun_out/$dir/pmc1.txt ] && [ -s gpurun_out/$dir/pmc2.txt ]
}
export RTD_NO_PIPELINE=1
step cfg4_serial_profile 1 pmc r06/cfg4_serial $out/cfg4_serial.txt python3 bench.py --no-cpu-baseline --no-extras --no-live-traffic --steps 2 --warmup 1 --total-columns 16384
step cfg5_profile 1 pmc r06/cfg5 $out/cfg5.txt python3 tools/profile_config.py cfg5 10000 128 2
unset RTD_NO_PIPELINE
if [ -s $out/cfg4_serial/pmc1.txt ]; then
  step cfg4_traffic_json 1 python3 tools/pmc_to_json.py $out/pmc_traffic.json "rocprofv3 --pmc passes of tools/profile_round6.sh (cfg4, windows of 256 columns one after the other)" --columns-per-launch 256 $out/cfg4_serial/pmc*.txt
  step cfg5_traffic_json 1 python3 tools/pmc_to_json.py $out/pmc_traffic_cfg5.json "rocprofv3 --pmc passes of tools/profile_round6.sh (cfg5, windows of 128 columns one after the other)" --columns-per-launch 128 $out/cfg5/pmc*.txt
fi
step single_column_latency 0 bash -c "python tools/single_column_latency.py > $out/single_column_latency.txt 2>&1"
step bench 1 bash -c "python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err && [ -s $out/bench.json ]"
tail -2 $out/bench.err
cat $out/status.txt
exit $fail
