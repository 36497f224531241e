cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r1v5
python bench.py > gpurun_out/r1v5/bench.json 2> gpurun_out/r1v5/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r1v5/stats -- python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 > gpurun_out/r1v5/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/r1v5/pmc_$c -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/r1v5/pmc_$c.log 2>&1
done
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d gpurun_out/r1v5/pmc_sq -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/r1v5/pmc_sq.log 2>&1
python tools/pmc_summary.py $(find gpurun_out/r1v5 -name '*counter_collection.csv') | grep -v tables
find gpurun_out/r1v5/stats -name '*kernel_stats.csv' | head -1 | xargs head -12
tail -c 600 gpurun_out/r1v5/bench.json
