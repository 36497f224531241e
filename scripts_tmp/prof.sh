cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r1v6
python bench.py > gpurun_out/r1v6/bench.json 2> gpurun_out/r1v6/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r1v6/stats -- python3 bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 > gpurun_out/r1v6/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/r1v6/pmc_$c -- python3 bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > gpurun_out/r1v6/pmc_$c.log 2>&1
done
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d gpurun_out/r1v6/pmc_sq -- python3 bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > gpurun_out/r1v6/pmc_sq.log 2>&1
python tools/pmc_summary.py $(find gpurun_out/r1v6 -name '*counter_collection.csv') | grep -v tables
find gpurun_out/r1v6/stats -name '*kernel_stats.csv' | head -1 | xargs head -5
python -c "
import json; d=json.load(open('gpurun_out/r1v6/bench.json')); print(d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_per_step'], d['parity'], d['cpu_baseline']['value'], d['only_flux']['value'])"
