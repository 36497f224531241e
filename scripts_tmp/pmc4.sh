cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export RTD_LIB=$GRAFT_REPO_ROOT/variants/librtd_g8.so
mkdir -p gpurun_out/pmc5
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU"; do
timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc5/s1 -- python3 bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > gpurun_out/pmc5/s1.log 2>&1
f=$(find gpurun_out/pmc5/s1 -name '*counter_collection.csv' | sort | tail -1)
[ -n "$f" ] && python tools/pmc_summary.py $f | grep "bc_mfma" || tail -5 gpurun_out/pmc5/s1.log
rm -rf gpurun_out/pmc5/s1
done
