cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc3
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_INSTS_SMEM SQ_WAVES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc3/s$i -- python3 bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > gpurun_out/pmc3/s$i.log 2>&1
  f=$(find gpurun_out/pmc3/s$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $f | grep "bc_mfma\|eigen"
done
