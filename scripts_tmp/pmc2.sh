cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_bc
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT" "SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_bc/s$i -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/pmc_bc/s$i.log 2>&1
  f=$(find gpurun_out/pmc_bc/s$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $f | grep "bc_mfma\|eigen" | grep -v "grid=5242880\|grid=1048576 "
done
