# rocprofv3 passes over one program: kernel stats + separate --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass).
# usage: bash tools/profile_pmc.sh <outdir under gpurun_out> <program and args ...>     (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/$1; shift
mkdir -p $out
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- "$@" > $out/stats.log 2>&1
find $out/stats -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
rm -rf $out/stats
head -8 $out/kernel_stats.csv
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU GRBM_GUI_ACTIVE" ${RTD_EXTRA_PMC:+"$RTD_EXTRA_PMC"}; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc$i -- "$@" > $out/pmc$i.log 2>&1
  f=$(find $out/pmc$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $f > $out/pmc$i.txt
  rm -rf $out/pmc$i
done
cat $out/pmc*.txt | grep -v "^rtd_tables\|^rtd_prep" | cut -c1-600
