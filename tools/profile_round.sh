cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_r2b
mkdir -p $out
python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 > $out/stats.log 2>&1
find $out/stats -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
head -6 $out/kernel_stats.csv
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc$i -- python3 bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > $out/pmc$i.log 2>&1
  f=$(find $out/pmc$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $f > $out/pmc$i.txt
  rm -rf $out/pmc$i
done
rm -rf $out/stats
python tools/config_throughput.py > $out/config_throughput.txt 2>&1
cat $out/config_throughput.txt
