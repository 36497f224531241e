cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r1v5
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r1v5/stats2 -- python3 bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2 > gpurun_out/r1v5/stats2.log 2>&1
find gpurun_out/r1v5/stats2 -name '*kernel_stats.csv' | head -1 | xargs head -12
tail -c 900 gpurun_out/r1v5/stats2.log
