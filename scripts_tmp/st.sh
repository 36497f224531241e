for t in 0.25f 0.0625f 0.01f; do
echo "thr $t"
RTD_BC_STATS=1 RTD_LIB=variants/librtd_t$t.so python bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | grep "GJ calls\|metric" | head -3 | cut -c1-200
done
