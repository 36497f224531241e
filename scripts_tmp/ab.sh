for lib in variants/librtd_bw4.so variants/librtd_bw2.so; do
  echo "lib=$lib"
  RTD_LIB=$lib python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms_per_step']['bc'])"
done
