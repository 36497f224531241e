for lib in variants/librtd_bw2.so; do
  echo "lib=$lib"
  RTD_LIB=$lib python bench.py --steps 5 --warmup 2 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms_per_step'], d.get('parity'))"
done
