for lib in variants/librtd_inl_g64.0.so variants/librtd_inl_g512.0.so; do
  echo $lib
  RTD_LIB=$lib python -m pytest tests -q -m gpu 2>&1 | grep -v "^Hostname\|^Librccl\|version" | tail -2
  RTD_LIB=$lib python bench.py --steps 10 --warmup 3 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', round(d['value']), round(d['roofline']['kernel_ms_per_step']['eigen'],3), round(d['roofline']['kernel_ms_per_step']['bc'],3), d['parity']['max_abs_dI'], d['parity']['max_rel_dI'])"
done
