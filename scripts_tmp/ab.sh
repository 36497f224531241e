for lib in variants/librtd_tol1e-11.so variants/librtd_tol1e-9.so variants/librtd_tol1e-7.so; do
  echo "lib=$lib"
  RTD_LIB=$lib python -m pytest tests -q -m gpu 2>&1 | grep -v "^Hostname\|^Librccl\|version" | tail -2
  RTD_LIB=$lib python bench.py --steps 5 --warmup 2 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms_per_step']['eigen'], d['config'].get('max_jacobi_sweeps'), d['parity'])"
done
