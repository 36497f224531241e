python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^Hostname\|^Librccl\|version" | tail -2
python bench.py --steps 5 --warmup 2 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['parity'])"
for rep in 1 2; do
for lib in pythonic-disort_amd/pydisort_amd/librtd.so variants/librtd_prev.so; do
  RTD_LIB=$lib python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', round(d['value']), round(d['roofline']['kernel_ms_per_step']['eigen'],3), round(d['roofline']['kernel_ms_per_step']['bc'],3))"
done
done
