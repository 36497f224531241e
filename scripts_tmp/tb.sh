python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^Hostname\|^Librccl\|version" | tail -2
python bench.py --steps 5 --warmup 2 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms_per_step'], d.get('parity'))"
