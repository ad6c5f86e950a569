for e in 2 4 8 16 40; do
  echo "== pedersen_epoch=$e"; ACVM_TUNING="pedersen_epoch=$e" timeout 900 python bench.py --workload arith_pedersen --total-log2 20 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep '"value"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
