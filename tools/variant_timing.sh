#!/bin/bash
# tuning experiment: time prebuilt library variants (human-robot-gym_amd/variant_*.so)
for v in "$@"; do
  echo -n "$v: "
  python bench.py --variant-lib $PWD/human-robot-gym_amd/variant_$v.so --steps 100 --warmup 100 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'])"
done
