#!/bin/bash
# timing experiment: kernel time with phases switched off (results invalid, timings only)
for m in 255 254 253 251 247 239 224 0; do
  echo -n "mask $m: "
  HRG_PHASE_MASK=$m python bench.py --steps 100 --warmup 100 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'])"
done
