#!/bin/bash
# Sweep the bench arena pad (bytes added between matrices) and print kernel ms per pad.
cd "$(dirname "$0")/.."
for a in "$@"; do
  python bench.py --cpu-sample 0 --steps 8 --warmup 2 --arena $a 2>/dev/null |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('arena', sys.argv[1], d['roofline']['kernel_ms'], d['roofline']['frac'])" $a
done
