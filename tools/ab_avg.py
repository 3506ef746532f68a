#!/usr/bin/env python3
"""Average the rows of a tools/lut_cold_ab.py log per (sheet, rows, variant).  usage: python tools/ab_avg.py <log>"""
import re
import sys
from collections import defaultdict

acc = defaultdict(list)
for line in open(sys.argv[1]):
    m = re.match(r"S=\s*(\d+) n=\s*(\d+) sets=(\d+) (\S+)\s+cold\s+([\d.]+) us ([\d.]+)\s+warm\s+([\d.]+) us ([\d.]+)", line)
    if m:
        acc[(int(m.group(1)), int(m.group(2)), m.group(4))].append(tuple(float(m.group(i)) for i in (5, 6, 7, 8)))
for k in sorted(acc):
    v = acc[k]
    a = [sum(x[i] for x in v) / len(v) for i in range(4)]
    print(f"S={k[0]:4d} n={k[1]:9d} {k[2]:50s} cold {a[0]:8.2f} us {a[1]:.3f}   warm {a[2]:8.2f} us {a[3]:.3f}")
