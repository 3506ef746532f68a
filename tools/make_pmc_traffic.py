#!/usr/bin/env python3
"""profiles/pmc_traffic.json from a tools/pmc_summary.py listing: HBM bytes per launch of the headline kernel, stamped with the
digest of everything the library was built from (sources, headers, flags; bench.py quotes the record only while that, the layout and the size match).
usage: make_pmc_traffic.py <pmc_summary.txt> <clusters> <layout> <kernel name prefix>"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

path, pairs, layout, kernel = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
cur = None
vals = {}
for line in open(path):
    if not line.startswith(" "):
        cur = line.strip()
        continue
    if cur and cur.startswith(kernel):
        f = line.split()
        vals.setdefault(cur, {})[f[0]] = float(f[-1].split("=")[1])
name, v = max(vals.items(), key=lambda kv: kv[1].get("FETCH_SIZE", 0))
rd = v["FETCH_SIZE"] * 1024 * 2
wr = v["WRITE_SIZE"] * 1024
print(json.dumps({
    "pairs": pairs, "layout": layout, "kernel": name, "FETCH_SIZE_KB": v["FETCH_SIZE"], "WRITE_SIZE_KB": v["WRITE_SIZE"],
    "correction": "hbm_bytes = FETCH_SIZE*1024*2 + WRITE_SIZE*1024 (FETCH_SIZE counts 64 B per 128 B request on gfx950; WRITE_SIZE is exact); separate --pmc passes, mean over the dispatches",
    "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
    "algorithmic_bytes_per_launch": bench.BYTES_PER_PAIR * pairs,
    "ratio_to_algorithmic": round((rd + wr) / (bench.BYTES_PER_PAIR * pairs), 4),
    "library_inputs_sha256_16": bench.kernel_sources_digest(),      # every source and header of the library + the compiler flags
    "source": "tools/profile_bench.sh -> tools/pmc_summary.py -> tools/make_pmc_traffic.py"}, indent=1))
