#!/usr/bin/env python3
"""Fold an earlier round's profile set (profiles/<tag>_*) into ONE summary file, profiles/<tag>_summary.md: the bench line's figures, and the
small text summaries as they are (kernel stats, PMC summaries, stamps, rates); the files themselves stay in the git history.
usage: python tools/fold_profiles.py r04 [--remove]"""
import glob
import json
import os
import sys

tag = sys.argv[1]
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
files = sorted(f for f in glob.glob(os.path.join(root, tag + "_*")) if not f.endswith("_summary.md"))
out = [f"# {tag}: what this round's profile set said (folded later: the {len(files)} files it was are in the git history)\n"]
for f in files:
    name = os.path.basename(f)
    text = open(f, errors="replace").read()
    if name.endswith(".json"):
        try:
            d = json.loads(text.strip().splitlines()[-1])
        except Exception:
            continue
        if "value" in d:
            r = d.get("roofline") or {}
            out.append(f"* `{name}`: {d['value']} {d.get('unit', '')}, {d.get('ms_per_step')} ms per step on {d.get('n_gpus')} GPU(s); roofline {r.get('achieved')} GB/s = {r.get('frac')} "
                       f"(as placed {r.get('frac_as_placed')}), kernel {r.get('kernel_ms')} ms, traffic {r.get('traffic')}; cpu_baseline {(d.get('cpu_baseline') or {}).get('value')}")
            for x in (d.get("extra") or {}).get("rates", []):
                out.append(f"  * {x['config']}: {x['ms']} ms, frac {x['frac']}" + (f" (pipelined {x['frac_pipelined']}, replayed {x['frac_warm']})" if "frac_warm" in x else ""))
        elif "hbm_bytes_per_launch" in d:
            out.append(f"* `{name}`: HBM traffic {d['hbm_bytes_per_launch']:.0f} B per launch = {d['ratio_to_algorithmic']} x algorithmic ({d.get('kernel')})")
        continue
    if name.endswith("trace_head.csv"):
        continue                                                  # (the first rows of a trace: nothing a summary needs)
    lines = text.rstrip().splitlines()
    keep = lines if len(lines) <= 60 else lines[:40] + [f"... ({len(lines) - 40} more lines)"]
    out.append(f"* `{name}`:\n```\n" + "\n".join(l[:220] for l in keep) + "\n```")
open(os.path.join(root, tag + "_summary.md"), "w").write("\n".join(out) + "\n")
print(f"{tag}: {len(files)} files -> profiles/{tag}_summary.md ({sum(len(x) for x in out)} bytes)")
if "--remove" in sys.argv:
    for f in files:
        os.remove(f)
