#!/usr/bin/env python3
"""Install what `bash profiles/collect.sh <tag>` left under gpurun_out/ into profiles/: the tag's kernel-stats / PMC
summaries and its pmc_roofline.json entry (replacing an older entry of the same tag).

    python3 profiles/install_entries.py r3 r3_uint4 r3_mixed80
"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
path = os.path.join(P, "pmc_roofline.json")
entries = json.load(open(path))
for tag in sys.argv[1:]:
    for suffix in ("_kernel_stats.csv", "_pmc.csv", "_trace_bench.json"):
        if suffix == "_trace_bench.json" and not os.path.exists(os.path.join(G, tag + suffix)):
            continue
        shutil.copy(os.path.join(G, tag + suffix), os.path.join(P, tag + suffix))
    new = json.load(open(os.path.join(G, tag + "_pmc_roofline.json")))
    entries = [e for e in entries if e.get("tag") != tag] + [new]
    print(tag, new.get("source"))
json.dump(entries, open(path, "w"), indent=1)
open(path, "a").write("\n")
