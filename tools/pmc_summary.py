"""Per-kernel averages of the counter_collection csv files written by tools/pmc_passes.sh.
python tools/pmc_summary.py <dir> > summary.json   (the first two dispatches of a kernel are dropped)"""
import collections
import csv
import glob
import json
import os
import re
import sys

out = {}
for path in sorted(glob.glob(os.path.join(sys.argv[1], "*_counter_collection.csv"))):
    tag = os.path.basename(path).replace("_counter_collection.csv", "")
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as f:
        for row in csv.DictReader(f):
            per[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in per.items():
        if "at::" in k or "rocclr" in k:
            continue
        m = re.search(r"(pp_\w+)(<[^(]*>)?", k.replace("(anonymous namespace)::", ""))
        short = (m.group(1) + (m.group(2) or "")) if m else k[:60]
        for c, vals in cs.items():
            v = vals[2:] if len(vals) > 4 else vals
            out.setdefault(tag.split("_")[0] if not tag.startswith("bwd") else "_".join(tag.split("_")[:2]), {}) \
               .setdefault(short, {})[c] = round(sum(v) / len(v), 1)
json.dump(out, sys.stdout, indent=1)
