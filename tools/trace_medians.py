"""Median kernel duration per (kernel, grid) from a rocprofv3 --kernel-trace csv, in launch order:
python tools/trace_medians.py <kernel_trace.csv> [substring]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
want = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.OrderedDict()
for r in rows:
    m = re.search(r"(pp_\w+)(<[^(]*>)?", r["Kernel_Name"])
    name = m.group(0) if m else r["Kernel_Name"][:60]
    if want not in name:
        continue
    key = (name[:72], r["Grid_Size_X"], r["Grid_Size_Y"])
    agg.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (name, gx, gy), v in agg.items():
    v.sort()
    print(f"{name:72s} grid {gx:>8s}x{gy:<3s} n={len(v):3d} median {v[len(v) // 2]:8.2f} us")
