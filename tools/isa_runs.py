"""Instruction-type runs (M mfma, V valu, D lds, G vmem, W waitcnt, B barrier, S scalar) of the
basic blocks of one kernel that hold at least `minm` MFMAs:
python tools/isa_runs.py file.s '<regex of the mangled kernel name>' [minm]"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
pat = re.compile(sys.argv[2])
minm = int(sys.argv[3]) if len(sys.argv) > 3 else 20
start = [i for i, l in enumerate(lines) if l.startswith("_Z") and ":" in l and pat.search(l)][0]
end = [i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end")][0]
seq = []
for l in lines[start:end]:
    t = l.strip()
    if not t or t.startswith(";"):
        continue
    if t.startswith(".LBB"):
        seq.append(("L", t.split(":")[0]))
        continue
    if t.startswith("."):
        continue
    op = t.split()[0]
    c = ("M" if op.startswith("v_mfma") else "V" if op.startswith("v_") else
         "D" if op.startswith("ds_") else "G" if op.startswith(("buffer_", "global_")) else
         "W" if op.startswith("s_waitcnt") else "B" if op.startswith("s_barrier") else "S")
    seq.append((c, op))
out, prev, cnt = [], None, 0
for c, op in seq:
    if c == "L":
        if prev:
            out.append(f"{prev}{cnt}")
        prev, cnt = None, 0
        out.append("\n" + op + "\n")
        continue
    if c == prev:
        cnt += 1
    else:
        if prev:
            out.append(f"{prev}{cnt}")
        prev, cnt = c, 1
if prev:
    out.append(f"{prev}{cnt}")
for seg in " ".join(out).split("\n.LBB"):
    if seg.count("M") >= minm:
        print(".LBB" + seg[:2400])
        print("=====")
