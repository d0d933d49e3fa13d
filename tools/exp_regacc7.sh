cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/r3ab; mkdir -p $D
export PP_LIB=$R/tools/exp/libpphip_tune.so
for x in 0 1; do
  PP_REGACC_XCD=$x timeout -s KILL 60 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D -o fetch$x -- python3 $R/tools/regacc_time.py > /dev/null 2>&1 || exit 1
  python3 - <<PY
import csv, collections
per=collections.defaultdict(list)
for r in csv.DictReader(open("$D/fetch${x}_counter_collection.csv")):
    if "regacc" in r["Kernel_Name"] and "prep" not in r["Kernel_Name"]:
        per[(r["Kernel_Name"][28:75], r["Grid_Size"])].append(float(r["Counter_Value"]))
print("XCD=$x")
for k,v in per.items(): print("  ", k, "FETCH_SIZE KB", round(sum(v)/len(v)))
PY
done
