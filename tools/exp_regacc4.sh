cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/r3r; mkdir -p $D
for lib in tune; do
PP_LIB=$R/tools/exp/libpphip_$lib.so PP_REGACC_LDS=1 timeout -k 10 120 python3 -u $R/tools/regacc_check.py > $D/check_$lib.txt 2>&1; echo "check $lib rc=$?"; grep -c ok $D/check_$lib.txt; grep FAIL $D/check_$lib.txt
done
run() { PP_LIB=$R/tools/exp/libpphip_$1.so PP_REGACC_LDS=1 PP_REGACC_NSLICE=$2 timeout -k 10 100 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o $1_ns$2 -- python3 $R/tools/regacc_time.py > /dev/null 2>&1; }
run tune 2 && run tune 5 && run tune 3 || exit 1
cd $D && python3 - <<'PY'
import csv,glob
for f in sorted(glob.glob('*_kernel_trace.csv')):
    rows=[r for r in csv.DictReader(open(f)) if 'regacc' in r['Kernel_Name'] or 'dense_reduce' in r['Kernel_Name']]
    out={}
    for r in rows:
        key=(r['Kernel_Name'][:60].replace('void (anonymous namespace)::',''), r.get('Grid_Size_X') or r.get('Grid_Size'))
        out.setdefault(key,[]).append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
    print(f)
    for k,v in out.items(): print('  ',k, round(sorted(v)[len(v)//2],1),'us')
PY
