cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/r3q; mkdir -p $D
run() { PP_LIB=$R/tools/exp/libpphip_$1.so PP_REGACC_LDS=1 PP_REGACC_NSLICE=$2 timeout -k 10 100 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o $1_ns$2 -- python3 $R/tools/regacc_time.py > /dev/null 2>&1; }
run tune 2 && run tune 5 && run tune 10 && run nowait 2 && run nobarrier 2 && run both 2 && run both 5 || exit 1
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
