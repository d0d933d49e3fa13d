cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in tune noloads; do for ns in 1 2 4; do
  PP_LIB=$R/tools/exp/libpphip_$lib.so PP_REGACC_NSLICE=$ns timeout -k 10 100 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3o -o ${lib}_ns$ns -- python3 $R/tools/regacc_time.py > /dev/null 2>&1 || exit 1
done; done
cd $R/gpurun_out/r3o && python3 - <<'PY'
import csv,glob
for f in sorted(glob.glob('*_kernel_trace.csv')):
    rows=[r for r in csv.DictReader(open(f)) if 'regacc_kernel' in r['Kernel_Name'] or 'dense_reduce' in r['Kernel_Name']]
    # group by grid size (shape) in order
    out={}
    for r in rows:
        key=(r['Kernel_Name'][:40].replace('void (anonymous namespace)::',''), r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size'))
        out.setdefault(key,[]).append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
    print(f)
    for k,v in out.items(): print('  ',k, round(sorted(v)[len(v)//2],1),'us')
PY
