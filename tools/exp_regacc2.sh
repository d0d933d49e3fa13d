cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r3p
export PP_LIB=$R/tools/exp/libpphip_tune.so
PP_REGACC_LDS=1 timeout -k 10 120 python3 -u $R/tools/regacc_check.py > $R/gpurun_out/r3p/check_lds.txt 2>&1 || { tail -20 $R/gpurun_out/r3p/check_lds.txt; exit 1; }
PP_REGACC_LDS=1 PP_REGACC_NSLICE=3 timeout -k 10 120 python3 -u $R/tools/regacc_check.py > $R/gpurun_out/r3p/check_lds_ns3.txt 2>&1 || { tail -20 $R/gpurun_out/r3p/check_lds_ns3.txt; exit 1; }
for lds in 0 1; do for ns in 1 2 4; do
  PP_REGACC_LDS=$lds PP_REGACC_NSLICE=$ns timeout -k 10 100 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3p -o lds${lds}_ns$ns -- python3 $R/tools/regacc_time.py > /dev/null 2>&1 || exit 1
done; done
cd $R/gpurun_out/r3p && cat check_lds.txt check_lds_ns3.txt && python3 - <<'PY'
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
