cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/r3aa; mkdir -p $D
export PP_LIB=$R/tools/exp/libpphip_tune.so
timeout -k 10 120 python3 -u $R/tools/regacc_check.py | grep -c " ok" || exit 1
for x in 0 1; do
  PP_REGACC_XCD=$x timeout -k 10 100 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o xcd$x -- python3 $R/tools/regacc_time.py > /dev/null 2>&1 || exit 1
  echo "== XCD=$x"; python3 $R/tools/trace_medians.py $D/xcd${x}_kernel_trace.csv 2>/dev/null | grep "regacc_\(lds_\)\?kernel"
done
