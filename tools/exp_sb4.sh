cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/r3ac; mkdir -p $D
for v in base sb4; do
  if [ $v = sb4 ]; then export PP_LIB=$R/tools/exp/libpphip_sb4.so; fi
  timeout -k 10 100 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o $v -- python3 $R/tools/run_one.py bwd_w 16384 cfg2_hidden > /dev/null 2>&1 || exit 1
  echo "== $v"; python3 $R/tools/trace_medians.py $D/${v}_kernel_trace.csv 2>/dev/null | grep -v "at::"
done
