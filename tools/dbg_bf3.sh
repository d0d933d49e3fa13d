R=$GRAFT_REPO_ROOT
for v in 11 12 21 22; do
  export PP_BF3_VARIANT=$v
  for B in 4096 16384 65536; do
  echo -n "variant=$v "
  timeout -k 10 60 $R/tools/fwd_bench $B 250 250 3 2 0 3 200 || exit 1
  done
done
unset PP_BF3_VARIANT
timeout -k 10 300 python -u -m pytest tests/test_parity_gpu.py -x -q -k "bf16x3" --timeout 120 --timeout-method thread 2>&1 | tail -3
