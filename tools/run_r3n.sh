# profiles for round 3: kernel stats of the bench step, PMC passes of the tile forward on cfg3 hidden / cfg4 input
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3n
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --no-cpu-baseline --no-hbm-layers --train-steps 0 > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
B="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM"
C="TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
for L in cfg3_hidden cfg4_input; do
  timeout -s KILL 90 rocprofv3 --pmc $A --output-format csv -d $OUT -o ${L}_fwd_A -- python3 $GRAFT_REPO_ROOT/tools/run_one.py fwd 16384 $L > /dev/null 2>&1
  timeout -s KILL 90 rocprofv3 --pmc $B --output-format csv -d $OUT -o ${L}_fwd_B -- python3 $GRAFT_REPO_ROOT/tools/run_one.py fwd 16384 $L > /dev/null 2>&1
  timeout -s KILL 90 rocprofv3 --pmc $C --output-format csv -d $OUT -o ${L}_fwd_C -- python3 $GRAFT_REPO_ROOT/tools/run_one.py fwd 16384 $L > /dev/null 2>&1
  timeout -s KILL 90 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o ${L}_fwd_FETCH -- python3 $GRAFT_REPO_ROOT/tools/run_one.py fwd 16384 $L > /dev/null 2>&1
  timeout -s KILL 90 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT -o ${L}_fwd_WRITE -- python3 $GRAFT_REPO_ROOT/tools/run_one.py fwd 16384 $L > /dev/null 2>&1
  echo "$L done"
done
ls $OUT | head -40
