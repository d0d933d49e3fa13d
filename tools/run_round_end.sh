# Full GPU suite, bench, bench under rocprofv3 (kernel stats): what the profiles/rNN_*_bench* files come from.
#   gpurun --timeout 1200 -- bash tools/run_round_end.sh
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/round_end; mkdir -p $D
cd $R
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > $D/gpu_tests.txt 2>&1 || { tail -30 $D/gpu_tests.txt; exit 1; }
tail -3 $D/gpu_tests.txt
timeout -k 10 400 python3 bench.py > $D/bench.json 2> $D/bench.err || { tail -20 $D/bench.err; exit 1; }
cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o bench -- python3 $R/bench.py --steps 20 --warmup 5 > $D/bench_under_rocprof.json 2> $D/rocprof.err || { tail -20 $D/rocprof.err; exit 1; }
python3 - <<PY
import json
d=json.load(open("$D/bench.json"))
print(d["ms_per_step"], d["value"], d.get("eager_ms_per_step"), d.get("hipgraph_ms_per_step"), d["train_chars_per_s"])
for r in d["roofline_hbm"]: print(r["layer"], r["rows"], r["launch_us"], r["frac"], r["auto_methods"])
print(d["roofline"]["frac"], d["roofline"]["launch_us"])
print("char_table", d.get("char_table"))
PY
