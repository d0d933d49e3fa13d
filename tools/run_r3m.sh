mkdir -p gpurun_out/r3m
timeout -k 10 300 python -u -m pytest tests/test_parallel_gpu.py tests/test_regress_gpu.py tests/test_zz_no_alloc.py -x -q --timeout 180 --timeout-method thread > gpurun_out/r3m/tests.txt 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r3m/tests.txt
timeout -k 10 400 python -u tests/fuzz_parity.py --seconds 200 --seed 7 --forced --fused --out gpurun_out/r3m/fuzz_seed7.txt > gpurun_out/r3m/fuzz_stdout.txt 2>&1
echo "fuzz rc=$?"; tail -3 gpurun_out/r3m/fuzz_stdout.txt
timeout -k 10 280 python bench.py > gpurun_out/r3m/bench.json 2> gpurun_out/r3m/bench.err; echo "bench rc=$?"; python -c "
import json; d=json.load(open('gpurun_out/r3m/bench.json')); print({k:d[k] for k in ('value','ms_per_step','eager_ms_per_step','hipgraph_ms_per_step','train_chars_per_s')}, d['config']['launch'])"
