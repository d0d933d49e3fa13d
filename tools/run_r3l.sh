mkdir -p gpurun_out/r3l
timeout -k 10 200 python -u tools/layer_bench.py --batch 256 1000 --only cfg3_hidden > gpurun_out/r3l/lb_small.txt 2>&1 || exit 1
timeout -k 10 200 python -u tools/layer_bench.py --batch 256 1000 --only cfg4_input >> gpurun_out/r3l/lb_small.txt 2>&1 || exit 1
timeout -k 10 400 python -u tests/fuzz_parity.py --seconds 240 --seed 7 --forced --fused --out gpurun_out/r3l/fuzz_seed7.txt > gpurun_out/r3l/fuzz_stdout.txt 2>&1
echo "fuzz rc=$?"
tail -4 gpurun_out/r3l/fuzz_stdout.txt
