O=gpurun_out/s4; mkdir -p $O; export TMPDIR=/tmp
python -m pytest tests/test_training_parity_gpu.py tests/test_engine_gpu.py tests/test_sharp_parity_gpu.py tests/test_reference_loop.py tests/test_boundary.py -m gpu -q -x > $O/parity_tests8.log 2>&1; tail -3 $O/parity_tests8.log
DW_STREAMS=1 DW_ROUNDS=6 DW_AB='[{}, {"pack": 0}]' timeout 500 python tools/ab_keys.py > $O/ab_pack.txt 2>&1; tail -2 $O/ab_pack.txt
timeout 600 python bench.py --mode recipe --no-cpu-baseline --no-reference-loop > $O/bench_recipe8.json 2> $O/bench_recipe8.err; python -c "
import json; d=json.load(open('$O/bench_recipe8.json')); print('recipe', d['ms_per_step'], d['value'], d['step_mfma_frac'], d['step_mode'])"
