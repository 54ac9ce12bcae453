O=gpurun_out/s4; mkdir -p $O; export TMPDIR=/tmp
python -m pytest tests/test_training_parity_gpu.py tests/test_engine_gpu.py tests/test_kernels_gpu_production_shapes.py -m gpu -q -x > $O/parity_tests3.log 2>&1; tail -3 $O/parity_tests3.log
DW_STREAMS=1 DW_ROUNDS=6 DW_AB='[{}, {"overwrite": 0}]' timeout 500 python tools/ab_keys.py > $O/ab_overwrite2.txt 2>&1; tail -3 $O/ab_overwrite2.txt
DW_LENS=1 timeout 300 python tools/step_breakdown.py 2>/dev/null | grep -E "m51866|m4160 n1280 k519|reduce|adamw|instrumented" 
