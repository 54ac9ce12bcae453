O=gpurun_out/s4; mkdir -p $O; export TMPDIR=/tmp
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attention or attn" > $O/attn_tests.log 2>&1; tail -2 $O/attn_tests.log
timeout 300 python tools/attn_ab_libs.py > $O/attn_idle_skip_ab.txt 2>&1; cat $O/attn_idle_skip_ab.txt
