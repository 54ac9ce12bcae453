mkdir -p gpurun_out/r2j; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2j/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2j/pytest.log
DW_AB="[(115,0,5,8,1,1,1,0,0),(115,0,5,8,1,1,1,0,1),(115,0,5,8,1,1,1,1,0),(115,0,5,8,1,1,1,1,1)]" timeout 600 python tools/ab_step.py > gpurun_out/r2j/ab_overlap.log 2>&1
timeout 900 python bench.py > gpurun_out/r2j/bench.json 2> gpurun_out/r2j/bench.err
tail -5 gpurun_out/r2j/pytest.log; cat gpurun_out/r2j/ab_overlap.log | tail -6; tail -c 1500 gpurun_out/r2j/bench.json
