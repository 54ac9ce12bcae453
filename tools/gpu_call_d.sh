mkdir -p gpurun_out/r2p; export TMPDIR=/tmp
timeout 140 python -m pytest tests -m gpu -x -q > gpurun_out/r2p/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2p/pytest.log
grep -E "passed|failed|rc=|Error" gpurun_out/r2p/pytest.log | tail -4
DW_AB="[(2163,0,5,8,1,1,1,1,1,0),(2163,0,5,8,1,1,1,1,1,1)]" timeout 70 python tools/ab_step.py > gpurun_out/r2p/ab_pad.log 2>&1; tail -3 gpurun_out/r2p/ab_pad.log
