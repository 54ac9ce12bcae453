mkdir -p gpurun_out/r2l; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2l/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2l/pytest.log
timeout 900 python bench.py > gpurun_out/r2l/bench.json 2> gpurun_out/r2l/bench.err
grep -E "passed|failed|rc=" gpurun_out/r2l/pytest.log | tail -3; grep "ms/step\|per-kernel" gpurun_out/r2l/bench.err | cut -c1-1500; tail -c 900 gpurun_out/r2l/bench.json
