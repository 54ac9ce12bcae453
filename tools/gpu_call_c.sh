mkdir -p gpurun_out/r2o; export TMPDIR=/tmp
timeout 60 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/r2o/pmc_f -o b -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r2o/pmc_f.log 2>&1; echo "pmc_f rc=$?"
timeout 60 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/r2o/pmc_w -o b -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r2o/pmc_w.log 2>&1; echo "pmc_w rc=$?"
python tools/pmc_traffic.py gpurun_out/r2o/pmc_f/b_results.db gpurun_out/r2o/pmc_w/b_results.db gpurun_out/r2o/pmc_traffic.json > gpurun_out/r2o/pmc_traffic.log 2>&1
rm -rf gpurun_out/r2o/pmc_f gpurun_out/r2o/pmc_w
grep -i "fault\|ms/step" gpurun_out/r2o/pmc_f.log gpurun_out/r2o/pmc_w.log | head; tail -2 gpurun_out/r2o/pmc_traffic.log
timeout 150 python -m pytest tests -m gpu -x -q > gpurun_out/r2o/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2o/pytest.log
grep -E "passed|failed|rc=" gpurun_out/r2o/pytest.log | tail -3
