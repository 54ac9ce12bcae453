# One GPU call that refreshes the judged measurements: vendor GEMM calibration, default bench line, rocprofv3 kernel stats of
# the bench command, two PMC passes (FETCH_SIZE / WRITE_SIZE) -> pmc_traffic.json.  Outputs under gpurun_out/r2m/.
mkdir -p gpurun_out/r2m; export TMPDIR=/tmp
python tools/hipblaslt_probe.py > gpurun_out/r2m/hipblaslt.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/r2m/probe_prof -o p -- python tools/hipblaslt_probe.py > /dev/null 2>&1
python tools/rocprof_summary.py gpurun_out/r2m/probe_prof/p_results.db gpurun_out/r2m/probe_kernels.md > /dev/null 2>&1
rm -rf gpurun_out/r2m/probe_prof
python bench.py > gpurun_out/r2m/bench.json 2> gpurun_out/r2m/bench.err
rocprofv3 --kernel-trace --stats -d gpurun_out/r2m/prof -o bench -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-overlap > gpurun_out/r2m/bench_prof.log 2>&1
python tools/rocprof_summary.py gpurun_out/r2m/prof/bench_results.db gpurun_out/r2m/kernel_stats.md > /dev/null 2>&1
rm -rf gpurun_out/r2m/prof
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/r2m/pmc_f -o b -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r2m/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/r2m/pmc_w -o b -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r2m/pmc_w.log 2>&1
python tools/pmc_traffic.py gpurun_out/r2m/pmc_f/b_results.db gpurun_out/r2m/pmc_w/b_results.db gpurun_out/r2m/pmc_traffic.json > gpurun_out/r2m/pmc_traffic.log 2>&1
ls -la gpurun_out/r2m/pmc_f gpurun_out/r2m/pmc_w >> gpurun_out/r2m/pmc_traffic.log 2>&1
rm -rf gpurun_out/r2m/pmc_f gpurun_out/r2m/pmc_w
cat gpurun_out/r2m/hipblaslt.log; tail -3 gpurun_out/r2m/pmc_traffic.log; tail -c 600 gpurun_out/r2m/bench.json
