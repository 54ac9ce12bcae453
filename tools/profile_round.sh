# One GPU call that refreshes the judged measurements: vendor GEMM / attention calibration, two PMC passes (FETCH_SIZE /
# WRITE_SIZE) -> pmc_traffic.json, the default bench line (which then quotes the traffic and the ceilings taken in THIS call),
# rocprofv3 kernel stats of the bench command.  Outputs under gpurun_out/prof/; copy what is to be judged into profiles/.
# Every stage is bounded by its own `timeout`: a rocprofv3 --pmc pass that faults does not return by itself (one such
# pass once held the box for 24 minutes).  The whole script is ~5 minutes of box time.
O=gpurun_out/prof; mkdir -p $O; export TMPDIR=/tmp
timeout 120 python tools/hipblaslt_probe.py $O/vendor_gemm_ceiling.json > $O/hipblaslt.log 2>&1
timeout 120 python tools/attn_vendor_probe.py $O/attn_vendor_ceiling.json > $O/attn_vendor.log 2>&1
timeout 90 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_f -o b -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-reference-loop --graph off --no-ab > $O/pmc_f.log 2>&1
timeout 90 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_w -o b -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-reference-loop --graph off --no-ab > $O/pmc_w.log 2>&1
python tools/pmc_traffic.py $O/pmc_f/b_results.db $O/pmc_w/b_results.db $O/pmc_traffic.json > $O/pmc_traffic.log 2>&1
rm -rf $O/pmc_f $O/pmc_w
# the bench below quotes these (same box, same kernel sources)
for f in pmc_traffic.json vendor_gemm_ceiling.json attn_vendor_ceiling.json; do [ -s $O/$f ] && cp $O/$f profiles/$f; done
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err     # the driver's command
timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-reference-loop --no-overlap --graph off --no-ab > $O/bench_prof.log 2>&1
python tools/rocprof_summary.py $O/prof/bench_results.db $O/kernel_stats.md > /dev/null 2>&1
rm -rf $O/prof
cat $O/hipblaslt.log; tail -3 $O/pmc_traffic.log; tail -c 600 $O/bench.json
