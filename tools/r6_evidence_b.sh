O=gpurun_out/r6b; mkdir -p $O; export TMPDIR=/tmp
python -m pytest tests/test_sharp_parity_gpu.py -q -s > $O/sharp.log 2>&1; tail -3 $O/sharp.log
# GEMM PMC passes (two counter groups, separate runs)
timeout 240 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY --kernel-trace -d $O/pmc1 -o g -- python tools/gemm_step_shapes.py > $O/pmc1.log 2>&1
timeout 240 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace -d $O/pmc2 -o g -- python tools/gemm_step_shapes.py > $O/pmc2.log 2>&1
python tools/gemm_pmc.py $(find $O/pmc1 $O/pmc2 -name "*_results.db") > $O/gemm_pmc_table.md 2> $O/gemm_pmc.err; head -5 $O/gemm_pmc_table.md | cut -c1-300
timeout 200 rocprofv3 --kernel-trace --stats -d $O/gs -o g -- python tools/gemm_step_shapes.py > $O/gs.log 2>&1
python tools/rocprof_summary.py $(find $O/gs -name "*_results.db" | head -1) $O/gemm_shapes_kernel_stats.md > /dev/null 2>&1
rm -rf $O/pmc1 $O/pmc2 $O/gs
# two ranks on one GPU over gloo: the data-parallel mode selection code path
timeout 600 python bench.py --gpus 2 --share-device --backend gloo --dp-probe --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/bench_dp2_probe.json 2> $O/bench_dp2_probe.err; tail -c 400 $O/bench_dp2_probe.json; grep "mode selection" $O/bench_dp2_probe.err
