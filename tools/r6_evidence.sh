# Round-6 evidence at the final kernel sources, one GPU call (every stage bounded by its own timeout).  Outputs under
# gpurun_out/r6/; the files to be judged are copied into profiles/ by hand afterwards.
O=gpurun_out/r6; mkdir -p $O; export TMPDIR=/tmp
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" >> $O/pytest_gpu.log 2>&1; tail -1 $O/pytest_gpu.log
timeout 150 python tools/hipblaslt_probe.py $O/vendor_gemm_ceiling.json > $O/hipblaslt.log 2>&1
timeout 150 python tools/attn_vendor_probe.py $O/attn_vendor_ceiling.json > $O/attn_vendor.log 2>&1
timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_f -o b -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-reference-loop --graph off --no-ab > $O/pmc_f.log 2>&1
timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_w -o b -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-reference-loop --graph off --no-ab > $O/pmc_w.log 2>&1
python tools/pmc_traffic.py $O/pmc_f/b_results.db $O/pmc_w/b_results.db $O/pmc_traffic.json > $O/pmc_traffic.log 2>&1
rm -rf $O/pmc_f $O/pmc_w
for f in pmc_traffic.json vendor_gemm_ceiling.json attn_vendor_ceiling.json; do [ -s $O/$f ] && cp $O/$f profiles/$f; done
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err      # the driver's command
timeout 240 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-reference-loop --no-overlap --graph off --no-ab > $O/bench_prof.log 2>&1
python tools/rocprof_summary.py $O/prof/bench_results.db $O/bench_kernel_stats.md > /dev/null 2>&1
rm -rf $O/prof
# GEMM PMC counters, one row per step shape (two counter groups, separate passes)
ARGS=""
i=0
for sh in "fwd qkv" "fwd fc1" "fwd fc2" "fwd out" "dX  fc2" "dX  fc1" "dX  qkv" "dW  fc1"; do
  i=$((i+1))
  SHAPES="$sh" timeout 90 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY --kernel-trace -d $O/g1_$i -o g -- python tools/gemm_step_shapes.py > $O/g1_$i.log 2>&1
  SHAPES="$sh" timeout 90 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $O/g2_$i -o g -- python tools/gemm_step_shapes.py > $O/g2_$i.log 2>&1
  A=$(find $O/g1_$i -name "*_results.db" | head -1); B=$(find $O/g2_$i -name "*_results.db" | head -1)
  ARGS="$ARGS|$sh=$A,$B"
done
python - "$ARGS" > $O/gemm_pmc_table.md 2> $O/gemm_pmc.err <<'PY'
import subprocess, sys
args = [a for a in sys.argv[1].split("|") if a]
sys.stdout.write(subprocess.run([sys.executable, "tools/gemm_pmc.py"] + args, capture_output=True, text=True).stdout)
PY
rm -rf $O/g1_* $O/g2_*
# other configs / modes
timeout 600 python bench.py --model small.en --no-cpu-baseline --no-reference-loop > $O/bench_small_en.json 2> $O/bench_small_en.err
MODEL=small.en timeout 300 python tools/step_breakdown.py > $O/small_en_step_breakdown.md 2> $O/small_en_step_breakdown.err
timeout 600 python bench.py --mode recipe --no-cpu-baseline > $O/bench_recipe.json 2> $O/bench_recipe.err
timeout 300 python bench.py --model tiny.en --batch 2 --no-cpu-baseline --no-reference-loop > $O/bench_tiny_en_b2.json 2> $O/bench_tiny.err
timeout 300 python tools/bench_longform.py > $O/longform_bench.json 2> $O/longform.err
timeout 300 python tools/gemm_m128_probe.py > $O/gemm_small_m_probe.txt 2>&1
timeout 300 python tools/bench_pseudo_label.py > $O/pseudo_label_bench.json 2> $O/pseudo_label.err
DW_LENS=1 timeout 300 python tools/step_breakdown.py > $O/step_breakdown.md 2> $O/step_breakdown.err
timeout 300 bash tools/decode_profile_step.sh > $O/decode_step_sequence.txt 2>&1
timeout 200 python tools/attn_pipe_ab.py > $O/attn_pipe_ab.txt 2>&1
DW_STREAMS=1 DW_ROUNDS=4 DW_AB="[{}, {25: 0}, {6: 8}, {25: 0, 6: 8}]" timeout 400 python tools/ab_keys.py > $O/ab_round6_kernel_changes.txt 2>&1
tail -c 400 $O/bench_full.json; echo; tail -2 $O/pmc_traffic.log; head -4 $O/gemm_pmc_table.md | cut -c1-250
