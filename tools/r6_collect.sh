# Copy the judged files of a tools/r6_evidence.sh run (gpurun_out/r6) and of a second contract bench sample (gpurun_out/r6b, optional)
# into profiles/ under their round-6 names.
set -e
S=gpurun_out/r6; P=profiles
cls() { grep "per-kernel-class" "$1" | sed 's/^.*per-kernel-class ms (instrumented step): //' ; }
cp $S/bench_kernel_stats.md $P/r6_bench_kernel_stats.md
cp $S/pmc_traffic.json $P/pmc_traffic.json; cp $S/pmc_traffic.json $P/r6_pmc_traffic.json
cp $S/vendor_gemm_ceiling.json $P/vendor_gemm_ceiling.json; cp $S/attn_vendor_ceiling.json $P/attn_vendor_ceiling.json
cp $S/hipblaslt.log $P/r6_vendor_gemm_probe.txt
cp $S/gemm_pmc_table.md $P/r6_gemm_pmc_table.md
cp $S/bench_small_en.json $P/r6_bench_small_en.json; cp $S/small_en_step_breakdown.md $P/r6_small_en_step_breakdown.md
cp $S/bench_recipe.json $P/r6_bench_large_v3_recipe.json; cp $S/bench_tiny_en_b2.json $P/r6_bench_tiny_en_b2.json
cp $S/longform_bench.json $P/r6_longform_bench.json; cp $S/pseudo_label_bench.json $P/r6_pseudo_label_bench.json
cp $S/gemm_small_m_probe.txt $P/r6_gemm_small_m_probe.txt; cp $S/step_breakdown.md $P/r6_step_breakdown.md
cp $S/decode_step_sequence.txt $P/r6_decode_step_sequence.txt; cp $S/attn_pipe_ab.txt $P/r6_attn_fwd_pipe_ab.txt
cp $S/ab_round6_kernel_changes.txt $P/r6_ab_kernel_changes.txt; cp $S/pytest_gpu.log $P/r6_pytest_gpu.log
if [ -s gpurun_out/r6b/bench_full.json ]; then
  cp gpurun_out/r6b/bench_full.json $P/r6_bench_full.json; cls gpurun_out/r6b/bench_full.err > $P/r6_bench_full_kernel_classes.txt
  cp $S/bench_full.json $P/r6_bench_full_evidence_call.json; cls $S/bench_full.err > $P/r6_bench_full_evidence_call_kernel_classes.txt
else
  cp $S/bench_full.json $P/r6_bench_full.json; cls $S/bench_full.err > $P/r6_bench_full_kernel_classes.txt
fi
ls -la $P | grep r6_ | wc -l
