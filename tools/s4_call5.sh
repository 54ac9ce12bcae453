O=gpurun_out/s4; mkdir -p $O; export TMPDIR=/tmp
DW_ATTN_ALL_LIBS=1 timeout 400 python tools/attn_ab_libs.py > $O/attn_compiler_flags_ab.txt 2>&1; grep -v "^/opt" $O/attn_compiler_flags_ab.txt
DW_STREAMS=1 DW_ROUNDS=3 DW_AB='[{}, {"lib": 1}, {"lib": 2}, {"lib": 4}, {"lib": 5}]' timeout 600 python tools/ab_keys.py > $O/ab_compiler_flags_step.txt 2>&1; tail -5 $O/ab_compiler_flags_step.txt
