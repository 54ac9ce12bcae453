O=gpurun_out/s4; mkdir -p $O; export TMPDIR=/tmp
DW_STREAMS=1 DW_ROUNDS=5 DW_AB='[{}, {"varlen": 0}, {"overwrite": 0}, {"varlen": 0, "overwrite": 0}]' timeout 500 python tools/ab_keys.py > $O/ab_varlen_overwrite.txt 2>&1; tail -5 $O/ab_varlen_overwrite.txt
DW_STREAMS=0 DW_ROUNDS=4 DW_AB='[{}, {"varlen": 0}, {"overwrite": 0}]' timeout 400 python tools/ab_keys.py > $O/ab_varlen_overwrite_single.txt 2>&1; tail -4 $O/ab_varlen_overwrite_single.txt
