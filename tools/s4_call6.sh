O=gpurun_out/s4; mkdir -p $O; export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x > $O/pytest_gpu6.log 2>&1; tail -3 $O/pytest_gpu6.log
DW_STREAMS=1 DW_ROUNDS=6 DW_AB='[{}, {"overwrite": 0}]' timeout 500 python tools/ab_keys.py > $O/ab_overwrite3.txt 2>&1; tail -2 $O/ab_overwrite3.txt
for l in full verbatim; do LEG=$l N=6 timeout 200 python tools/ref_loop_profile.py 2>&1 | grep LEG; done
