# session-4 call 1: GPU suite at the optimizer change, merged-segment A/B of the reference-loop leg, long-form batch sweep
O=gpurun_out/s4; mkdir -p $O; export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" >> $O/pytest_gpu.log 2>&1; tail -1 $O/pytest_gpu.log
for m in 1 0 1 0; do MERGE=$m LEG=full N=6 timeout 200 python tools/ref_loop_profile.py 2>&1 | grep LEG | sed "s/^/MERGE=$m /" >> $O/merge_ab.txt; done
cat $O/merge_ab.txt
for b in 16 32 64; do B=$b CLIPS=$((b/4)) timeout 300 python tools/bench_longform.py > $O/longform_b$b.json 2> $O/longform_b$b.err; python - $O/longform_b$b.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print(d["batch"], d["eager"], d.get("graphs"))
PY
done
