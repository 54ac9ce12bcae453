O=gpurun_out/s4; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_final2.json 2> $O/bench_final2.err; tail -c 300 $O/bench_final2.json
timeout 600 python bench.py --model small.en --no-cpu-baseline --no-reference-loop > $O/bench_small_en2.json 2> $O/bench_small_en2.err
timeout 600 python bench.py --mode recipe --no-cpu-baseline > $O/bench_recipe2.json 2> $O/bench_recipe2.err
python -m pytest tests -m gpu -q > $O/pytest_gpu9.log 2>&1; tail -2 $O/pytest_gpu9.log
python -c "import __graft_entry__ as g; g.smoke()" >> $O/pytest_gpu9.log 2>&1; tail -1 $O/pytest_gpu9.log
