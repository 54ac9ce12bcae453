O=gpurun_out/s4; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_final3.json 2> $O/bench_final3.err; tail -c 200 $O/bench_final3.json
timeout 600 python bench.py --model small.en --no-cpu-baseline --no-reference-loop > $O/bench_small_en3.json 2> $O/bench_small_en3.err
timeout 600 python bench.py --mode recipe --no-cpu-baseline > $O/bench_recipe3.json 2> $O/bench_recipe3.err
timeout 300 python bench.py --model tiny.en --batch 2 --no-cpu-baseline --no-reference-loop > $O/bench_tiny3.json 2> $O/bench_tiny3.err
