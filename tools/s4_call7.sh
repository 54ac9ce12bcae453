O=gpurun_out/s4; mkdir -p $O; export TMPDIR=/tmp
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --share-device --backend gloo --dp-probe --steps 3 --warmup 1 --no-cpu-baseline --no-reference-loop --no-ab > $O/bench_dp2.json 2> $O/bench_dp2.log; tail -c 1500 $O/bench_dp2.json; grep -i "replica\|mode selection\|error" $O/bench_dp2.log | tail -5
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_final.json 2> $O/bench_final.err; tail -c 600 $O/bench_final.json
