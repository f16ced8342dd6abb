set -e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -3 gpurun_out/gpu_tests.log
timeout -k 10 300 python bench.py --no-cpu-baseline --no-configs > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err
cat gpurun_out/bench_quick.json | cut -c1-400
timeout -k 10 300 python bench.py --gather loopback --no-cpu-baseline --no-configs > gpurun_out/bench_loop.json 2> gpurun_out/bench_loop.err
cat gpurun_out/bench_loop.json | cut -c1-300
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --backend gloo --log2-samples 20 --steps 3 --warmup 1 > gpurun_out/bench_gloo2.json 2> gpurun_out/bench_gloo2.err || { tail -30 gpurun_out/bench_gloo2.err; exit 1; }
cat gpurun_out/bench_gloo2.json
