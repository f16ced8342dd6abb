set -e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1 || { tail -40 gpurun_out/gpu_tests.log; exit 1; }
tail -3 gpurun_out/gpu_tests.log
for i in 1 2; do
timeout -k 10 300 python bench.py --no-cpu-baseline --no-configs > gpurun_out/bench_q$i.json 2> gpurun_out/bench_q$i.err
python3 -c "import json; d=json.load(open('gpurun_out/bench_q$i.json')); print('bench', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done
