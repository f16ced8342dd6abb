set -e
export TMPDIR=/tmp
mkdir -p gpurun_out
EMSPEC_PP=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -m gpu -x -q -k "fused or segment or golden or batch" > gpurun_out/pp_tests.log 2>&1 || { tail -40 gpurun_out/pp_tests.log; exit 1; }
tail -3 gpurun_out/pp_tests.log
timeout -k 10 300 python bench.py --no-cpu-baseline --no-configs > gpurun_out/bench_r8.json 2> gpurun_out/bench_r8.err
python3 -c "import json; d=json.load(open('gpurun_out/bench_r8.json')); print('r8', d['value'], d['ms_per_step'])"
EMSPEC_PP=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-configs > gpurun_out/bench_pp.json 2> gpurun_out/bench_pp.err
python3 -c "import json; d=json.load(open('gpurun_out/bench_pp.json')); print('pp', d['value'], d['ms_per_step'])"
