set -e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python bench.py > gpurun_out/r02_bench_n1.json 2> gpurun_out/bench_n1.err
python3 -c "
import json; d=json.load(open('gpurun_out/r02_bench_n1.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
print('rc', {k: d['roofline_compute'].get(k) for k in ('flop_frac','valu_util','wait_any_share','clock_ghz')})
for k,v in d['configs'].items(): print(k, '%.3e' % v['columns_per_s'], 'ms', round(v['kernel_ms'],3), 'frac', round(v['roofline']['frac'],4), v['roofline'].get('traffic'))
print('dump', d['roofline_parity_dump']['frac'], d['roofline_parity_dump']['columns_per_s'])
"
timeout -k 10 300 python bench.py --gather loopback --no-cpu-baseline --no-configs > gpurun_out/r02_bench_loopback.json 2> gpurun_out/bench_loop.err
python3 -c "
import json; d=json.load(open('gpurun_out/r02_bench_loopback.json')); print('loopback', d['value'], d['ms_per_step'], d['gather'])"
timeout -k 10 300 python tools/host_rates.py > gpurun_out/r02_host_api_rate.txt 2>&1 || true
tail -12 gpurun_out/r02_host_api_rate.txt
