set -e
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
timeout -k 10 300 python tools/ab_kernel.py ab/lib_cur.so em-spec_amd/libemspec.so --workload n16384 --rounds 2 2>&1 | tail -2
timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('headline', d['value'], d['ms_per_step']); print('dump', d['roofline_parity_dump']['frac'], d['roofline_parity_dump']['columns_per_s'])
for k,v in d['configs'].items(): print(k, '%.3e' % v['columns_per_s'])"
