set -e
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -m gpu -x -q -k "16384 or low_end or golden" 2>&1 | tail -2
timeout -k 10 300 python tools/ab_kernel.py ab/lib_cur.so em-spec_amd/libemspec.so --workload n16384 --rounds 2 2>&1 | tail -4
