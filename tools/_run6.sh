set -e
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | tail -2
timeout -k 10 300 python tools/ab_kernel.py ab/lib_cur.so em-spec_amd/libemspec.so --n 1024 --hop 256 --rounds 2 2>&1 | tail -2
timeout -k 10 300 python tools/ab_kernel.py ab/lib_cur.so em-spec_amd/libemspec.so --n 8192 --hop 512 --rounds 2 2>&1 | tail -2
timeout -k 10 300 python tools/ab_kernel.py ab/lib_cur.so em-spec_amd/libemspec.so --n 2048 --hop 128 --rounds 2 2>&1 | tail -2
timeout -k 10 300 python tools/ab_kernel.py ab/lib_cur.so em-spec_amd/libemspec.so --n 4096 --hop 300 --rounds 2 2>&1 | tail -2
