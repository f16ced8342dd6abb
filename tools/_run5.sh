set -e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1 || { tail -40 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
timeout -k 10 400 python tools/ab_kernel.py ab/lib_noasm.so em-spec_amd/libemspec.so --workload n16384 --rounds 2 2>&1 | tail -2
timeout -k 10 400 python tools/ab_kernel.py ab/lib_noasm.so em-spec_amd/libemspec.so --n 1024 --hop 256 --rounds 2 2>&1 | tail -2
timeout -k 10 400 python tools/ab_kernel.py ab/lib_noasm.so em-spec_amd/libemspec.so --n 8192 --hop 512 --rounds 2 2>&1 | tail -2
