set -e
export TMPDIR=/tmp
R=$(pwd)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1 || { tail -40 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
bash tools/profile_workload.sh r02_batch64 --no-configs > gpurun_out/prof_batch64.log 2>&1
echo "batch64 profiled"
bash tools/profile_workload.sh r02_n16384 --workload n16384 > gpurun_out/prof_n16384.log 2>&1
echo "n16384 profiled"
cd $R
timeout -k 10 200 python tools/phase_cycles.py 64 waves > gpurun_out/r02_batch64_phase_cycles.txt 2>&1
timeout -k 10 200 python tools/phase_cycles_n16384.py > gpurun_out/r02_n16384_phase_cycles.txt 2>&1 || true
timeout -k 10 200 python tools/kernel_clock.py > gpurun_out/r02_kernel_clock.txt 2>&1 || true
