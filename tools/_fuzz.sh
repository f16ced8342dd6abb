set -e
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout -k 10 500 python tools/fuzz_long.py 140 411 > gpurun_out/fuzz_long_411.txt 2>&1; tail -2 gpurun_out/fuzz_long_411.txt ) &
P1=$!
( timeout -k 10 500 python tools/fuzz_parity.py 400 412 > gpurun_out/fuzz_parity_412.txt 2>&1; tail -2 gpurun_out/fuzz_parity_412.txt ) &
P2=$!
wait $P1; wait $P2
bash tools/profile_workload.sh r02_batch64 --no-configs > gpurun_out/prof_batch64.log 2>&1
echo "batch64 profiled"
bash tools/profile_workload.sh r02_n16384 --workload n16384 > gpurun_out/prof_n16384.log 2>&1
echo "n16384 profiled"
