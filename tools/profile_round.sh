#!/bin/bash
# Regenerate the per-round profile set on an MI355X box (run through gpurun from the repo root):
#   gpurun --timeout 1200 -- 'bash tools/profile_round.sh r02'
# then, back in the build container:
#   python tools/profile_summary.py gpurun_out/<tag> <tag> && python tools/sq_summary.py gpurun_out/<tag> <tag>
#   cp gpurun_out/<tag>/phase_cycles.txt profiles/<tag>_phase_cycles.txt
#   cp gpurun_out/<tag>/host_rates.txt   profiles/<tag>_host_api_rate.txt
# Counters are collected in their own passes (never together with the hip/hsa trace domains), one
# rocprofv3 invocation per counter group, the program itself after "--".
set -o pipefail
tag=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag
mkdir -p "$out"
python3 "$R/bench.py" > "$out/bench_n1.json" 2> "$out/bench_n1.err" || exit 1
python3 "$R/tools/phase_cycles.py" 8 v > "$out/phase_cycles.txt" 2>&1
python3 "$R/tools/host_rates.py" > "$out/host_rates.txt" 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 "$R/bench.py" --no-cpu-baseline > "$out/trace.log" 2>&1 || exit 2
for c in FETCH_SIZE WRITE_SIZE; do
  d=$(echo $c | tr A-Z a-z | sed 's/_size//')
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/pmc_$d" -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$out/pmc_$d.log" 2>&1 || exit 3
done
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES \
  --output-format csv -d "$out/pmc_sq1" -- python3 "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline > "$out/pmc_sq1.log" 2>&1 || exit 4
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS \
  --output-format csv -d "$out/pmc_sq2" -- python3 "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline > "$out/pmc_sq2.log" 2>&1 || exit 5
echo "profile set written under gpurun_out/$tag"
