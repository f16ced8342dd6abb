#!/bin/bash
# Profile one bench.py workload on an MI355X box (run through gpurun from the repo root):
#   gpurun --timeout 900 -- 'bash tools/profile_workload.sh r02_n16384 --workload n16384'
# Writes gpurun_out/<tag>/{trace,pmc_fetch,pmc_write,pmc_sq1,pmc_sq2}; summarise with tools/profile_summary.py /
# tools/sq_summary.py.  Counters are collected in their own passes (never with the hip/hsa trace domains), the program
# itself after "--".
set -o pipefail
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out/$tag
mkdir -p "$out"
python3 "$R/tools/sources_sha.py" > "$out/sources_sha.txt" || exit 1
# the library that will run must have been built from exactly these sources (emspec_build_info)
python3 -c "import sys; sys.path[:0] = ['$R', '$R/em-spec_amd']; import emspec; i = emspec.build_info(); print(i); sys.exit(0 if ('sources=' + open('$out/sources_sha.txt').read().strip()) in i else 1)" > "$out/build_info.txt" || { echo "libemspec.so is not built from the sources in the tree: rebuild before profiling"; exit 1; }
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 "$R/bench.py" --no-cpu-baseline --steps 9 --warmup 2 "$@" > "$out/trace.log" 2>&1 || exit 2
for c in FETCH_SIZE WRITE_SIZE; do
  d=$(echo $c | tr A-Z a-z | sed 's/_size//')
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$out/pmc_$d" -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline "$@" > "$out/pmc_$d.log" 2>&1 || exit 3
done
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES \
  --output-format csv -d "$out/pmc_sq1" -- python3 "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline "$@" > "$out/pmc_sq1.log" 2>&1 || exit 4
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS \
  --output-format csv -d "$out/pmc_sq2" -- python3 "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline "$@" > "$out/pmc_sq2.log" 2>&1 || exit 5
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d "$out/pmc_clk" -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline "$@" > "$out/pmc_clk.log" 2>&1 || exit 6
# keep only this library's kernels in the per-dispatch CSVs (the synthetic-input generator launches thousands of
# torch kernels; unfiltered the set exceeds what gpurun copies back)
for f in "$out"/*/*/*_counter_collection.csv "$out"/*/*/*_kernel_trace.csv; do
  [ -f "$f" ] && { head -1 "$f"; grep emspec "$f"; } > "$f.tmp" && mv "$f.tmp" "$f"
done
rm -f "$out"/*/*/*_agent_info.csv
echo "profile set written under gpurun_out/$tag"
