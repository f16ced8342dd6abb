#!/bin/bash
# Everything a round's numbers come from, on one MI355X box (run from the repo root through gpurun):
#   gpurun --timeout 1200 -- 'bash tools/round_check.sh r05 ab'    (after a kernel change: suite + profiles; ~3.5 GPU-minutes)
#   HERE: the five profile_json.py lines below, commit profiles/<tag>_*.json
#   gpurun --timeout 1200 -- 'bash tools/round_check.sh r05 abc'   (profiles AND bench lines on ONE box; summarise again, copy, commit)
# Parts: `a` = suite + the float32 workloads, `b` = the EXACT workloads + the stamped builds and rate tools, `c` = the four
# bench lines.  A line quotes the PMC counters only while profiles/<tag>_*.json carry the library's sources sha - hence the
# first call - and tests/test_bench_helpers.py holds every profile's median duration to <= 1.02 x the line's kernel time, which
# between boxes does not hold (their clocks differ by more) - hence `abc` in one call.
# 1. the GPU test suite; 2. rocprofv3 profiles of the two bench workloads (tools/profile_workload.sh: kernel trace +
# separate PMC passes); 3. stamped-build phase cycles and the in-kernel clock; 4. gather cost; 5. host-API rates.
# Afterwards, HERE:  python tools/profile_json.py gpurun_out/<tag>_batch64 <tag> batch64 1047616
#                    python tools/profile_json.py gpurun_out/<tag>_n16384 <tag> n16384 522304
#                    python tools/profile_json.py gpurun_out/<tag>_paritydump <tag> paritydump 65296 --last 20
#                    python tools/profile_json.py gpurun_out/<tag>_exact64 <tag> exact64 1047616
#                    python tools/profile_json.py gpurun_out/<tag>_exact_n16384 <tag> exact_n16384 522304 --all-kernels
#                    python tools/profile_json.py gpurun_out/<tag>_exact_n1024 <tag> exact_n1024 1048384
#                    cp gpurun_out/<tag>_*.txt profiles/   and commit;
set -e
tag=${1:-r02}
part=${2:-ab}
export TMPDIR=/tmp
R=$(pwd)
mkdir -p gpurun_out
if [[ $part == *a* ]]; then
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1 || { tail -40 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
bash tools/profile_workload.sh ${tag}_batch64 --no-configs > gpurun_out/prof_batch64.log 2>&1
echo "batch64 profiled"
bash tools/profile_workload.sh ${tag}_n16384 --workload n16384 > gpurun_out/prof_n16384.log 2>&1
echo "n16384 profiled"
bash tools/profile_workload.sh ${tag}_paritydump --workload paritydump --steps 20 > gpurun_out/prof_paritydump.log 2>&1
echo "paritydump profiled"
fi
if [[ $part == *b* ]]; then
bash tools/profile_workload.sh ${tag}_exact64 --mode exact --no-configs > gpurun_out/prof_exact64.log 2>&1
echo "exact64 profiled"
bash tools/profile_workload.sh ${tag}_exact_n16384 --mode exact --workload n16384 > gpurun_out/prof_exact_n16384.log 2>&1
echo "exact_n16384 profiled"
bash tools/profile_workload.sh ${tag}_exact_n1024 --mode exact --workload n1024 > gpurun_out/prof_exact_n1024.log 2>&1
echo "exact_n1024 profiled"
cd "$R"
timeout -k 10 200 python tools/phase_cycles.py 64 waves > gpurun_out/${tag}_batch64_phase_cycles.txt 2>&1
timeout -k 10 200 python tools/phase_cycles_n16384.py > gpurun_out/${tag}_n16384_phase_cycles.txt 2>&1 || true
timeout -k 10 200 python tools/phase_cycles_exact.py 64 waves > gpurun_out/${tag}_exact64_phase_cycles.txt 2>&1 || true
timeout -k 10 200 python tools/kernel_clock.py > gpurun_out/${tag}_kernel_clock.txt 2>&1 || true
timeout -k 10 200 python tools/gather_cost.py > gpurun_out/${tag}_gather_cost.txt 2>&1 || true
timeout -k 10 300 python tools/host_rates.py > gpurun_out/${tag}_host_api_rate.txt 2>&1 || true
timeout -k 10 300 python tools/host_pipeline_rates.py > gpurun_out/${tag}_host_pipeline_rate.txt 2>&1 || true
(timeout -k 10 200 python tools/host_pageable_rate.py; timeout -k 10 200 python tools/host_pageable_rate.py --exact) > gpurun_out/${tag}_host_pageable_rate.txt 2>&1 || true
(timeout -k 10 200 python tools/host_few_streams_rate.py; timeout -k 10 200 python tools/host_few_streams_rate.py --exact) > gpurun_out/${tag}_host_few_streams_rate.txt 2>&1 || true
(timeout -k 10 200 python tools/host_small_batch_rate.py; timeout -k 10 200 python tools/host_small_batch_rate.py --exact) > gpurun_out/${tag}_host_small_batch_rate.txt 2>&1 || true
# round 6: the live multi-stream entries (per-call host time of every variant; the frame kernel's phases), EXACT at the small sizes
timeout -k 10 300 python tools/live_rate.py 64 400 > gpurun_out/${tag}_live_rate.txt 2>&1 || true
timeout -k 10 200 python tools/live_phases.py 64 200 > gpurun_out/${tag}_live_phases.txt 2>&1 || true
timeout -k 10 300 python tools/exact_small_rate.py > gpurun_out/${tag}_exact_small_rate.txt 2>&1 || true
(cd /tmp && /opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -o /tmp/launch_sync "$R/tools/ubench/launch_sync.hip" > /dev/null 2>&1 && timeout -k 10 60 /tmp/launch_sync > "$R/gpurun_out/${tag}_launch_sync.txt" 2>&1) || true
fi
if [[ $part == *c* ]]; then
# the bench lines quote a profile's counters only when profiles/<tag>_<workload>.json carries the library's sources sha: condense
# this call's profiles HERE, on the box, before the lines are taken (the same command is repeated on the merged gpurun_out/
# afterwards - the result is identical - and THAT copy is committed); one call then does for the whole set
cd "$R"
for spec in "batch64 batch64 1047616" "n16384 n16384 522304" "paritydump paritydump 65296 --last 20" "exact64 exact64 1047616" \
            "exact_n16384 exact_n16384 522304 --all-kernels" "exact_n1024 exact_n1024 1048384"; do
  set -- $spec
  d=$1; wl=$2; cols=$3; shift 3
  [ -d gpurun_out/${tag}_$d ] && python tools/profile_json.py gpurun_out/${tag}_$d $tag $wl $cols "$@" > /dev/null 2>> gpurun_out/profile_json.err
done
# the bench lines, on the SAME box as the profiles they quote (tests/test_bench_helpers.py holds every profile's median
# duration to <= 1.02 x the line's kernel time: between boxes the clocks differ by more than that)
cd "$R"
python bench.py > gpurun_out/${tag}_bench_n1.json 2> gpurun_out/${tag}_bench_n1.err
python bench.py --mode exact --no-configs > gpurun_out/${tag}_bench_exact.json 2>> gpurun_out/${tag}_bench_n1.err
python bench.py --gather loopback --no-cpu-baseline --no-configs > gpurun_out/${tag}_bench_loopback.json 2>> gpurun_out/${tag}_bench_n1.err
python bench.py --gather loopback --gather-expand --no-cpu-baseline --no-configs > gpurun_out/${tag}_bench_loopback_expand.json 2>> gpurun_out/${tag}_bench_n1.err
echo "bench lines written"
fi
echo "round check ($part) written under gpurun_out/"
