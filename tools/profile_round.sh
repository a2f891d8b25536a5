#!/bin/bash
# The three rocprofv3 passes behind profiles/<tag>_*: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in their own
# runs (never combined with another trace domain), all around the same bench command.  Run on the GPU box:
#   gpurun -- bash tools/profile_round.sh r01e
set -u
tag=${1:-r01e}
repo=$(pwd)
out=$repo/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
cmd="python $repo/bench.py --steps 5 --warmup 2"
rocprofv3 --kernel-trace --stats -d "$out/stats" -o stats -- $cmd > "$out/bench_under_rocprof.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$out/fetch" -o fetch -- $cmd --steps 2 > "$out/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$out/write" -o write -- $cmd --steps 2 > "$out/pmc_write.log" 2>&1
cd "$repo"
db() { find "$out/$1" -name '*.db' | head -1; }
python tools/rocprof_summary.py stats "$(db stats)" > "$out/${tag}_kernel_stats.csv"
python tools/rocprof_summary.py pmc "$(db fetch)" "$(db write)" > "$out/${tag}_pmc_fetch_write.csv"
tail -1 "$out/bench_under_rocprof.log" | cut -c1-400
head -12 "$out/${tag}_kernel_stats.csv"
grep -i "call_filter\|locus_count\|assoc_scan\|k_synth" "$out/${tag}_pmc_fetch_write.csv" | head -20
