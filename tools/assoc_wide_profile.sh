#!/bin/bash
export TRK_LAB=1   # tools are lab runs: lab knobs are honoured (trtools_amd/_knobs.py)
# per-kernel durations of tools/assoc_wide_probe.py under rocprofv3; usage: assoc_wide_profile.sh 32 62
repo=$(cd "$(dirname "$0")/.." && pwd)
out=$repo/gpurun_out/r03/assoc_wide_prof
mkdir -p "$out"; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$out/stats" -o stats -- python "$repo/tools/assoc_wide_probe.py" --m "$@" > "$out/under_rocprof.log" 2>&1
db=$(find "$out/stats" -name '*.db' | head -1)
if [ -n "$db" ]; then ( cd "$repo" && python tools/rocprof_summary.py stats "$db" | head -14 | cut -c1-200 ); else tail -5 "$out/under_rocprof.log"; fi
