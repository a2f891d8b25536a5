#!/bin/bash
export TRK_LAB=1   # tools are lab runs: lab knobs are honoured (trtools_amd/_knobs.py)
# The rocprofv3 passes behind profiles/<tag>_*: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in their own runs
# (never combined with another trace domain), around (a) the bench command -- headline step + the associaTR extra --
# (b) tools/config_probe.py (BASELINE configs[1] and configs[2], HipSTR five-filter set) and (c) tools/qc_probe.py.  Run on the GPU box:
#   gpurun -- bash tools/profile_round.sh r05
set -u
tag=${1:-r03}
repo=$(pwd)
out=$repo/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
db() { find "$out/$1" -name '*.db' | head -1; }
run3() {   # name, command...: stats pass + two PMC passes of the same command
  local name=$1; shift
  rocprofv3 --kernel-trace --stats -d "$out/${name}_stats" -o stats -- "$@" > "$out/${name}_under_rocprof.log" 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$out/${name}_fetch" -o fetch -- "$@" > "$out/${name}_pmc_fetch.log" 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$out/${name}_write" -o write -- "$@" > "$out/${name}_pmc_write.log" 2>&1
  ( cd "$repo" && python tools/rocprof_summary.py stats "$(db ${name}_stats)" > "$out/${tag}_${name}_kernel_stats.csv" \
    && python tools/rocprof_summary.py pmc "$(db ${name}_fetch)" "$(db ${name}_write)" > "$out/${tag}_${name}_pmc_fetch_write.csv" )
}
run3 bench python "$repo/bench.py" --steps 5 --warmup 2 --no-check --no-cpu-baseline --no-extras
run3 configs python "$repo/tools/config_probe.py"
run3 qc python "$repo/tools/qc_probe.py" --iters 3
# round 5: the call-filter pass's builds one by one -- the 13 B compact build dumpSTR's command line launches, the HipSTR
# five-filter set full and compact -- with their own FETCH / WRITE passes
run3 variants python "$repo/tools/cf_variants_probe.py" --iters 3 full compact hipstr5 hipstr5c
# round 5: BGZF members inflated on the device (the 1 GB probe file: written here if it is not there)
mkdir -p /tmp/e2e
[ -f /tmp/e2e/synth_17000x5000.vcf.gz ] || python "$repo/tools/e2e_probe.py" --loci 17000 --samples 5000 --no-gpu > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d "$out/inflate_stats" -o stats -- python "$repo/tools/inflate_probe.py" /tmp/e2e/synth_17000x5000.vcf.gz 4096 0 > "$out/inflate_under_rocprof.log" 2>&1
( cd "$repo" && python tools/rocprof_summary.py stats "$(db inflate_stats)" > "$out/${tag}_inflate_kernel_stats.csv" )
# round 6: BGZF members deflated on the device (150 MB of dumpSTR-like text per call)
rocprofv3 --kernel-trace --stats -d "$out/deflate_stats" -o stats -- python "$repo/tools/deflate_probe.py" 150 > "$out/deflate_under_rocprof.log" 2>&1
( cd "$repo" && python tools/rocprof_summary.py stats "$(db deflate_stats)" > "$out/${tag}_deflate_kernel_stats.csv" )
# statSTR --samples (sample groups): kernel trace only
rocprofv3 --kernel-trace --stats -d "$out/groups_stats" -o stats -- python "$repo/tools/groups_probe.py" > "$out/groups_under_rocprof.log" 2>&1
( cd "$repo" && python tools/rocprof_summary.py stats "$(db groups_stats)" > "$out/${tag}_groups_kernel_stats.csv" )
# associaTR with 1 ... 62 trait columns (MFMA scan, Gram correction, per-locus solve): kernel trace only
rocprofv3 --kernel-trace --stats -d "$out/assoc_stats" -o stats -- python "$repo/tools/assoc_wide_probe.py" --m 1 15 31 45 62 > "$out/assoc_under_rocprof.log" 2>&1
( cd "$repo" && python tools/rocprof_summary.py stats "$(db assoc_stats)" > "$out/${tag}_assoc_kernel_stats.csv" )
cd "$repo"
python tools/pmc_traffic.py "$out/${tag}_bench_pmc_fetch_write.csv" "$out/${tag}_configs_pmc_fetch_write.csv" "$tag" > "$out/${tag}_pmc_traffic.json"
grep -h "^{" "$out/bench_under_rocprof.log" | tail -1 | cut -c1-300
head -14 "$out/${tag}_bench_kernel_stats.csv"
head -12 "$out/${tag}_configs_kernel_stats.csv"
grep "configs\|HipSTR" "$out/configs_under_rocprof.log" | cut -c1-260
cat "$out/${tag}_pmc_traffic.json" | head -40
