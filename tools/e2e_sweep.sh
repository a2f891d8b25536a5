#!/bin/bash
export TRK_LAB=1   # tools are lab runs: lab knobs are honoured (trtools_amd/_knobs.py)
# Host-side knobs of the 1 GB end-to-end runs (tools/e2e_dump_only.py, e2e_stat_only.py) on the GPU box: reader threads,
# formatter threads, read-ahead.  usage: tools/e2e_sweep.sh  (writes gpurun_out/e2e_sweep.txt)
cd "$(dirname "$0")/.."
out=gpurun_out/e2e_sweep.txt
mkdir -p gpurun_out /tmp/e2e
nproc > $out; df /tmp | tail -1 >> $out
timeout 900 python tools/e2e_probe.py --loci 17000 --samples 5000 --no-gpu 2>&1 | head -3 >> $out
f=/tmp/e2e/synth_17000x5000.vcf.gz
run() {   # $1 = label, rest = env assignments
  local label=$1; shift
  echo "== dumpSTR $label" >> $out
  env "$@" timeout 120 python tools/e2e_dump_only.py $f 2>&1 | grep "^run" >> $out
}
runs() {
  local label=$1; shift
  echo "== statSTR $label" >> $out
  env "$@" timeout 120 python tools/e2e_stat_only.py $f 2>&1 | grep -E "^run|timed" >> $out
}
run default A=1
for vt in 16 32 48; do for ft in 16 32; do
  run "RA=1 VCF_THREADS=$vt FMT_THREADS=$ft" TRK_VCF_READ_AHEAD=1 TRK_VCF_THREADS=$vt TRK_FMT_THREADS=$ft
done; done
run "VCF_THREADS=32" TRK_VCF_THREADS=32
run "VCF_THREADS=96" TRK_VCF_THREADS=96
run "VCF_THREADS=32 FMT_THREADS=48" TRK_VCF_THREADS=32 TRK_FMT_THREADS=48
runs default A=1
for vt in 16 32 48; do
  runs "RA=1 VCF_THREADS=$vt" TRK_VCF_READ_AHEAD=1 TRK_VCF_THREADS=$vt
done
runs "VCF_THREADS=32" TRK_VCF_THREADS=32
runs "VCF_THREADS=96" TRK_VCF_THREADS=96
cat $out
