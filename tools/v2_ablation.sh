#!/bin/bash
export TRK_LAB=1   # tools are lab runs: lab knobs are honoured (trtools_amd/_knobs.py)
# Timing-only ablation builds of the call-filter kernel (-DTRK_V2_ABL=<bits>, see trk_kernels.hip), built here (no GPU
# needed) into trtools_amd/abl/; `gpurun -- bash tools/v2_ablation.sh run` times each on the headline step.
cd "$(dirname "$0")/.."
if [ "$1" = run ]; then
  for f in trtools_amd/abl/libtrk_*.so; do
    echo "== $f"; TRK_LIBTRK=$PWD/$f python tools/v2_mode_probe.py --rounds 2 ${MODES:-0} 2>&1 | tail -${NTAIL:-1}
  done
  exit 0
fi
mkdir -p trtools_amd/abl /tmp/abl
cd trtools_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable -Wno-unused-but-set-variable -ffp-contract=off"
for spec in "$@"; do      # NAME or NAME:EXTRA_FLAGS (NAME alone = -DTRK_V2_ABL=NAME)
  n=${spec%%:*}; x=""; if [ "$spec" != "$n" ]; then x=${spec#*:}; else x="-DTRK_V2_ABL=$n"; fi
  ( hipcc $FLAGS $x -c trk_kernels.hip -o /tmp/abl/k_$n.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC /tmp/abl/k_$n.o trk_assoc.o trk_qc.o trk_api.o trk_vcf.o -o ../abl/libtrk_$n.so -ldl -lz -lpthread ) &
done
wait
ls -la ../abl
