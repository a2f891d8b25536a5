#!/bin/bash
export TRK_LAB=1   # tools are lab runs: lab knobs are honoured (trtools_amd/_knobs.py)
# Kernels of libtrk.so that use scratch memory, per translation unit, from the compiler's own resource remarks
# (`-Rpass-analysis=kernel-resource-usage`): name, VGPRs, scratch bytes per lane.  Run from anywhere; no GPU needed.
cd "$(dirname "$0")/../trtools_amd/csrc" || exit 1
for f in trk_kernels trk_hwe trk_assoc trk_qc trk_parse trk_api; do
    [ -f $f.hip ] || continue
    extra=""; [ $f = trk_hwe ] && extra="-mllvm -disable-machine-licm"
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $extra --cuda-device-only -c $f.hip -o /dev/null \
        -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|  VGPRs:|ScratchSize" |
        sed 's/.*remark: *//; s/ \[-Rpass.*//' | paste - - - |
        awk -v f=$f '{n++} $NF>0{print f".hip\t"$3"\tVGPRs "$5"\tscratch "$NF} END{print f".hip: "n" kernels"}'
done | c++filt | cut -c1-170
