#!/bin/bash
export TRK_LAB=1   # tools are lab runs: lab knobs are honoured (trtools_amd/_knobs.py)
# what the GPU box lets a process see about the placement of its device memory (round 5: the output-pair effect)
echo "== debugfs"; ls /sys/kernel/debug 2>&1 | head; mount | grep -i debug
ls /sys/kernel/debug/dri 2>&1 | head
for f in /sys/kernel/debug/dri/*/amdgpu_vram_mm; do echo "-- $f"; head -40 "$f" 2>&1; done
echo "== partitions"; rocm-smi --showmemorypartition --showcomputepartition 2>&1 | head -20
echo "== kfd mem banks"; for f in /sys/class/kfd/kfd/topology/nodes/*/mem_banks/*/properties; do echo "-- $f"; cat $f; done 2>&1 | head -60
echo "== amdgpu params"; for p in vm_fragment_size vm_block_size vm_size mtype_local; do echo -n "$p = "; cat /sys/module/amdgpu/parameters/$p 2>&1; done
echo "== drm mem info"; for f in /sys/class/drm/card*/device/mem_info_vram_total /sys/class/drm/card*/device/mem_info_vram_used /sys/class/drm/card*/device/current_memory_partition /sys/class/drm/card*/device/current_compute_partition; do echo -n "$f = "; cat $f 2>&1; done
nproc; cat /sys/fs/cgroup/cpu.max
