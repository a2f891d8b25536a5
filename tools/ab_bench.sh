#!/bin/bash
export TRK_LAB=1   # tools are lab runs: lab knobs are honoured (trtools_amd/_knobs.py)
# A/B of two library builds on ONE GPU box (box-to-box variance is +-5 %): build the variant next to the product
# library, e.g.
#   hipcc ... -DTRK_V2_WRED=0 -c trk_kernels.hip -o /tmp/k.o && hipcc -shared ... -o trtools_amd/libtrk_w0.so
# then `gpurun -- bash tools/ab_bench.sh`: three alternating bench runs (ms/step, call-filter ms, roofline fraction).
for r in 1 2 3; do
for v in w1 w0; do
  if [ $v = w0 ]; then cp trtools_amd/libtrk.so /tmp/keep.so; cp trtools_amd/libtrk_w0.so trtools_amd/libtrk.so; fi
  echo "== $v"; python bench.py --no-assoc --steps 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernels_ms']['k_call_filter'], d['roofline']['frac'])"
  if [ $v = w0 ]; then cp /tmp/keep.so trtools_amd/libtrk.so; fi
done; done
python -m pytest tests/test_gpu_callfilters.py tests/test_gpu_baseline_configs.py -q -m gpu -x 2>&1 | tail -1
