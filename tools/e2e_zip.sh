#!/bin/bash
# dumpSTR's command line at 1.02 GB: plain, --zip with the host compressor at levels 6 / 1, --zip with the members deflated on
# the device, and (round 5's path) with the interpreter's zlib members; the writer-built index against a scan of the file
export TRK_LAB=1
mkdir -p /tmp/e2e
[ -f /tmp/e2e/synth_17000x5000.vcf.gz ] || python tools/e2e_probe.py --loci 17000 --samples 5000 --no-gpu > /dev/null 2>&1
echo "== plain"; python tools/e2e_dump_only.py /tmp/e2e/synth_17000x5000.vcf.gz 2>&1 | tail -2
echo "== --zip, libtrk members level 6"; E2E_ZIP=1 python tools/e2e_dump_only.py /tmp/e2e/synth_17000x5000.vcf.gz 2>&1 | tail -4
echo "== --zip, libtrk members level 1"; TRK_ZIP_LEVEL=1 E2E_ZIP=1 python tools/e2e_dump_only.py /tmp/e2e/synth_17000x5000.vcf.gz 2>&1 | tail -4
echo "== --zip, members deflated on the device"; TRK_DEVICE_DEFLATE=1 E2E_ZIP=1 python tools/e2e_dump_only.py /tmp/e2e/synth_17000x5000.vcf.gz 2>&1 | tail -4
echo "== check (the last file)"; python - <<'P'
import sys; sys.path.insert(0,'.')
from trtools_amd import tabix, vcfnative
import time
t=time.time(); scan=tabix.build('/tmp/e2e/dump.vcf.gz','/tmp/e2e/scan.tbi'); print('tabix.build scan %.2f s'%(time.time()-t))
idx=tabix.TabixIndex.load('/tmp/e2e/dump.vcf.gz.tbi')
print('index equal', (idx.names,idx.bins,idx.linear,idx.meta)==(scan.names,scan.bins,scan.linear,scan.meta))
r=vcfnative.NativeVCFReader('/tmp/e2e/dump.vcf.gz'); n=0
while True:
    rb=r.read_raw_batch(4096)
    if rb.n==0: break
    n+=rb.n
print('records read back by the native reader:', n)
P
echo "== --zip, python members"; TRK_BGZF_PYTHON=1 E2E_ZIP=1 python tools/e2e_dump_only.py /tmp/e2e/synth_17000x5000.vcf.gz 2>&1 | head -2
