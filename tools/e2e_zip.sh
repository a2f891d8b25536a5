mkdir -p /tmp/e2e
python tools/e2e_probe.py --loci 17000 --samples 5000 --no-gpu > /dev/null 2>&1
ls -la /tmp/e2e/
echo "== plain"; python tools/e2e_dump_only.py /tmp/e2e/synth_17000x5000.vcf.gz 2>&1 | tail -3
echo "== --zip (libtrk members)"; E2E_ZIP=1 python tools/e2e_dump_only.py /tmp/e2e/synth_17000x5000.vcf.gz 2>&1 | tail -6
echo "== check"; python - <<'P'
import sys; sys.path.insert(0,'.')
import os; os.environ['TRK_LAB']='1'
from trtools_amd import tabix
import time, hashlib, gzip
t=time.time(); scan=tabix.build('/tmp/e2e/dump.vcf.gz','/tmp/e2e/scan.tbi'); print('tabix.build scan %.2f s'%(time.time()-t))
idx=tabix.TabixIndex.load('/tmp/e2e/dump.vcf.gz.tbi')
print('index equal', (idx.names,idx.bins,idx.linear,idx.meta)==(scan.names,scan.bins,scan.linear,scan.meta))
P
echo "== --zip, python members"; TRK_BGZF_PYTHON=1 E2E_ZIP=1 python tools/e2e_dump_only.py /tmp/e2e/synth_17000x5000.vcf.gz 2>&1 | head -2
