"""The caller-side worker pool of libtrk (trk_vcf_harmonize, trk_vcf_statstr_rows, the record writers: one process-wide
pool of parked threads) in a FORKED child: the parent's threads do not exist there, the child must get a pool of its own
instead of waiting for them.  CPU only."""
import os
import sys

import pytest

from test_vcfnative_hook import _synthetic


def _harmonise_all(path):
    from trtools_amd import vcfnative
    r = vcfnative.NativeVCFReader(path, batch_records=1000)
    n = 0
    while True:
        rb = r._read_raw_batch(1000)
        if rb.n == 0:
            break
        rb.harmonize('hipstr')          # (3000 records: 47 chunks on the pool's threads)
        n += rb.n
    r.close()
    return n


@pytest.mark.timeout(120)
def test_harmonise_in_a_forked_child(tmp_path):
    path = str(tmp_path / 'f.vcf')
    open(path, 'wb').write(_synthetic(3000, 3, seed=2))
    assert _harmonise_all(path) == 3000          # the parent's pool exists now
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:
        try:
            os.write(w, str(_harmonise_all(path)).encode())
        finally:
            os._exit(0)
    os.close(w)
    _, status = os.waitpid(pid, 0)
    assert status == 0 and os.read(r, 64) == b'3000'
    assert _harmonise_all(path) == 3000          # ... and the parent's still works
