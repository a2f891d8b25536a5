"""Native VCF / BGZF reader (include/trk_vcf.h) against the Python decoder (trtools_amd/vcfio.py,
itself pinned through the reference's golden outputs): same records, same genotype arrays, same
FORMAT arrays, for plain / gzip / bgzip inputs and every decode kind.  CPU only."""
import glob
import gzip
import os
import shutil

import numpy as np
import pytest

from helpers import GOLDEN

D = os.path.join(GOLDEN, 'data')
FILES = [os.path.join(D, 'many_samples.vcf.gz')] + sorted(glob.glob(os.path.join(D, 'dumpSTR', '*.sorted.vcf.gz'))) + \
    [os.path.join(D, 'dumpSTR', 'test_gangstr.vcf.gz'), os.path.join(D, 'dumpSTR', 'longtr_testfile.vcf.gz'),
     os.path.join(GOLDEN, 'dumpstr_synth', 'synth_hipstr.vcf'), os.path.join(GOLDEN, 'dumpstr_synth', 'synth_gangstr.vcf')]


def _same(a, b):
    if a.dtype.kind == 'f':
        return np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)])
    return np.array_equal(a, b)


def _compare(path, batch_records=None, max_ploidy=2):
    from trtools_amd import vcfio, vcfnative
    py = list(vcfio.VCFReader(path))
    r = vcfnative.NativeVCFReader(path, batch_records=batch_records, max_ploidy=max_ploidy)
    num = [k for k, (t, n) in r.format_types.items() if t in ('Integer', 'Float') and k != 'GT']
    ncols = {}
    for v in py:
        for k in num:
            if k in v.FORMAT:
                ncols[k] = max(ncols.get(k, 1), v.format(k).shape[1])
    for k in num:
        r.select_format(k, ncol=ncols.get(k, 1))
    nat = list(r)
    assert len(py) == len(nat)
    assert r.samples == vcfio.VCFReader(path).samples
    for x, y in zip(py, nat):
        assert (x.CHROM, x.POS, x.ID, x.REF, x.ALT, x.FILTER, x.FORMAT) == (y.CHROM, y.POS, y.ID, y.REF, y.ALT, y.FILTER, y.FORMAT)
        assert dict(x.INFO) == dict(y.INFO)
        if x.genotype is not None:
            assert np.array_equal(x.genotype.array(), y.genotype.array())
            assert x.ploidy == y.ploidy
        for k in num:
            if k in x.FORMAT:
                xa, ya = x.format(k), y.format(k)
                assert _same(xa, ya[:, :xa.shape[1]]), (path, x.POS, k)
                # beyond the record's own width the native plane holds padding only
                pad = ya[:, xa.shape[1]:]
                assert np.all(np.isnan(pad)) if pad.dtype.kind == 'f' else np.all((pad == -2147483647) | (pad == -2147483648))
        # undecoded fields are still reachable through the record text
        for k in x.FORMAT:
            if k not in num and k != 'GT':
                assert np.array_equal(x.format(k), y.format(k))
        assert str(x) == str(y)
    return len(py)


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_native_equals_python_reader(path):
    assert _compare(path) > 0


def test_batch_boundaries_and_gzip_flavours(tmp_path):
    src = os.path.join(GOLDEN, 'dumpstr_synth', 'synth_hipstr.vcf')
    assert _compare(src, batch_records=7) == 40
    plain_gz = str(tmp_path / 'plain.vcf.gz')          # ordinary gzip stream (not BGZF)
    with open(src, 'rb') as fi, gzip.open(plain_gz, 'wb') as fo:
        shutil.copyfileobj(fi, fo)
    assert _compare(plain_gz, batch_records=13) == 40
    nonl = str(tmp_path / 'nonl.vcf')                   # no trailing newline, CRLF line ends
    open(nonl, 'wb').write(open(src, 'rb').read().rstrip(b'\n').replace(b'\n', b'\r\n'))
    assert _compare(nonl) == 40


def test_string_field_preparse_kinds():
    """TRK_VCF_MINSUPP / INT (RC) / INT_RANGES (REPCI) against the Python pre-parsers of
    trtools_amd/dumpSTR/filters.py (which follow the reference's filters.py:519-567, 692, 745-748)."""
    from trtools_amd import vcfio, vcfnative
    from trtools_amd.dumpSTR import filters
    from trtools_amd.utils import tr_harmonizer as trh
    for path, caller in ((os.path.join(D, 'dumpSTR', 'trio_chr21_hipstr.sorted.vcf.gz'), 'hipstr'),
                         (os.path.join(GOLDEN, 'dumpstr_synth', 'synth_hipstr.vcf'), 'hipstr'),
                         (os.path.join(D, 'dumpSTR', 'trio_chr21_gangstr.sorted.vcf.gz'), 'gangstr'),
                         (os.path.join(GOLDEN, 'dumpstr_synth', 'synth_gangstr.vcf'), 'gangstr')):
        r = vcfnative.NativeVCFReader(path)
        if caller == 'hipstr':
            r.select_format('ALLREADS', vcfnative.KIND_MINSUPP, 1, alias='__minsupp')
        else:
            r.select_format('RC', vcfnative.KIND_INT, 4, alias='__rc')
            r.select_format('REPCI', vcfnative.KIND_INT_RANGES, 4, alias='__repci')
        n = 0
        for vpy, vnat in zip(vcfio.VCFReader(path), r):
            rec = trh.HarmonizeRecord(caller, vpy)
            if caller == 'hipstr':
                want = filters._min_supp_reads(rec)[:, 0]
                got = vnat.format('__minsupp')[:, 0]
                called = rec.GetCalledSamples()
                assert np.array_equal(want[called], got[called]), (path, vpy.POS)
            else:
                called = rec.GetCalledSamples()     # the GangSTR filters only look at called samples
                if not called.any():
                    n += 1
                    continue
                assert np.array_equal(filters._rc_plane(rec)[called], vnat.format('__rc')[called]), (path, vpy.POS)
                assert np.array_equal(filters._repci_plane(rec)[called], vnat.format('__repci')[called]), (path, vpy.POS)
            n += 1
            if n >= 1500:
                break
        assert n > 0


def test_ploidy_above_tensor_is_an_error(tmp_path):
    from trtools_amd import vcfnative
    p = tmp_path / 't.vcf'
    p.write_text('##fileformat=VCFv4.1\n##FORMAT=<ID=GT,Number=1,Type=String,Description="g">\n'
                 '#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tA\tB\n'
                 '1\t5\t.\tA\tAA\t.\t.\t.\tGT\t0/1/1\t0\n')
    with pytest.raises(ValueError):
        list(vcfnative.NativeVCFReader(str(p), max_ploidy=2))
    v = list(vcfnative.NativeVCFReader(str(p), max_ploidy=3))[0]
    assert v.genotype.array().tolist() == [[0, 1, 1, 0], [0, -2, -2, 0]] and v.ploidy == 3
