#!/usr/bin/env python3
"""Record what the REAL reference's dumpSTR does with the argument sets of tests/dumpstr_more_cases.py
(build container only; cyvcf2/pysam through tools/refshim).

    python tools/gen_golden_dumpstr_more.py     # rewrites tests/golden/dumpstr_more/

Per case: the return code (results.json) and, when it is 0, <case>.samplog.tab, <case>.loclog.tab and
<case>.vcf.gz (the output VCF re-compressed with plain gzip; only data the reference wrote)."""
import contextlib
import gzip
import io
import json
import os
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, 'refshim'))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
sys.path.insert(0, HERE)

from dumpstr_more_cases import CASES, OUT   # noqa: E402
import gen_golden_dumpstr as gg            # noqa: E402


def run_cases(main, outdir, quiet=True):
    """Run every case with ``main`` (the reference's or this repo's); returns {name: rc}."""
    rcs = {}
    for name, vcf, kw in CASES:
        if vcf.startswith('@'):
            vcf = os.path.join(outdir, vcf[1:] + '.vcf')
        kw = dict(kw)
        args = gg.make_args(os.path.join(outdir, name), vcf, kw.pop('vcftype', 'auto'), **kw)
        argv, sys.argv = sys.argv, ['dumpSTR', '--case', name]
        try:
            with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
                try:
                    rcs[name] = int(main(args))
                except SystemExit as e:
                    rcs[name] = 'exit:%s' % e.code
                except Exception as e:          # noqa: BLE001 -- the exception type is the golden
                    rcs[name] = 'raise:%s' % type(e).__name__
        finally:
            sys.argv = argv
    return rcs


def main():
    sys.path.insert(0, '/root/reference')
    import trtools.dumpSTR.dumpSTR as rdump     # the reference
    tmp = tempfile.mkdtemp()
    rcs = run_cases(rdump.main, tmp)
    shutil.rmtree(OUT, ignore_errors=True)
    os.makedirs(OUT)
    for name, _, kw in CASES:
        print('%-32s rc=%s' % (name, rcs[name]))
        if rcs[name] != 0:
            continue
        for ext in ('.samplog.tab', '.loclog.tab'):
            shutil.copy(os.path.join(tmp, name + ext), os.path.join(OUT, name + ext))
        src = os.path.join(tmp, name + ('.vcf.gz' if kw.get('zip') else '.vcf'))
        opener = gzip.open if kw.get('zip') else open
        with opener(src, 'rb') as fin, gzip.open(os.path.join(OUT, name + '.vcf.gz'), 'wb') as fout:
            fout.write(fin.read())
    with open(os.path.join(OUT, 'results.json'), 'w') as fh:
        json.dump({'generator': 'tools/gen_golden_dumpstr_more.py', 'rc': rcs}, fh, indent=1, sort_keys=True)
    shutil.rmtree(tmp)


if __name__ == '__main__':
    main()
