#!/usr/bin/env python3
"""Record what the REAL reference's statSTR writes for tests/statstr_more_cases.py (build container only).

    python tools/gen_golden_statstr_more.py     # rewrites tests/golden/statstr_more/
"""
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

from statstr_more_cases import CASES, OUT    # noqa: E402
import gen_golden_dumpstr as gg             # noqa: E402


def run_cases(main, outdir):
    rcs = {}
    for name, vcf, vcftype, kw in CASES:
        args = gg.stat_args(os.path.join(outdir, name), vcf, **kw)
        args.vcftype = vcftype
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            try:
                rcs[name] = int(main(args))
            except SystemExit as e:
                rcs[name] = 'exit:%s' % e.code
            except Exception as e:      # noqa: BLE001 -- the exception type is the golden
                rcs[name] = 'raise:%s' % type(e).__name__
    return rcs


def main():
    sys.path.insert(0, '/root/reference')
    import trtools.statSTR.statSTR as rstat     # the reference
    tmp = tempfile.mkdtemp()
    rcs = run_cases(rstat.main, tmp)
    shutil.rmtree(OUT, ignore_errors=True)
    os.makedirs(OUT)
    for name, *_ in CASES:
        print('%-28s rc=%s' % (name, rcs[name]))
        if rcs[name] == 0:
            with open(os.path.join(tmp, name + '.tab'), 'rb') as fin, \
                    gzip.GzipFile(os.path.join(OUT, name + '.tab.gz'), 'wb', mtime=0) as fout:
                fout.write(fin.read())
    with open(os.path.join(OUT, 'results.json'), 'w') as fh:
        json.dump({'generator': 'tools/gen_golden_statstr_more.py', 'rc': rcs}, fh, indent=1, sort_keys=True)
    shutil.rmtree(tmp)


if __name__ == '__main__':
    main()
