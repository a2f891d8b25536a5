"""The 1 GB command lines' outputs against the per-record loop's (TRK_DUMPSTR_BATCH=0 / TRK_STATSTR_BATCH=0) on the SAME
file: byte for byte (VERDICT r03 item 2's closing criterion).  The per-record loop takes a minute per GB; the file
is tools/e2e_probe.py's, cut to --loci records so that the slow side stays bounded.
usage: e2e_identity.py /tmp/e2e/synth_17000x5000.vcf.gz [--loci 3000]"""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import argparse, hashlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument('vcf')
ap.add_argument('--loci', type=int, default=3000)
a = ap.parse_args()
dump = ['--vcftype', 'hipstr', '--hipstr-min-call-DP', '10', '--hipstr-max-call-DP', '55', '--hipstr-min-call-Q', '0.9',
        '--min-locus-callrate', '0.8', '--min-locus-hwep', '0.001', '--min-locus-het', '0.05', '--max-locus-het', '0.9',
        '--num-records', str(a.loci)]
stat = ['--vcftype', 'hipstr', '--thresh', '--afreq', '--acount', '--hwep', '--het', '--entropy', '--mean', '--mode', '--var',
        '--numcalled', '--nalleles']


def run(mod, args, out, env):
    e = dict(os.environ, **env)
    t = time.time()
    subprocess.run([sys.executable, '-c', 'import sys; sys.path.insert(0, %r); from trtools_amd.%s import %s as m; sys.argv = ["x"] + sys.argv[1:]; raise SystemExit(m.run())'
                    % (ROOT, mod, mod)] + args + ['--vcf', a.vcf, '--out', out], check=True, env=e,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return time.time() - t


def digest(path):
    """sha256 and size of the file without its '##command-DumpSTR=' header line (which quotes the --out prefix)."""
    h = hashlib.sha256()
    n = 0
    with open(path, 'rb') as fh:
        for line in fh:
            if line.startswith(b'##command-DumpSTR='):
                continue
            h.update(line)
            n += len(line)
    return h.hexdigest()[:16], n


os.makedirs('/tmp/e2e/id', exist_ok=True)
ok = True
tb = run('dumpSTR', dump, '/tmp/e2e/id/batch', {})
tr = run('dumpSTR', dump, '/tmp/e2e/id/rec', {'TRK_DUMPSTR_BATCH': '0'})
print("dumpSTR first %d records: batch pipeline %.2f s, per-record loop %.2f s" % (a.loci, tb, tr))
for suf in ('.vcf', '.samplog.tab', '.loclog.tab'):
    x, y = digest('/tmp/e2e/id/batch' + suf), digest('/tmp/e2e/id/rec' + suf)
    print("  %-13s %s %d bytes  %s" % (suf, x[0], x[1], 'identical' if x == y else 'DIFFERENT (%s, %d)' % y))
    ok &= x == y
    if x != y and suf == '.vcf':
        with open('/tmp/e2e/id/batch.vcf') as fa, open('/tmp/e2e/id/rec.vcf') as fb:
            shown = 0
            for i, (la, lb) in enumerate(zip(fa, fb)):
                if la != lb:
                    ca, cb = la.rstrip('\n').split('\t'), lb.rstrip('\n').split('\t')
                    cols = [j for j, (p, q) in enumerate(zip(ca, cb)) if p != q]
                    print("    line %d: %d / %d columns, differing columns %s" % (i, len(ca), len(cb), cols[:6]))
                    for j in cols[:3]:
                        print("      col %d: batch %r   per-record %r" % (j, ca[j][:80], cb[j][:80]))
                    shown += 1
                    if shown >= 4:
                        break
# statSTR has no --num-records: the whole file through both
tb = run('statSTR', stat, '/tmp/e2e/id/sbatch', {})
tr = run('statSTR', stat, '/tmp/e2e/id/srec', {'TRK_STATSTR_BATCH': '0'}) if os.environ.get('E2E_STAT_PER_RECORD', '1') == '1' else None
if tr is not None:
    x, y = digest('/tmp/e2e/id/sbatch.tab'), digest('/tmp/e2e/id/srec.tab')
    print("statSTR whole file: batch pipeline %.2f s, per-record loop %.2f s;  .tab %s %d bytes  %s" % (
        tb, tr, x[0], x[1], 'identical' if x == y else 'DIFFERENT'))
    ok &= x == y
raise SystemExit(0 if ok else 1)
