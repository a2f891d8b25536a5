#!/usr/bin/env python3
"""End-to-end probe: text VCF -> native reader -> packed batch -> GPU -> statSTR table.
Generates a synthetic HipSTR-shape bgzip VCF (GT:DP:Q), then times the readers and the CLI."""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
ap = argparse.ArgumentParser()
ap.add_argument('--loci', type=int, default=2000)
ap.add_argument('--samples', type=int, default=5000)
ap.add_argument('--out', default='/tmp/e2e')
ap.add_argument('--no-gpu', action='store_true')
a = ap.parse_args()
from trtools_amd import synth, vcfio, vcfnative
from trtools_amd.bgzf import BgzfWriter
os.makedirs(a.out, exist_ok=True)
path = os.path.join(a.out, 'synth_%dx%d.vcf.gz' % (a.loci, a.samples))
t = time.time()
if not os.path.exists(path):
    loci = synth.make_loci(a.loci, a.samples, seed=5)
    with BgzfWriter(path, level=1) as fh:
        fh.write('##fileformat=VCFv4.1\n##command=HipSTR-v0.6.2 --synthetic\n')
        for k in ('START', 'END', 'PERIOD'):
            fh.write('##INFO=<ID=%s,Number=1,Type=Integer,Description="%s">\n' % (k, k))
        fh.write('##FORMAT=<ID=GT,Number=1,Type=String,Description="GT">\n##FORMAT=<ID=DP,Number=1,Type=Integer,Description="DP">\n##FORMAT=<ID=Q,Number=1,Type=Float,Description="Q">\n')
        fh.write('#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t' + '\t'.join('S%05d' % i for i in range(a.samples)) + '\n')
        for l0 in range(0, a.loci, 64):
            idx = np.arange(l0, min(a.loci, l0 + 64))
            rows = synth.cells_numpy(5, loci, idx, a.samples)
            for r, l in enumerate(idx):
                strs = loci.allele_strs[l]
                pos = 1000 + 500 * int(l)
                g0 = np.where(rows['gt'][r, :, 0] < 0, '.', rows['gt'][r, :, 0].astype(str))
                g1 = np.where(rows['gt'][r, :, 1] < 0, '.', rows['gt'][r, :, 1].astype(str))
                dp = np.where(rows['dp'][r] == -2147483648, '.', rows['dp'][r].astype(str))
                q = np.where(np.isnan(rows['q'][r]), '.', np.char.mod('%g', rows['q'][r]))
                cols = np.char.add(np.char.add(np.char.add(np.char.add(g0, '|'), g1), ':'), np.char.add(np.char.add(dp, ':'), q))
                fh.write('\t'.join(['chr1', str(pos), 'STR_%d' % l, strs[0], ','.join(strs[1:]) or '.', '.', '.',
                                    'START=%d;END=%d;PERIOD=%d' % (pos, pos + len(strs[0]) - 1, len(loci.motifs[l])),
                                    'GT:DP:Q']) + '\t' + '\t'.join(cols) + '\n')
    print("generated %s (%.1f MB) in %.1fs" % (path, os.path.getsize(path) / 1e6, time.time() - t))
cells = a.loci * a.samples
t = time.time(); r = vcfnative.NativeVCFReader(path); n = sum(1 for _ in r); t1 = time.time() - t
print("native reader GT only      : %6.2fs  %.2e cells/s" % (t1, cells / t1))
t = time.time(); r = vcfnative.NativeVCFReader(path); r.select_format('DP'); r.select_format('Q'); n = sum(1 for _ in r); t2 = time.time() - t
print("native reader GT+DP+Q      : %6.2fs  %.2e cells/s" % (t2, cells / t2))
t = time.time()
for i, v in enumerate(vcfio.VCFReader(path)):
    v.format('DP'); v.format('Q')
    if i >= 99: break
t3 = (time.time() - t) / 100 * a.loci
print("python reader GT+DP+Q (extrapolated from 100 records): %6.2fs  %.2e cells/s" % (t3, cells / t3))
if not a.no_gpu:
    import argparse as ap2
    from trtools_amd.statSTR import statSTR
    ns = ap2.Namespace(vcf=path, out=os.path.join(a.out, 'stat'), vcftype='hipstr', samples=None, sample_prefixes=None,
                       plot_afreq=False, region=None, thresh=True, afreq=True, acount=True, hwep=True, het=True,
                       entropy=True, mean=True, mode=True, var=True, numcalled=True, use_length=False, precision=4,
                       nalleles=True, nalleles_thresh=0.01, only_passing=False)
    for mode in ('1', '1', '0'):
        os.environ['TRK_STATSTR_BATCH'] = mode
        t = time.time(); rc = statSTR.main(ns); t4 = time.time() - t
        print("statSTR CLI end to end (11 stats, %s path): rc=%d %6.3fs  %.0f loci/s  %.2e cells/s" % (
            'batch' if mode == '1' else 'per-record', rc, t4, a.loci / t4, cells / t4))
        if mode == '1':
            os.replace(ns.out + '.tab', ns.out + '.batch.tab')
    print("batch path table == per-record table:", open(ns.out + '.batch.tab').read() == open(ns.out + '.tab').read())
    os.environ['TRK_STATSTR_BATCH'] = '1'
    os.environ['TRK_VCF_TIMING'] = '1'
    statSTR.main(ns)
    del os.environ['TRK_VCF_TIMING']
    if os.environ.get('E2E_PROFILE'):
        import cProfile, pstats
        pr = cProfile.Profile(); pr.enable(); statSTR.main(ns); pr.disable()
        pstats.Stats(pr).sort_stats('tottime').print_stats(16)
    # dumpSTR end to end: the same file through call filters + locus filters to an output VCF and the two logs
    from trtools_amd.dumpSTR import dumpSTR
    old = sys.argv
    sys.argv = ['dumpSTR', '--vcf', path, '--out', os.path.join(a.out, 'dump'), '--vcftype', 'hipstr',
                '--hipstr-min-call-DP', '10', '--hipstr-max-call-DP', '55', '--hipstr-min-call-Q', '0.9',
                '--min-locus-callrate', '0.8', '--min-locus-hwep', '0.001', '--min-locus-het', '0.05',
                '--max-locus-het', '0.9']
    dargs = dumpSTR.getargs()
    sys.argv = old
    t = time.time(); rc = dumpSTR.main(dargs); t5 = time.time() - t
    print("dumpSTR CLI end to end (3 call + 4 locus filters, VCF out %.0f MB): rc=%d %6.2fs  %.0f loci/s  %.2e cells/s" % (
        os.path.getsize(os.path.join(a.out, 'dump.vcf')) / 1e6, rc, t5, a.loci / t5, cells / t5))
    print("dumpSTR path / phases:", {k: (v if not isinstance(v, dict) else {p: round(x, 3) for p, x in v.items()})
                                     for k, v in dumpSTR.LAST_RUN.items()})
    t = time.time(); rc = dumpSTR.main(dargs); t5 = time.time() - t
    print("dumpSTR CLI second run: %6.2fs  %.2e cells/s" % (t5, cells / t5))
    if os.environ.get('E2E_PROFILE'):
        import cProfile, pstats
        pr = cProfile.Profile(); pr.enable(); dumpSTR.main(dargs); pr.disable()
        pstats.Stats(pr).sort_stats('tottime').print_stats(18)
