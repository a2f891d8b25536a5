"""Second half of tools/gen_golden.py: dumpSTR goldens from the IMPORTED reference on small
synthetic VCFs (text rendering of trtools_amd.synth).  For every case the input VCF and the
reference's three outputs (.vcf through this repo's VCF writer, .samplog.tab, .loclog.tab)
are stored under tests/golden/dumpstr_synth/."""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
OUT = os.path.join(REPO, 'tests', 'golden', 'dumpstr_synth')

CASES = {
    'hipstr_all': ('hipstr', dict(hipstr_max_call_flank_indel=0.12, hipstr_max_call_stutter=0.12,
                                  hipstr_min_supp_reads=8, hipstr_min_call_DP=12, hipstr_max_call_DP=50,
                                  hipstr_min_call_Q=0.9, min_locus_callrate=0.6, min_locus_hwep=0.01,
                                  min_locus_het=0.1, max_locus_het=0.85, filter_hrun=True)),
    'hipstr_uselength_drop': ('hipstr', dict(hipstr_min_call_DP=20, hipstr_min_call_Q=0.95, use_length=True,
                                             min_locus_het=0.2, min_locus_hwep=0.05, drop_filtered=True)),
    'gangstr_all': ('gangstr', dict(gangstr_min_call_DP=12, gangstr_max_call_DP=50, gangstr_min_call_Q=0.9,
                                    gangstr_expansion_prob_het=0.05, gangstr_expansion_prob_hom=0.05,
                                    gangstr_expansion_prob_total=0.2, gangstr_filter_span_only=True,
                                    gangstr_filter_spanbound_only=True, gangstr_filter_badCI=True,
                                    min_locus_callrate=0.5)),
}

ARG_NAMES = """min_locus_callrate min_locus_hwep min_locus_het max_locus_het filter_regions filter_regions_names
hipstr_min_call_DP hipstr_max_call_DP hipstr_min_call_Q hipstr_max_call_flank_indel hipstr_max_call_stutter
hipstr_min_supp_reads longtr_min_call_DP longtr_max_call_DP longtr_min_call_Q longtr_max_call_flank_indel
longtr_min_supp_reads gangstr_expansion_prob_het gangstr_expansion_prob_hom gangstr_expansion_prob_total
gangstr_filter_badCI gangstr_min_call_DP gangstr_max_call_DP gangstr_min_call_Q advntr_min_call_DP
advntr_max_call_DP advntr_min_spanning advntr_min_flanking advntr_min_ML eh_min_ADFL eh_min_ADIR eh_min_ADSP
eh_min_call_LC eh_max_call_LC popstr_min_call_DP popstr_max_call_DP popstr_require_support num_records""".split()


def make_args(out, vcf, vcftype, **kw):
    ns = argparse.Namespace(vcf=vcf, out=out, zip=False, vcftype=vcftype, use_length=False, filter_hrun=False,
                            drop_filtered=False, gangstr_filter_span_only=False,
                            gangstr_filter_spanbound_only=False, die_on_warning=False, verbose=False)
    for n in ARG_NAMES:
        setattr(ns, n, None)
    for k, v in kw.items():
        setattr(ns, k, v)
    return ns


def write_inputs():
    from trtools_amd import synth
    os.makedirs(OUT, exist_ok=True)
    S, Lc = 24, 40
    for caller in ('hipstr', 'gangstr'):
        loci = synth.make_loci(Lc, S, seed=11 + len(caller), max_alleles=7, all_missing_frac=0.05,
                               pure_repeats=(caller == 'gangstr'))
        rows = synth.cells_numpy(77, loci, np.arange(Lc), S)
        extra = synth.gangstr_planes_numpy(77, loci, np.arange(Lc), S, rows['gt'], rows['dp']) \
            if caller == 'gangstr' else None
        synth.render_vcf(os.path.join(OUT, 'synth_%s.vcf' % caller), loci, rows, caller=caller, extra=extra)


def gen_dumpstr_synth():
    import trtools.dumpSTR.dumpSTR as rdump   # the reference
    write_inputs()
    for name, (caller, kw) in CASES.items():
        out = os.path.join(OUT, name)
        argv = sys.argv
        sys.argv = ['dumpSTR', '--synthetic-golden', name]
        try:
            rc = rdump.main(make_args(out, os.path.join(OUT, 'synth_%s.vcf' % caller), caller, **kw))
        finally:
            sys.argv = argv
        assert rc == 0, name
        print("dumpstr_synth/%s: ok" % name)


GENERATORS = {'dumpstr_synth': gen_dumpstr_synth}


# ---------------------------------------------------------------------------------------
# harmonised records of every fixture VCF (tr_harmonizer.py:264-550, 693-773) and extra
# statSTR flag combinations (use-length, region, only-passing, precision) from the reference
# ---------------------------------------------------------------------------------------
def gen_harmonized():
    import json
    import trtools.utils.tr_harmonizer as trh
    import trtools.utils.utils as rutils
    data = os.path.join(REPO, 'tests', 'golden', 'data')
    files = [('many_samples.vcf.gz', 'hipstr'), ('dumpSTR/trio_chr21_hipstr.sorted.vcf.gz', 'hipstr'),
             ('dumpSTR/trio_chr21_gangstr.sorted.vcf.gz', 'gangstr'), ('dumpSTR/test_gangstr.vcf.gz', 'gangstr'),
             ('dumpSTR/NA12878_chr21_advntr.sorted.vcf.gz', 'advntr'), ('dumpSTR/NA12878_chr21_popstr.sorted.vcf.gz', 'popstr'),
             ('dumpSTR/longtr_testfile.vcf.gz', 'longtr')]
    out = {}
    for rel, vt in files:
        reader = rutils.LoadSingleReader(os.path.join(data, rel), checkgz=False)
        assert trh.InferVCFType(reader, vt).name == vt
        recs = []
        for i, r in enumerate(trh.TRRecordHarmonizer(reader, vt)):
            if i % 7 and i > 40:
                continue            # every 7th record after the first 40
            recs.append([r.chrom, int(r.pos), int(r.end_pos), int(r.full_alleles_pos), int(r.full_alleles_end_pos),
                         r.record_id, r.motif, r.ref_allele, list(r.alt_alleles), float(r.ref_allele_length),
                         [float(x) for x in r.alt_allele_lengths], r.full_alleles is not None, r.quality_field,
                         str(r)[:200]])
        out[rel] = {'vcftype': vt, 'records': recs}
        print("harmonized %s: %d records" % (rel, len(recs)))
    with open(os.path.join(REPO, 'tests', 'golden', 'harmonized_records.json'), 'w') as fh:
        json.dump({'generator': 'tools/gen_golden_dumpstr.py gen_harmonized', 'files': out}, fh)


STAT_CASES = {
    'uselength': dict(use_length=True),
    'region': dict(region='1:1000000-2000000', afreq=True, mean=True),
    'precision7_few': dict(precision=7, thresh=False, acount=False, nalleles=False, entropy=False, mode=False),
    'only_passing': dict(only_passing=True, hwep=False),
}


def stat_args(out, vcf, **kw):
    ns = argparse.Namespace(vcf=vcf, out=out, vcftype='hipstr', samples=None, sample_prefixes=None, plot_afreq=False,
                            region=None, thresh=True, afreq=True, acount=True, hwep=True, het=True, entropy=True,
                            mean=True, mode=True, var=True, numcalled=True, use_length=False, precision=4,
                            nalleles=True, nalleles_thresh=0.1, only_passing=False)
    for k, v in kw.items():
        setattr(ns, k, v)
    return ns


def gen_statstr_flags():
    import trtools.statSTR.statSTR as rstat
    d = os.path.join(REPO, 'tests', 'golden', 'statstr_flags')
    os.makedirs(d, exist_ok=True)
    vcf = os.path.join(REPO, 'tests', 'golden', 'data', 'many_samples.vcf.gz')
    for name, kw in STAT_CASES.items():
        assert rstat.main(stat_args(os.path.join(d, name), vcf, **kw)) == 0
        print("statstr_flags/%s: ok" % name)


GENERATORS['harmonized'] = gen_harmonized
GENERATORS['statstr_flags'] = gen_statstr_flags
