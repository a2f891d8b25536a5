"""Second half of tools/gen_golden.py: dumpSTR goldens from the IMPORTED reference on small
synthetic VCFs (text rendering of trtools_amd.synth).  For every case the input VCF and the
reference's three outputs (.vcf through this repo's VCF writer, .samplog.tab, .loclog.tab)
are stored under tests/golden/dumpstr_synth/."""
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
