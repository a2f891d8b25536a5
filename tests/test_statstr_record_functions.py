"""statSTR's per-record statistic functions (GetThresh, GetAFreq, GetNAlleles, GetHWEP, GetHet, GetEntropy, GetMean,
GetMode, GetVariance, GetNumSamples: reference statSTR/statSTR.py:104-431) against the REAL reference's outputs for the
210 records of tests/golden/trrecord_vectors.json (tools/gen_golden.py calls the reference's functions of the same
names; the `statstr` entry of a case holds what they returned).  On the CPU the record's histogram comes through the
oracle seam (tests/oracle_compute.py), on the GPU from the device (TRRecord._device_stats: a one-locus batch)."""
import math
import types
import warnings

import numpy as np
import pytest

from helpers import load_golden, unjf, close


class _Variant:
    """What TRRecord reads of a cyvcf2.Variant, for a record given as arrays."""

    def __init__(self, gt, ref, alts):
        g = np.asarray(gt, dtype=np.int16)
        self.CHROM, self.POS, self.ID, self.REF, self.ALT = 'chrT', 1000, 'rec', ref, list(alts)
        self.INFO, self.FORMAT = {}, ['GT']
        self._g = np.concatenate([g, np.zeros((g.shape[0], 1), dtype=np.int16)], axis=1)
        self.genotype = types.SimpleNamespace(array=lambda: self._g, n_samples=g.shape[0])
        self.ploidy = g.shape[1]

    def format(self, key):
        raise KeyError(key)


def _checks(compute):
    from trtools_amd import runtime
    from trtools_amd.statSTR import statSTR
    from trtools_amd.utils import tr_harmonizer as trh
    old = runtime.set_compute(compute)
    n = 0
    try:
        for c in load_golden('trrecord_vectors.json')['cases']:
            rec = trh.TRRecord(_Variant(c['gt'], c['ref'], c['alts']), c['ref'], list(c['alts']), c['motif'], 'id', None)
            si = None if c['sample_index'] is None else np.array(c['sample_index'], dtype=bool)
            sis, st, tag = [si], c['statstr'], (c['kind'], c['ploidy'], len(c['gt']))
            assert close(statSTR.GetThresh(rec, sis)[0], unjf(st['thresh'])), tag
            for ul in (True, False):
                t = 'len' if ul else 'str'
                with warnings.catch_warnings():
                    warnings.simplefilter('ignore')
                    assert statSTR.GetAFreq(rec, sis, uselength=ul)[0] == st['afreq_' + t], tag
                    assert statSTR.GetAFreq(rec, sis, uselength=ul, count=True)[0] == st['acount_' + t], tag
                    assert statSTR.GetNAlleles(rec, sis, nalleles_thresh=0.1, uselength=ul)[0] == st['nalleles_' + t], tag
                    assert close(statSTR.GetHet(rec, sis, uselength=ul)[0], unjf(st['het_' + t])), tag
                    assert close(statSTR.GetEntropy(rec, sis, uselength=ul)[0], unjf(st['entropy_' + t])), tag
                    want = st['hwep_' + t]
                    if 'raises' in want:
                        with pytest.raises({'ValueError': ValueError, 'IndexError': IndexError}[want['raises']]):
                            statSTR.GetHWEP(rec, sis, uselength=ul)
                    else:
                        got = statSTR.GetHWEP(rec, sis, uselength=ul)[0]
                        w = unjf(want['ok'])
                        assert (math.isnan(got) and math.isnan(w)) or close(got, w, 1e-9, 1e-300), (tag, got, w)
            assert close(statSTR.GetMean(rec, sis)[0], unjf(st['mean'])), tag
            assert close(statSTR.GetMode(rec, sis)[0], unjf(st['mode'])), tag
            assert close(statSTR.GetVariance(rec, sis)[0], unjf(st['var'])), tag
            assert statSTR.GetNumSamples(rec, sis)[0] == st['numcalled'], tag
            # several groups in one call: one value per entry, in order
            if si is not None:
                both = statSTR.GetHet(rec, [None, si])
                assert len(both) == 2 and close(both[1], unjf(st['het_len']))
            n += 1
    finally:
        runtime.set_compute(old)
    assert n == 210


def test_record_functions_through_the_oracle_seam():
    from oracle_compute import OracleCompute
    _checks(OracleCompute())


@pytest.mark.gpu
def test_record_functions_on_the_device():
    from trtools_amd.compute import DeviceCompute
    _checks(DeviceCompute())
