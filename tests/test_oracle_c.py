"""Pin the C half of the oracle against the numpy/scipy oracle (which is pinned
against the real reference).  CPU only."""
import collections
import math

import numpy as np

from oracle import trtools_oracle as orc
from oracle import oracle_c
from helpers import load_golden, unjf, close


def test_c_binomtest_matches_scipy_vectors():
    lib = oracle_c.load()
    for k, n, p, pv in load_golden('binomtest_vectors.json')['cases']:
        got = lib.orc_binomtest(k, n, p)
        assert close(got, unjf(pv), 1e-9, 1e-300), (k, n, p, got, pv)


def test_c_batch_stats_match_numpy_oracle():
    from trtools_amd.synth import make_loci, cells_numpy, pack_alleles
    for S, Lc in ((400, 60), (33, 40)):
        loci = make_loci(Lc, S, seed=5 + S)
        h = cells_numpy(5, loci, np.arange(Lc), S)
        off, lc, sc, cv = pack_alleles(loci.allele_lens, loci.allele_strs)
        cnt, oi, of = oracle_c.batch_stats(h['gt'], None, off, lc, sc, cv)
        for l in range(Lc):
            ol = orc.locus_stats(h['gt'][l], loci.allele_lens[l], loci.allele_strs[l], None, True)
            os_ = orc.locus_stats(h['gt'][l], loci.allele_lens[l], loci.allele_strs[l], None, False)
            assert np.array_equal(cnt[off[l]:off[l + 1]], ol['index_counts'])
            assert oi[l, 0] == ol['numcalled']
            for j, (o, key) in enumerate(((ol, 'thresh'), (ol, 'mean'), (ol, 'mode'), (ol, 'var'), (ol, 'het'),
                                          (os_, 'het'), (ol, 'entropy'), (os_, 'entropy'), (ol, 'hwep'),
                                          (os_, 'hwep'))):
                if key == 'hwep' and o['hwep_status'] != orc.HWE_OK:
                    assert oi[l, 5 if o is ol else 6] == o['hwep_status']
                    continue
                assert close(of[l, j], o[key], 1e-9, 1e-300), (l, key, of[l, j], o[key])


def test_c_call_filters_match_numpy_oracle():
    from trtools_amd.synth import make_loci, cells_numpy
    S, Lc = 120, 50
    loci = make_loci(Lc, S, seed=9)
    h = cells_numpy(9, loci, np.arange(Lc), S)
    gout, mask, counters, totaldp, dpmiss = oracle_c.call_filters_dpq(h['gt'], h['dp'], h['q'], 10, 55, 0.9)
    info = collections.OrderedDict([('numcalls', np.zeros(S, dtype=int)), ('totaldp', np.zeros(S)),
                                    ('a', np.zeros(S, dtype=int)), ('b', np.zeros(S, dtype=int)),
                                    ('c', np.zeros(S, dtype=int))])
    for l in range(Lc):
        dp, q = h['dp'][l].reshape(-1, 1), h['q'][l].reshape(-1, 1)
        outs = [('a', orc.filt_min_value(dp, 10)), ('b', orc.filt_max_value(dp, 55)), ('c', orc.filt_min_value(q, 0.9))]
        g2, _ = orc.apply_call_filters(h['gt'][l], outs, info, dp=dp)
        assert np.array_equal(g2, gout[l])
    assert np.array_equal(counters[0], info['numcalls'])
    for k, n in enumerate('abc'):
        assert np.array_equal(counters[1 + k], info[n])
    tot = totaldp.astype(float)
    tot[dpmiss > 0] = np.nan
    assert np.array_equal(np.isnan(tot), np.isnan(info['totaldp']))
    assert np.array_equal(tot[~np.isnan(tot)], info['totaldp'][~np.isnan(tot)])
