"""GPU parity: trk_locus_stats (HIP, through the C ABI) vs the reference's own
outputs (tests/golden/trrecord_vectors.json) and vs the oracle on seeded
synthetic batches.  Integers bit-exact, floats within 1e-9 (BASELINE.json)."""
import math

import numpy as np
import pytest

from helpers import load_golden, unjf, close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def eng():
    from trtools_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def _fetch(res):
    return res.allele_count.get(), res.locus_int.get(), res.locus_f64.get()


def check_against_oracle(orc, L, cnt, li, lf, off, gt, lens, strs, groups, nalleles_thresh, loci=None):
    """Compare device outputs of every (group, locus) with oracle.locus_stats."""
    G = len(groups)
    for l in (range(len(gt)) if loci is None else loci):
        for g in range(G):
            si = groups[g]
            ol = orc.locus_stats(gt[l], lens[l], strs[l], si, use_length=True, nalleles_thresh=nalleles_thresh)
            os_ = orc.locus_stats(gt[l], lens[l], strs[l], si, use_length=False, nalleles_thresh=nalleles_thresh)
            a0, a1 = off[l], off[l + 1]
            assert np.array_equal(cnt[g, a0:a1], ol['index_counts']), (l, g)
            I, F = li[g, l], lf[g, l]
            assert I[L.LI_N_CALLED] == ol['numcalled'] == ol['n_called'], (l, g)
            assert I[L.LI_N_SAMPLES] == ol['n_samples']
            assert I[L.LI_N_ALLELES] == int(ol['index_counts'].sum())
            assert I[L.LI_N_BAD] == 0
            assert I[L.LI_NALLELES_LEN] == ol['nalleles'], (l, g)
            assert I[L.LI_NALLELES_STR] == os_['nalleles'], (l, g)
            for col, o, key in ((L.LF_HET_LEN, ol, 'het'), (L.LF_HET_STR, os_, 'het'),
                                (L.LF_ENTROPY_LEN, ol, 'entropy'), (L.LF_ENTROPY_STR, os_, 'entropy'),
                                (L.LF_THRESH, ol, 'thresh'), (L.LF_MEAN, ol, 'mean'),
                                (L.LF_MODE, ol, 'mode'), (L.LF_VAR, ol, 'var')):
                assert close(F[col], o[key]), (l, g, key, F[col], o[key])
            for scol, fcol, o in ((L.LI_HWE_STATUS_LEN, L.LF_HWEP_LEN, ol),
                                  (L.LI_HWE_STATUS_STR, L.LF_HWEP_STR, os_)):
                if o['hwep_status'] != orc.HWE_OK:
                    assert I[scol] == o['hwep_status'], (l, g, I[scol], o['hwep_status'])
                elif math.isnan(o['hwep']):
                    assert I[scol] == L.HWE_NAN and math.isnan(F[fcol]), (l, g)
                else:
                    assert I[scol] == L.HWE_OK
                    assert close(F[fcol], o['hwep'], 1e-9, 1e-300), (l, g, F[fcol], o['hwep'])


def test_reference_golden_vectors(eng):
    """Every case of trrecord_vectors.json (outputs of the REAL reference)."""
    from trtools_amd import _lib as L
    from trtools_amd.synth import pack_alleles
    cases = load_golden('trrecord_vectors.json')['cases']
    n_checked = 0
    for c in cases:
        gt = np.array(c['gt'], dtype=np.int16)[None, :, :]
        strs = [c['ref']] + list(c['alts'])
        lens = [unjf(x) for x in c['allele_lens']]
        off, lc, sc, cv = pack_alleles([lens], [strs])
        gb = None
        if c['sample_index'] is not None:
            gb = np.array(c['sample_index'], dtype=np.uint8)
        b = eng.make_batch(gt, off, lc, sc, cv, group_bits=gb, n_groups=1)
        res = eng.locus_stats(b, nalleles_thresh=0.1)
        cnt, li, lf = _fetch(res)
        st = c['statstr']
        want_idx = np.zeros(len(lens), dtype=np.int64)
        for k, v in c['counts_idx']:
            want_idx[int(k)] = v
        assert np.array_equal(cnt[0], want_idx), c['kind']
        I, F = li[0, 0], lf[0, 0]
        assert I[L.LI_N_CALLED] == st['numcalled']
        assert I[L.LI_NALLELES_LEN] == st['nalleles_len']
        assert I[L.LI_NALLELES_STR] == st['nalleles_str']
        for col, key in ((L.LF_THRESH, 'thresh'), (L.LF_MEAN, 'mean'), (L.LF_MODE, 'mode'), (L.LF_VAR, 'var'),
                         (L.LF_HET_LEN, 'het_len'), (L.LF_HET_STR, 'het_str'),
                         (L.LF_ENTROPY_LEN, 'entropy_len'), (L.LF_ENTROPY_STR, 'entropy_str')):
            assert close(F[col], unjf(st[key])), (c['kind'], key, F[col], st[key])
        for scol, fcol, key in ((L.LI_HWE_STATUS_LEN, L.LF_HWEP_LEN, 'hwep_len'),
                                (L.LI_HWE_STATUS_STR, L.LF_HWEP_STR, 'hwep_str')):
            h = st[key]
            if 'raises' in h:
                want = L.HWE_VALUE_ERROR if h['raises'] == 'ValueError' else L.HWE_INDEX_ERROR
                assert I[scol] == want, (c['kind'], c['ploidy'], key)
            else:
                v = unjf(h['ok'])
                if math.isnan(v):
                    assert I[scol] == L.HWE_NAN and math.isnan(F[fcol])
                else:
                    assert I[scol] == L.HWE_OK and close(F[fcol], v, 1e-9, 1e-300), (key, F[fcol], v)
        if c['sample_index'] is None:
            assert close(F[L.LF_CALLRATE], unjf(c['callrate']))
        n_checked += 1
        for a in b.arrays.values():
            a.free()
        for a in (res.allele_count, res.locus_int, res.locus_f64):
            a.free()
    assert n_checked == len(cases)


@pytest.mark.parametrize("n_loci,n_samples", [(120, 1000), (40, 50), (30, 1003), (16, 4099)])
def test_synth_batch_vs_oracle(eng, n_loci, n_samples):
    from oracle import trtools_oracle as orc
    from trtools_amd import _lib as L
    from trtools_amd.synth import SynthBatch
    sb = SynthBatch(eng, n_loci, n_samples, seed=20260928 + n_samples)
    host = sb.host_rows(np.arange(n_loci))
    # generator twin: device == numpy, bit for bit
    assert np.array_equal(sb.dev['gt'].get(), host['gt'])
    assert np.array_equal(sb.dev['dp'].get(), host['dp'])
    assert np.array_equal(sb.dev['q'].get().view(np.uint32), host['q'].view(np.uint32))
    res = eng.locus_stats(sb.batch, nalleles_thresh=0.01)
    cnt, li, lf = _fetch(res)
    off = sb.tables[0]
    check_against_oracle(orc, L, cnt, li, lf, off, host['gt'], sb.loci.allele_lens, sb.loci.allele_strs,
                         [None], 0.01)


def _random_batch(rng, n_loci, n_samples, ploidy, max_alt, with_low=False):
    from trtools_amd.synth import pack_alleles
    lens, strs, gts, lp = [], [], [], []
    for l in range(n_loci):
        motif = ''.join(rng.choice(list('ACGT'), size=int(rng.integers(1, 5))))
        A = 1 + int(rng.integers(0, max_alt + 1))
        ss, seen = [], set()
        while len(ss) < A:
            s = motif * int(rng.integers(1, 3 * A + 4))
            if rng.random() < 0.3:
                s = s + 'N' * int(rng.integers(1, 3))
            if rng.random() < 0.1 and ss:
                s = ss[int(rng.integers(0, len(ss)))]     # duplicate sequence
            elif s in seen:
                continue
            seen.add(s)
            ss.append(s)
        strs.append(ss)
        lens.append([len(s) / len(motif) for s in ss])
        pl = ploidy if not with_low else int(rng.integers(1, ploidy + 1))
        lp.append(pl)
        g = rng.integers(0, A, size=(n_samples, ploidy)).astype(np.int16)
        g[rng.random(n_samples) < 0.1, :] = -1
        g[rng.random(n_samples) < 0.05, int(rng.integers(0, ploidy))] = -1
        if pl < ploidy:
            g[:, pl:] = -2
        elif ploidy > 1 and rng.random() < 0.5:
            low = rng.random(n_samples) < 0.1
            g[low, ploidy - 1] = -2
        gts.append(g)
    return np.stack(gts), lens, strs, np.array(lp, dtype=np.uint8), pack_alleles(lens, strs)


@pytest.mark.parametrize("S", [257, 256])     # 256: aligned rows, the streaming kernels with a per-locus ploidy table
@pytest.mark.parametrize("ploidy,n_groups", [(1, 1), (2, 3), (3, 1), (3, 2), (4, 8), (2, 1), (2, 0)])
def test_general_ploidy_and_groups(eng, ploidy, n_groups, S):
    from oracle import trtools_oracle as orc
    from trtools_amd import _lib as L
    rng = np.random.default_rng(100 * ploidy + n_groups)
    n_loci = 25
    gt, lens, strs, lp, (off, lc, sc, cv) = _random_batch(rng, n_loci, S, ploidy, 12, with_low=True)
    gb, groups = None, [None]
    if n_groups:
        gb = rng.integers(0, 1 << n_groups, size=S).astype(np.uint8)
        groups = [((gb >> g) & 1).astype(bool) for g in range(n_groups)]
    b = eng.make_batch(gt, off, lc, sc, cv, locus_ploidy=lp, group_bits=gb, n_groups=max(n_groups, 1))
    res = eng.locus_stats(b, nalleles_thresh=0.05)
    cnt, li, lf = _fetch(res)
    # the oracle sees what the reference would see: only the locus's own ploidy columns
    gt_view = [gt[l][:, :lp[l]] for l in range(n_loci)]
    check_against_oracle(orc, L, cnt, li, lf, off, gt_view, lens, strs, groups, 0.05)


def test_many_alleles_lds_and_direct_paths(eng):
    """A > 255 (uint8 would overflow), A beyond the LDS histogram (global-atomic path)."""
    from oracle import trtools_oracle as orc
    from trtools_amd import _lib as L
    from trtools_amd.synth import pack_alleles
    rng = np.random.default_rng(5)
    S = 3000
    lens, strs, gts = [], [], []
    for A in (300, 1500, 5000, 2):
        ss = ['AC' * (i + 1) for i in range(A)]
        strs.append(ss)
        lens.append([len(s) / 2 for s in ss])
        g = rng.integers(0, A, size=(S, 2)).astype(np.int16)
        g[rng.random(S) < 0.05, :] = -1
        gts.append(g)
    gt = np.stack(gts)
    off, lc, sc, cv = pack_alleles(lens, strs)
    b = eng.make_batch(gt, off, lc, sc, cv)
    res = eng.locus_stats(b)
    cnt, li, lf = _fetch(res)
    check_against_oracle(orc, L, cnt, li, lf, off, gt, lens, strs, [None], 0.01)


def test_empty_and_tiny_batches(eng):
    from trtools_amd import _lib as L
    from trtools_amd.synth import pack_alleles
    off, lc, sc, cv = pack_alleles([[3.0]], [['CAGCAGCAG']])
    # one sample, no call (reference: get_nocall_record)
    b = eng.make_batch(np.array([[[-1, -1]]], dtype=np.int16), off, lc, sc, cv)
    cnt, li, lf = _fetch(eng.locus_stats(b))
    assert cnt.tolist() == [[0]] and li[0, 0, L.LI_N_CALLED] == 0
    assert math.isnan(lf[0, 0, L.LF_HET_LEN]) and math.isnan(lf[0, 0, L.LF_THRESH])
    assert li[0, 0, L.LI_HWE_STATUS_LEN] == L.HWE_NAN
    # zero loci
    b0 = eng.make_batch(np.zeros((0, 5, 2), dtype=np.int16), np.zeros(1, dtype=np.int32),
                        np.zeros(0, dtype=np.uint16), np.zeros(0, dtype=np.uint16), np.zeros(0))
    res0 = eng.locus_stats(b0)
    assert res0.locus_int.get().shape == (1, 0, L.TRK_LI_COLS)
    # out-of-range allele index is reported, not counted (reference: IndexError :1242)
    b2 = eng.make_batch(np.array([[[0, 7], [0, 0]]], dtype=np.int16), off, lc, sc, cv)
    cnt, li, lf = _fetch(eng.locus_stats(b2))
    assert li[0, 0, L.LI_N_BAD] == 1 and cnt.tolist() == [[3]]


def test_rccl_single_rank_collectives(eng):
    """The RCCL path (dlopen'ed librccl, ncclCommInitRank) on a 1-rank communicator."""
    uid = eng.comm_unique_id()
    assert len(uid) == 128
    eng.comm_init(0, 1, uid)
    a = eng.upload(np.arange(1000, dtype=np.int64))
    eng.allreduce_sum_i64(a)
    assert np.array_equal(a.get(), np.arange(1000, dtype=np.int64))
    src = eng.upload(np.arange(256, dtype=np.uint8))
    dst = eng.zeros((1, 256), np.uint8)
    eng.allgather(src, dst)
    assert np.array_equal(dst.get()[0], np.arange(256, dtype=np.uint8))


def test_sentinel_mixes_on_the_streaming_kernel(eng):
    """Diploid batch without a ploidy table (the streaming kernel): haploid calls (a,-2),
    partial calls (a,-1), (-1,-2) / (-2,-1) / (-2,-2) / (-1,-1) rows, duplicate classes."""
    from oracle import trtools_oracle as orc
    from trtools_amd import _lib as L
    rng = np.random.default_rng(21)
    n_loci, S = 40, 512
    gt, lens, strs, _, (off, lc, sc, cv) = _random_batch(rng, n_loci, S, 2, 9)
    for l in range(n_loci):
        r = rng.random(S)
        gt[l][r < 0.06, 1] = -2
        gt[l][(r >= 0.06) & (r < 0.09), 0] = -2
        gt[l][(r >= 0.09) & (r < 0.11)] = -2
        gt[l][(r >= 0.11) & (r < 0.13)] = (-1, -2)
        gt[l][(r >= 0.13) & (r < 0.15)] = (-2, -1)
    gt[5] = -1
    gt[6] = -2
    b = eng.make_batch(gt, off, lc, sc, cv)
    cnt, li, lf = _fetch(eng.locus_stats(b, nalleles_thresh=0.02))
    check_against_oracle(orc, L, cnt, li, lf, off, gt, lens, strs, [None], 0.02)


@pytest.mark.parametrize("S", [4, 60, 64, 68, 252, 1000, 2048, 4096, 5000])
def test_loci_per_wave_variants_agree(eng, S):
    """k_locus_count_v2 (one locus per wave) vs k_locus_count_v3<2> / <4> (two / four loci side by side in a wave,
    the short-row kernel): identical counts and row predicates on rows with every sentinel mix, duplicate classes,
    a locus count that leaves the last wave partly empty, and allele counts that differ inside a wave; the
    one-locus-per-wave result is checked against the oracle."""
    import os
    from oracle import trtools_oracle as orc
    from trtools_amd import _lib as L
    rng = np.random.default_rng(100 + S)
    n_loci = 23
    gt, lens, strs, _, (off, lc, sc, cv) = _random_batch(rng, n_loci, S, 2, 14)
    for l in range(n_loci):
        r = rng.random(S)
        gt[l][r < 0.05, 1] = -2
        gt[l][(r >= 0.05) & (r < 0.08)] = (-1, -2)
        gt[l][(r >= 0.08) & (r < 0.10)] = (-2, -1)
        gt[l][(r >= 0.10) & (r < 0.12)] = -2
        gt[l][(r >= 0.12) & (r < 0.16)] = -1
        gt[l][(r >= 0.16) & (r < 0.19), 1] = -1
    gt[3] = -1
    gt[4] = -2
    b = eng.make_batch(gt, off, lc, sc, cv)
    got = {}
    for rr in ('1', '2', '4'):
        L.set_option('TRK_CNT_R', rr)
        try:
            got[rr] = _fetch(eng.locus_stats(b, nalleles_thresh=0.02))
        finally:
            L.set_option('TRK_CNT_R', None)
    check_against_oracle(orc, L, *got['1'], off, gt, lens, strs, [None], 0.02)
    for rr in ('2', '4'):
        assert np.array_equal(got[rr][0], got['1'][0]), rr
        assert np.array_equal(got[rr][1], got['1'][1]), rr
        assert np.array_equal(got[rr][2], got['1'][2], equal_nan=True), rr


@pytest.mark.parametrize("S,ploidy,groups", [(1000, 2, 0), (8000, 2, 0), (1003, 2, 0), (512, 3, 0), (1000, 2, 2)])
def test_twin_count_outputs(eng, S, ploidy, groups):
    """TRK_STATS_TWIN: the second copy of allele_count / locus_int equals the first on the streaming kernels (which
    store it in the same pass) and on the general paths (copied after the kernel)."""
    rng = np.random.default_rng(S + ploidy)
    gt, lens, strs, _, (off, lc, sc, cv) = _random_batch(rng, 37, S, ploidy, 9)
    gb = None
    if groups:
        gb = rng.integers(0, 1 << groups, size=S).astype(np.uint8)
    b = eng.make_batch(gt, off, lc, sc, cv, group_bits=gb, n_groups=max(groups, 1))
    plain = eng.locus_stats(b, count_only=True)
    out = eng.alloc_stats(b, twin=True)
    eng.locus_stats(b, out=out, count_only=True)
    for res in (out, out.twin):
        assert np.array_equal(res.allele_count.get(), plain.allele_count.get())
        assert np.array_equal(res.locus_int.get()[..., :6], plain.locus_int.get()[..., :6])
        assert np.array_equal(res.locus_int.get()[..., 8], plain.locus_int.get()[..., 8])


def test_two_queues_give_the_same_statistics(eng):
    """trk_stream_select / trk_stream_wait: the finaliser on queue 1 beside a call-filter pass on queue 0 (what
    bench.py overlaps) returns what the single-queue sequence returns."""
    from trtools_amd import _lib as L
    from trtools_amd.synth import SynthBatch
    sb = SynthBatch(eng, 300, 2000, seed=99, planes=('dp', 'q'))
    want = eng.locus_stats(sb.batch)
    filters = [dict(op=L.F_LT, plane_a=0, thr=10), dict(op=L.F_LT, plane_a=1, thr=0.9)]
    planes = [sb.dev['dp'], sb.dev['q']]
    ref_b = eng.locus_stats(sb.batch, count_only=True)
    ref_call = eng.call_filters(sb.batch, planes, filters, dp_plane=0, delta_stats=ref_b)
    eng.locus_finalize(sb.batch, ref_b)
    for _ in range(3):
        a = eng.locus_stats(sb.batch, count_only=True)
        b = eng.alloc_stats(sb.batch)
        b.allele_count.copy_from(a.allele_count)
        b.locus_int.copy_from(a.locus_int)
        eng.queue_wait(1, 0)
        with eng.on_queue(1):
            eng.locus_finalize(sb.batch, a)
        res = eng.call_filters(sb.batch, planes, filters, dp_plane=0, delta_stats=b)
        eng.locus_finalize(sb.batch, b)
        eng.queue_wait(0, 1)
        eng.sync()
        assert np.array_equal(a.locus_int.get(), want.locus_int.get())
        assert np.array_equal(a.locus_f64.get(), want.locus_f64.get(), equal_nan=True)
        assert np.array_equal(b.locus_int.get(), ref_b.locus_int.get())
        assert np.array_equal(b.locus_f64.get(), ref_b.locus_f64.get(), equal_nan=True)
        assert np.array_equal(res.filter_mask.get(), ref_call.filter_mask.get())
    with pytest.raises(Exception):
        eng.queue_wait(0, 7)


@pytest.mark.parametrize("n_groups,S,layout", [(1, 1000, 'subset'), (2, 1000, 'disjoint'), (2, 1004, 'overlap'),
                                                (3, 2000, 'overlap'), (3, 64, 'disjoint')])
def test_sample_groups_on_the_streaming_kernel(eng, n_groups, S, layout):
    """statSTR --samples (statSTR.py:520-542): diploid batches with S % 4 == 0 take k_locus_count_v2g (one histogram
    per class of group bits).  Groups may overlap, samples may be in no group; sentinel mixes, out-of-class
    duplicates and an all-missing locus included."""
    from oracle import trtools_oracle as orc
    from trtools_amd import _lib as L
    rng = np.random.default_rng(1000 * n_groups + S)
    n_loci = 30
    gt, lens, strs, lp, (off, lc, sc, cv) = _random_batch(rng, n_loci, S, 2, 12)
    gt[3] = -1                       # no call at all
    gt[4, :, 1] = -2                 # haploid calls in a diploid tensor
    gt[5, : S // 2] = [-1, -2]
    gt[6, : S // 3] = [-2, -1]
    gt[7, : S // 4] = [-2, -2]
    if layout == 'subset':
        gb = (rng.random(S) < 0.6).astype(np.uint8)
    elif layout == 'disjoint':
        gb = (np.uint8(1) << rng.integers(0, n_groups, size=S).astype(np.uint8)).astype(np.uint8)
        gb[rng.random(S) < 0.1] = 0
    else:
        gb = rng.integers(0, 1 << n_groups, size=S).astype(np.uint8)
    gb |= (rng.integers(0, 2, size=S).astype(np.uint8) << np.uint8(7))      # bits above n_groups are ignored
    groups = [((gb >> g) & 1).astype(bool) for g in range(n_groups)]
    b = eng.make_batch(gt, off, lc, sc, cv, group_bits=gb, n_groups=n_groups)
    res = eng.locus_stats(b, nalleles_thresh=0.05)
    cnt, li, lf = _fetch(res)
    check_against_oracle(orc, L, cnt, li, lf, off, [gt[l] for l in range(n_loci)], lens, strs, groups, 0.05)
    for g in range(n_groups):
        assert np.all(li[g][:, L.LI_N_SAMPLES] == int(groups[g].sum()))
    # the per-call kernel (TRK_CNT_NOGROUPFAST) gives the same integers
    import os
    L.set_option('TRK_CNT_NOGROUPFAST', '1')
    try:
        ref = eng.locus_stats(b, nalleles_thresh=0.05)
    finally:
        L.set_option('TRK_CNT_NOGROUPFAST', None)
    cols = [L.LI_N_CALLED, L.LI_N_LOWPLOIDY, L.LI_N_HOM_LEN, L.LI_N_HOM_STR, L.LI_N_SAMPLES]
    assert np.array_equal(ref.allele_count.get(), res.allele_count.get())
    assert np.array_equal(ref.locus_int.get()[:, :, cols], res.locus_int.get()[:, :, cols])


@pytest.mark.parametrize("n_groups,S,layout", [(1, 1000, 'subset'), (2, 1003, 'overlap'), (3, 64, 'disjoint'),
                                                (5, 2501, 'overlap'), (8, 4000, 'disjoint'), (8, 997, 'overlap'),
                                                (4, 37, 'empty_group')])
def test_sample_groups_as_class_column_ranges(eng, n_groups, S, layout):
    """trk_batch.class_runs (DeviceBatch.sorted_by_class): the columns gathered into class order on the device, each
    class counted by the ungrouped streaming kernel through a column-range view, classes added into groups.  Any
    sample count, up to 8 overlapping groups, samples in no group, a group without samples; against the oracle and
    bit for bit against the per-call group kernel on the unsorted batch (incl. the float columns: same finaliser)."""
    from oracle import trtools_oracle as orc
    from trtools_amd import _lib as L
    rng = np.random.default_rng(77 * n_groups + S)
    n_loci = 26
    gt, lens, strs, lp, (off, lc, sc, cv) = _random_batch(rng, n_loci, S, 2, 12)
    gt[3] = -1
    gt[4, :, 1] = -2
    gt[5, : S // 2] = [-1, -2]
    gt[6, : S // 3] = [-2, -1]
    gt[7, : S // 4] = [-2, -2]
    if layout == 'subset':
        gb = (rng.random(S) < 0.6).astype(np.uint8)
    elif layout == 'disjoint':
        gb = (np.uint8(1) << rng.integers(0, n_groups, size=S).astype(np.uint8)).astype(np.uint8)
        gb[rng.random(S) < 0.1] = 0
    elif layout == 'empty_group':
        gb = rng.integers(0, 1 << n_groups, size=S).astype(np.uint8) & np.uint8(0b1011)   # group 2 has nobody
    else:
        gb = rng.integers(0, 1 << n_groups, size=S).astype(np.uint8)
    groups = [((gb >> g) & 1).astype(bool) for g in range(n_groups)]
    plain = eng.make_batch(gt, off, lc, sc, cv)
    b = plain.sorted_by_class(eng, gb, n_groups)
    assert getattr(b, 'class_sorted', False) and b.struct.n_class_runs >= 1
    res = eng.locus_stats(b, nalleles_thresh=0.05)
    cnt, li, lf = _fetch(res)
    if layout != 'empty_group':      # (the oracle raises for a group without samples as the reference does)
        check_against_oracle(orc, L, cnt, li, lf, off, [gt[l] for l in range(n_loci)], lens, strs, groups, 0.05)
    for g in range(n_groups):
        assert np.all(li[g][:, L.LI_N_SAMPLES] == int(groups[g].sum()))
    ref = eng.locus_stats(plain.with_groups(eng, gb, n_groups), nalleles_thresh=0.05)
    assert np.array_equal(ref.allele_count.get(), cnt)
    assert np.array_equal(ref.locus_int.get(), li)
    assert np.array_equal(ref.locus_f64.get(), lf, equal_nan=True)
    # twin outputs (dumpSTR's copy of the counts) come out of the combine step too
    tw = eng.alloc_stats(b, twin=True)
    eng.locus_stats(b, out=tw, count_only=True)
    cols = [L.LI_N_CALLED, L.LI_N_LOWPLOIDY, L.LI_N_HOM_LEN, L.LI_N_HOM_STR, L.LI_N_BAD, L.LI_N_SAMPLES]
    assert np.array_equal(tw.allele_count.get(), cnt) and np.array_equal(tw.twin.allele_count.get(), cnt)
    assert np.array_equal(tw.twin.locus_int.get()[:, :, cols], li[:, :, cols])


def test_column_range_view_is_rejected_outside_the_count_entry(eng):
    """trk_batch.row_stride is honoured by trk_locus_stats' streaming kernels only; the other entries refuse it."""
    from trtools_amd import _lib as L
    import ctypes as C
    rng = np.random.default_rng(5)
    gt, lens, strs, lp, (off, lc, sc, cv) = _random_batch(rng, 8, 64, 2, 6)
    b = eng.make_batch(gt, off, lc, sc, cv)
    # a view of columns 16..47: equal to the statistics of the sliced tensor
    s = L.Batch()
    C.memmove(C.byref(s), C.byref(b.struct), C.sizeof(L.Batch))
    s.gt = b.arrays['gt'].ptr + 16 * 2 * 2
    s.n_samples, s.row_stride = 32, 64
    from trtools_amd.engine import DeviceBatch
    view = DeviceBatch(s, b.arrays, 1, b.sum_alleles)
    res = eng.locus_stats(view, nalleles_thresh=0.05)
    ref = eng.locus_stats(eng.make_batch(np.ascontiguousarray(gt[:, 16:48]), off, lc, sc, cv), nalleles_thresh=0.05)
    assert np.array_equal(res.allele_count.get(), ref.allele_count.get())
    assert np.array_equal(res.locus_int.get(), ref.locus_int.get())
    assert np.array_equal(res.locus_f64.get(), ref.locus_f64.get(), equal_nan=True)
    with pytest.raises(L.TrkError):
        eng.call_filters(view, [], [])
    s.row_stride = 62
    with pytest.raises(L.TrkError):
        eng.locus_stats(DeviceBatch(s, b.arrays, 1, b.sum_alleles))
