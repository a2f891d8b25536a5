#!/usr/bin/env python3
"""k_locus_count across row lengths x loci-per-wave (TRK_CNT_R) x loads in flight (TRK_CNT_U), inputs resident;
kernel time from the library's HIP-event brackets.  `gpurun -- python tools/count_probe.py`."""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from trtools_amd.engine import Engine
from trtools_amd.synth import SynthBatch, make_loci


def tiled(base, reps):
    loci = copy.copy(base)
    loci.motifs = base.motifs * reps
    loci.allele_strs = base.allele_strs * reps
    loci.allele_lens = base.allele_lens * reps
    nA = int(base.allele_off[-1])
    loci.allele_off = np.concatenate([base.allele_off[:-1] + r * nA for r in range(reps)] + [np.array([reps * nA])]).astype(np.int32)
    loci.cdf24 = np.tile(base.cdf24, reps)
    loci.miss_thr16 = np.tile(base.miss_thr16, reps)
    loci.inbreed_thr16 = np.tile(base.inbreed_thr16, reps)
    return loci


eng = Engine(0)
if '--bench-data' in sys.argv:      # the bench's own cohort (100k x 10k, seed 20260931) and its 12.5k-locus shard
    loci = make_loci(100000, 10000, 20260931)
    for n in (100000, 12500):
        sb = SynthBatch(eng, n, 10000, seed=20260931, planes=(), loci=loci.slice(0, n))
        for twin in (False, True):
            res = eng.alloc_stats(sb.batch, twin=twin)
            for R in ('1', '2', '4'):
                for U in ('1', '2', '4'):
                    os.environ['TRK_CNT_R'], os.environ['TRK_CNT_U'] = R, U
                    eng.profile(True)
                    for it in range(13):
                        if it == 3:
                            eng.sync(); eng.profile_reset()
                        eng.locus_stats(sb.batch, out=res, count_only=True)
                    eng.sync()
                    k, ms = eng.profile_get()['k_locus_count']
                    eng.profile(False)
                    print("bench data %6d x 10000 twin=%d R=%s U=%s  %.4f ms (%.2f of peak)" % (
                        n, twin, R, U, ms / k, n * 1e4 * 4 / (ms / k * 1e-3) / 8e12), flush=True)
        for a in list(sb.dev.values()) + list(sb.batch.arrays.values()):
            a.free()
    sys.exit(0)
shapes = [(10000, 1000, 1), (400000, 1000, 40), (1600000, 252, 160), (200000, 2000, 20), (100000, 4000, 10), (50000, 5000, 5),
          (100000, 10000, 10)]
if len(sys.argv) > 1:
    shapes = [s for s in shapes if str(s[1]) in sys.argv[1:]]
for Lc, S, reps in shapes:
    base = make_loci(Lc // reps, S, 20260929)
    sb = SynthBatch(eng, Lc, S, seed=20260929, planes=(), loci=tiled(base, reps) if reps > 1 else base)
    res = eng.alloc_stats(sb.batch)
    ref = None
    for R in ('1', '2', '4'):
        for U in ('2', '4'):
            os.environ['TRK_CNT_R'], os.environ['TRK_CNT_U'] = R, U
            eng.profile(True)
            for it in range(8):
                if it == 3:
                    eng.sync(); eng.profile_reset()
                eng.locus_stats(sb.batch, out=res, count_only=True)
            eng.sync()
            n, ms = eng.profile_get()['k_locus_count']
            eng.profile(False)
            got = (res.allele_count.get(), res.locus_int.get()[..., :6])
            same = True if ref is None else all(np.array_equal(a, b) for a, b in zip(ref, got))
            ref = ref or got
            print("%8d x %6d  maxA %2d  R=%s U=%s  %.4f ms  %.0f GB/s (%.2f of peak)  same=%s" % (
                Lc, S, sb.batch.struct.max_alleles, R, U, ms / n, Lc * S * 4 / (ms / n * 1e-3) / 1e9,
                Lc * S * 4 / (ms / n * 1e-3) / 8e12, same), flush=True)
    del os.environ['TRK_CNT_R'], os.environ['TRK_CNT_U']
    for a in [res.allele_count, res.locus_int, res.locus_f64] + list(sb.dev.values()) + list(sb.batch.arrays.values()):
        a.free()
