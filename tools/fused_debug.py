import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from trtools_amd.engine import Engine
from test_gpu_stats import _random_batch
from test_gpu_fused_stats import _both
eng = Engine(0)
S, max_alt = 252, 3
rng = np.random.default_rng(1000 * max_alt + S)
n_loci = 45
gt, lens, strs, _, (off, lc, sc, cv) = _random_batch(rng, n_loci, S, 2, max_alt)
for l in range(n_loci):
    r = rng.random(S)
    if l % 3 == 0:
        gt[l][r < 0.05, 1] = -2
        gt[l][(r >= 0.10) & (r < 0.12)] = -2
    gt[l][(r >= 0.05) & (r < 0.08)] = (-1, -2)
    gt[l][(r >= 0.12) & (r < 0.16)] = -1
    gt[l][(r >= 0.16) & (r < 0.19), 1] = -1
gt[3] = -1; gt[4] = -2; gt[7] = 0; gt[8][:, 0] = 0
b = eng.make_batch(gt, off, lc, sc, cv)
chain, fused = _both(eng, b, 0.02)
for l in (8, 41):
    a = chain[0][0][off[l]:off[l+1]]
    print(l, 'counts', a, 'lc', lc[off[l]:off[l+1]], 'sc', sc[off[l]:off[l+1]], strs[l], lens[l])
    print(' chain', chain[2][0][l][4:8], [hex(x) for x in chain[2][0][l][4:8].view(np.uint64)])
    print(' fused', fused[2][0][l][4:8], [hex(x) for x in fused[2][0][l][4:8].view(np.uint64)])
    A = len(a)
    for name, cls in (('len', lc), ('str', sc)):
        cc = np.zeros(A, dtype=np.int64)
        for i in range(A): cc[cls[off[l]+i]] += a[i]
        ft = float(a.sum()); fsum = 0.0
        for n in cc:
            if n: fsum += float(n)/ft
        ent = 0.0
        for n in cc:
            if n:
                pk = (float(n)/ft)/fsum
                ent -= pk*np.log(pk)
        ent /= 0.693147180559945309417232
        print('  numpy', name, cc, ent, hex(np.float64(ent).view(np.uint64)), 'fsum', fsum.hex())
