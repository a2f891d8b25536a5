#!/usr/bin/env python3
"""Record serialiser microbenchmark (trk_vcf_format_samples): one S-sample record, columns GT / +DP / +Q; run with
TRK_FMT_THREADS=1 for the serial path."""
import os as _os; _os.environ.setdefault('TRK_LAB', '1')   # a tool: lab knobs are honoured (trtools_amd/_knobs.py)
import sys, time, numpy as np, ctypes, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trtools_amd import vcfio
lib, Column = vcfio._serializer()
rng=np.random.default_rng(0)
S=int(os.environ.get('S',5000))
gt = rng.integers(0, 5, size=(S, 3)).astype(np.int16)
dp = rng.integers(0, 60, size=(S, 1)).astype(np.int32)
q = np.round(rng.random((S,1)),2).astype(np.float32)
cols=(Column*3)(Column(0,3,0,0,gt.ctypes.data),Column(1,1,0,0,dp.ctypes.data),Column(2,1,0,0,q.ctypes.data))
buf=ctypes.create_string_buffer(S*100)
for nc,name in ((1,'GT'),(2,'GT+DP'),(3,'GT+DP+Q')):
    for _ in range(20): lib.trk_vcf_format_samples(S,nc,cols,buf,S*100)
    t=time.time()
    for _ in range(300): n=lib.trk_vcf_format_samples(S,nc,cols,buf,S*100)
    dt=(time.time()-t)/300
    print("threads=%s %s: %.0f us" % (os.environ.get('TRK_FMT_THREADS','default'), name, dt*1e6))
