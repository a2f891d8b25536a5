"""Allele-frequency bar plots of statSTR --plot-afreq (reference
trtools/statSTR/statSTR.py:31-80); at most ten loci, host matplotlib, the
allele frequencies themselves come from the device via TRRecord.GetAlleleFreqs."""
import numpy as np


def PlotAlleleFreqs(trrecord, outprefix, sample_indexes=None, sampleprefixes=None):
    import matplotlib
    matplotlib.use('Agg')
    import matplotlib.pyplot as plt
    sample_indexes = sample_indexes or [None]
    sampleprefixes = sampleprefixes or []
    freqs = [trrecord.GetAlleleFreqs(sample_index=si, uselength=True) for si in sample_indexes]
    lengths = sorted({k for f in freqs for k in f})
    if not lengths:
        return
    xs = np.arange(int(np.floor(min(lengths))) - 2, int(np.ceil(max(lengths))) + 3)
    width = 0.9 / len(freqs)
    fig, ax = plt.subplots()
    for i, f in enumerate(freqs):
        label = sampleprefixes[i] if i < len(sampleprefixes) else None
        ax.bar([k + i * width for k in f], [f[k] for k in f], width=width, label=label)
    ax.set_xticks(xs)
    ax.set_xlabel("TR allele length (repeat units)")
    ax.set_ylabel("Frequency")
    if sampleprefixes:
        ax.legend()
    fig.savefig("%s-%s-%s.pdf" % (outprefix, trrecord.chrom, trrecord.pos))
    plt.close(fig)
