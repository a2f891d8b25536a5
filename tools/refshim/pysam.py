"""Build-container-only stand-in for ``pysam`` (TabixFile / asBed), enough for
the reference's BED region filter (dumpSTR/filters.py:266-292) when generating
golden vectors.  Linear scan of a (b)gzipped BED."""
import gzip


def asBed():
    return None


class TabixFile:
    def __init__(self, filename, parser=None):
        self.rows = []
        with gzip.open(filename, 'rt') as fh:
            for line in fh:
                if not line.strip() or line.startswith('#'):
                    continue
                f = line.rstrip('\n').split('\t')
                self.rows.append((f[0], int(f[1]), int(f[2])))
        self.contigs = {r[0] for r in self.rows}

    def fetch(self, region=None, multiple_iterators=False):
        chrom, rng = region.split(':')
        a, b = rng.split('-')
        start, end = int(float(a)) - 1, int(float(b))
        if chrom not in self.contigs:
            raise ValueError("could not create iterator for region '%s'" % region)
        return iter([r for r in self.rows if r[0] == chrom and r[1] < end and r[2] > start])
