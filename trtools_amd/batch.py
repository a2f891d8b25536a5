"""Packing of harmonised TR records into the ragged device batch of include/trk.h.

A ``HostBatch`` is the host-side (numpy) image of ``trk_batch``: the genotype
index tensor ``gt [L, S, P]`` int16 (cyvcf2 sentinels, phase column dropped),
per-locus ploidy, and the per-allele class tables derived from what
``TRRecord.__init__`` precomputes per record (reference
tr_harmonizer.py:693-773).  FORMAT planes for the dumpSTR call filters are
stacked the same way (``[L, S, k]`` int32 / float32).
"""
import numpy as np

from .synth import pack_alleles

INT_MISSING = -2147483648


class HostBatch:
    def __init__(self, gt, locus_ploidy, allele_lens, allele_strs, group_bits=None, n_groups=1):
        self.gt = np.ascontiguousarray(gt, dtype=np.int16)
        self.n_loci, self.n_samples, self.ploidy = self.gt.shape
        self.locus_ploidy = np.ascontiguousarray(locus_ploidy, dtype=np.uint8)
        self._lens, self._strs, self._lists = allele_lens, allele_strs, None
        self.allele_off, self.len_class, self.str_class, self.len_class_value = \
            pack_alleles(allele_lens, allele_strs)
        self.max_alleles = int(np.max(np.diff(self.allele_off))) if self.n_loci else 0
        self.group_bits = None if group_bits is None else np.ascontiguousarray(group_bits, dtype=np.uint8)
        self.n_groups = n_groups if group_bits is not None else 1

    @classmethod
    def from_tables(cls, gt, locus_ploidy, allele_off, len_class, str_class, len_class_value, group_bits=None,
                    n_groups=1, lists=None):
        """A HostBatch whose class tables were computed elsewhere (the native batch harmoniser,
        vcfnative.RawBatch.harmonize): no per-locus Python work.  ``lists``: a callable returning
        (allele_lens, allele_strs) for the few users that want the per-locus Python lists (the oracle-backed
        compute stand-in of the tests)."""
        hb = cls.__new__(cls)
        if not isinstance(gt, np.ndarray) and hasattr(gt, 'ptr'):
            hb.gt = gt                        # an engine.DeviceArray: the tensor was parsed on the device (trk_parse_samples)
        else:
            hb.gt = gt if (gt.dtype == np.int16 and gt.flags['C_CONTIGUOUS']) else np.ascontiguousarray(gt, dtype=np.int16)
        hb.n_loci, hb.n_samples, hb.ploidy = hb.gt.shape
        hb.locus_ploidy = np.ascontiguousarray(locus_ploidy, dtype=np.uint8)
        hb.allele_off, hb.len_class, hb.str_class, hb.len_class_value = allele_off, len_class, str_class, len_class_value
        hb.max_alleles = int(np.max(np.diff(allele_off))) if hb.n_loci else 0
        hb.group_bits = None if group_bits is None else np.ascontiguousarray(group_bits, dtype=np.uint8)
        hb.n_groups = n_groups if group_bits is not None else 1
        hb._lists = lists
        hb._lens = hb._strs = None
        return hb

    def _materialise(self):
        if self._lens is None:
            self._lens, self._strs = self._lists()

    def class_keys(self, l, use_length):
        """Sorted distinct allele representations of locus ``l`` and the class of every index."""
        o, e = int(self.allele_off[l]), int(self.allele_off[l + 1])
        if use_length:
            ranks = self.len_class[o:e]
            n = int(ranks.max()) + 1 if e > o else 0
            return [np.float64(v) for v in self.len_class_value[o:o + n]], ranks
        ranks = self.str_class[o:e]
        keys = sorted(set(self.allele_strs[l]))
        return [np.str_(k) for k in keys], ranks


def _lists_property(name):
    def get(self):
        self._materialise() if getattr(self, '_lists', None) is not None else None
        return getattr(self, name)

    def set_(self, value):
        setattr(self, name, value)
    return property(get, set_)


HostBatch.allele_lens = _lists_property('_lens')
HostBatch.allele_strs = _lists_property('_strs')


def genotype_matrix(vcfrecord):
    """cyvcf2-style ``genotype.array()`` -> (int16 [S, P] without the phase column, P)."""
    g = vcfrecord.genotype
    if g is None:
        return None, 0
    arr = np.asarray(g.array())
    return arr[:, :-1].astype(np.int16), arr.shape[1] - 1


def pack_records(records, group_masks=None):
    """List of TRRecord facades (same sample set) -> HostBatch.

    Loci of lower ploidy are padded with -2 columns and flagged in
    ``locus_ploidy`` so that the kernels ignore the padding (the reference sees
    each record's own ``[S, ploidy]`` matrix)."""
    mats, pls = [], []
    for r in records:
        m, p = genotype_matrix(r.vcfrecord)
        if m is None:
            raise ValueError("record without samples cannot be packed")
        mats.append(m)
        pls.append(p)
    S = mats[0].shape[0] if mats else 0
    P = max(pls) if pls else 1
    gt = np.full((len(mats), S, P), -2, dtype=np.int16)
    for i, m in enumerate(mats):
        gt[i, :, :m.shape[1]] = m
    lens = [[r.ref_allele_length] + list(r.alt_allele_lengths) for r in records]
    strs = [[r.ref_allele] + list(r.alt_alleles) for r in records]
    gb, ng = None, 1
    if group_masks is not None:
        ng = len(group_masks)
        if ng > 8:
            raise ValueError("at most 8 sample groups per pass")
        gb = np.zeros(S, dtype=np.uint8)
        for g, m in enumerate(group_masks):
            gb |= (np.asarray(m, dtype=bool).astype(np.uint8) << g)
    return HostBatch(gt, pls, lens, strs, gb, ng)


def stack_plane(arrays, dtype):
    """Per-record FORMAT arrays ``[S, k_l]`` -> one ``[L, S, k]`` plane (k = max k_l,
    short rows padded with the dtype's missing value)."""
    k = max(a.shape[1] for a in arrays)
    fill = INT_MISSING if np.dtype(dtype) == np.int32 else np.nan
    out = np.full((len(arrays), arrays[0].shape[0], k), fill, dtype=dtype)
    for i, a in enumerate(arrays):
        out[i, :, :a.shape[1]] = a
    return out
