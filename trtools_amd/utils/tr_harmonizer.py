"""Caller-agnostic TR record model with the reference's public surface
(trtools/utils/tr_harmonizer.py): ``VcfTypes``, ``InferVCFType``,
``HarmonizeRecord``, ``TRRecord``, ``TRRecordHarmonizer``.

Per-record *string* work (flank trimming, motif inference, fabricated alleles)
is host Python, as in the reference.  Per-record *reductions* over the samples
(`GetAlleleCounts`, `GetAlleleFreqs`, `GetMaxAllele`, `GetCallRate`,
`GetCalledSamples` counts, the genotype summary used by HWE) are computed by
the HIP kernels of libtrk through ``trtools_amd.runtime.get_compute()``; the
batched tools (statSTR / dumpSTR mirrors) bypass the per-record methods and
send whole batches of loci to the device.
"""
import enum
import re
import warnings

import numpy as np

from . import common
from . import utils
from .. import _lib as L

_beagle_error = "If this file was imputed by Beagle, did you remember to copy the info fields over?"


class VcfTypes(enum.Enum):
    """Supported TR genotypers (tr_harmonizer.py:23-38)."""
    gangstr = "gangstr"
    advntr = "advntr"
    hipstr = "hipstr"
    eh = "eh"
    popstr = "popstr"
    longtr = "longtr"

    def __repr__(self):
        return '<{}.{}>'.format(self.__class__.__name__, self.name)


class TRDosageTypes(enum.Enum):
    """tr_harmonizer.py:40-47."""
    bestguess = "bestguess"
    beagleap = "beagleap"
    bestguess_norm = "bestguess_norm"
    beagleap_norm = "beagleap_norm"

    def __repr__(self):
        return '<{}.{}>'.format(self.__class__.__name__, self.name)


def _ToVCFType(vcftype):
    if isinstance(vcftype, str):
        if vcftype not in VcfTypes.__members__:
            raise ValueError("{} is not an excepted TR vcf type. Expected one of {}".format(
                vcftype, list(VcfTypes.__members__)))
        return VcfTypes[vcftype]
    if isinstance(vcftype, VcfTypes):
        return vcftype
    raise TypeError("{} (of type {}) is not a vcftype".format(vcftype, type(vcftype)))


_IMPURE = {VcfTypes.gangstr: False, VcfTypes.hipstr: True, VcfTypes.longtr: True,
           VcfTypes.advntr: True, VcfTypes.popstr: True, VcfTypes.eh: False}
_LEN_REF = {VcfTypes.gangstr: False, VcfTypes.hipstr: False, VcfTypes.longtr: False,
            VcfTypes.advntr: False, VcfTypes.popstr: False, VcfTypes.eh: True}
_LEN_ALT = {VcfTypes.gangstr: False, VcfTypes.hipstr: False, VcfTypes.longtr: False,
            VcfTypes.advntr: False, VcfTypes.popstr: True, VcfTypes.eh: True}


def MayHaveImpureRepeats(vcftype):
    """tr_harmonizer.py:69-104."""
    return _IMPURE[_ToVCFType(vcftype)]


def HasLengthRefGenotype(vcftype):
    """tr_harmonizer.py:107-139."""
    return _LEN_REF[_ToVCFType(vcftype)]


def HasLengthAltGenotypes(vcftype):
    """tr_harmonizer.py:142-172."""
    return _LEN_ALT[_ToVCFType(vcftype)]


def InferVCFType(vcffile, vcftype="auto"):
    """Guess the genotyper from header text (tr_harmonizer.py:180-244)."""
    header = vcffile.raw_header.lower()
    possible = set()
    has_cmd = 'command=' in header
    if has_cmd and 'gangstr' in header:
        possible.add(VcfTypes.gangstr)
    if has_cmd and 'hipstr' in header:
        possible.add(VcfTypes.hipstr)
    if has_cmd and 'longtr' in header:
        possible.add(VcfTypes.longtr)
    if 'source=advntr' in header:
        possible.add(VcfTypes.advntr)
    if 'source=popstr' in header:
        possible.add(VcfTypes.popstr)
    if re.search(r'alt=<id=str\d+', header):
        possible.add(VcfTypes.eh)
    if len(possible) == 0:
        raise TypeError('Could not identify the type of this vcf')
    if vcftype == 'auto':
        if len(possible) == 1:
            return next(iter(possible))
        raise TypeError(('Confused - this vcf looks like it could have been any of the types: {}. '
                         'Please specify --vcftype to choose one of them').format(possible))
    user = _ToVCFType(vcftype)
    if user in possible:
        return user
    raise TypeError(('Confused - this vcf looks like it could have been any of the types: {}. '
                     'But you specified: --vcftype {} which is not one of those types.'
                     .format(possible, vcftype)))


def IsBeagleVCF(vcffile):
    """tr_harmonizer.py:246-262."""
    return bool(re.search('##source=(\'|")beagle', vcffile.raw_header.lower()))


# ---------------------------------------------------------------------------
# per-caller harmonisation (tr_harmonizer.py:264-550)
# ---------------------------------------------------------------------------

def _upper_all(alleles):
    return [a.upper() for a in alleles]


def _HarmonizeGangSTRRecord(rec):
    where = "{}:{}".format(rec.CHROM, rec.POS)
    if rec.INFO.get('RU') is None:
        raise TypeError("Record at {} is missing mandatory GangSTR info field RU. ".format(where) + _beagle_error)
    if rec.INFO.get('VID') is not None:
        raise TypeError("Trying to read an AdVNTR record as a GangSTR record {}".format(where))
    if rec.INFO.get('VARID') is not None:
        raise TypeError("Trying to read an EH record as a GangSTR record {}".format(where))
    alts = _upper_all(rec.ALT) if rec.ALT else []
    qual = 'Q' if rec.INFO.get('IMP') is None else None
    return TRRecord(rec, rec.REF.upper(), alts, rec.INFO["RU"].upper(), None, qual)


def _HarmonizeHipSTRRecord(rec):
    info = rec.INFO
    if info.get('START') is None or info.get('END') is None or info.get('PERIOD') is None:
        raise TypeError("Record at {}:{} is missing one of the mandatory HipSTR/LongTR info fields "
                        "START, END, PERIOD. ".format(rec.CHROM, rec.POS) + _beagle_error)
    pos = int(rec.POS)
    lead = int(info['START']) - pos                 # flanking bp before the repeat
    tail = int(info['END']) - pos + 1 - len(rec.REF)  # <= 0: flanking bp after the repeat
    alts_raw = [str(a) for a in rec.ALT] if rec.ALT else []
    full_alleles = None
    if lead != 0 or tail != 0:
        full_alleles = (rec.REF.upper(), _upper_all(alts_raw))
    stop = None if tail == 0 else tail
    ref_allele = rec.REF[lead:stop].upper()
    alt_alleles = [a[lead:stop].upper() for a in alts_raw]
    # the reference slices the already trimmed allele by the leading offset again
    # before inferring the motif (tr_harmonizer.py:397); kept for parity
    motif = utils.InferRepeatSequence(ref_allele[lead:], info["PERIOD"])
    qual = 'Q' if info.get('IMP') is None else None
    return TRRecord(rec, ref_allele, alt_alleles, motif, rec.ID, qual,
                    harmonized_pos=int(info['START']), full_alleles=full_alleles)


def _HarmonizeAdVNTRRecord(rec):
    if rec.INFO.get('RU') is None or rec.INFO.get('VID') is None:
        raise TypeError("Record at {}:{} is missing one of the mandatory ADVNTR info fields RU, VID. "
                        .format(rec.CHROM, rec.POS) + _beagle_error)
    alts = _upper_all(rec.ALT) if rec.ALT else []
    qual = 'ML' if rec.INFO.get('IMP') is None else None
    return TRRecord(rec, rec.REF.upper(), alts, rec.INFO["RU"].upper(), rec.INFO["VID"], qual)


def _bracket_lengths(alts, prefix, what):
    out = []
    for alt in alts:
        alt = str(alt)
        if not alt.startswith(prefix) or not alt.endswith(">"):
            raise TypeError("This record does not look like {} record. Alt alleles were not formatted"
                            " as expected".format(what))
        out.append(float(alt[len(prefix):-1]))
    return out


def _HarmonizePopSTRRecord(rec):
    if rec.INFO.get('Motif') is None:
        raise TypeError("Record at {}:{} is missing mandatory PopSTR info field MOTIF".format(rec.CHROM, rec.POS))
    lengths = _bracket_lengths(rec.ALT, "<", "a PopSTR") if rec.ALT else []
    return TRRecord(rec, rec.REF.upper(), None, rec.INFO["Motif"].upper(), rec.ID, None,
                    alt_allele_lengths=lengths)


def _HarmonizeEHRecord(rec):
    if rec.INFO.get('VARID') is None or rec.INFO.get('RU') is None:
        raise TypeError("Record at {}:{} is missing one of the mandatory ExpansionHunter info fields VARID, RU. "
                        .format(rec.CHROM, rec.POS) + _beagle_error)
    motif = rec.INFO["RU"].upper()
    ref_len = int(rec.INFO["RL"]) / len(motif)
    lengths = _bracket_lengths(rec.ALT, "<STR", "an EH ") if rec.ALT else []
    return TRRecord(rec, None, None, motif, rec.INFO["VARID"], None,
                    ref_allele_length=ref_len, alt_allele_lengths=lengths)


_HARMONIZERS = {
    VcfTypes.gangstr: _HarmonizeGangSTRRecord, VcfTypes.hipstr: _HarmonizeHipSTRRecord,
    VcfTypes.longtr: _HarmonizeHipSTRRecord, VcfTypes.advntr: _HarmonizeAdVNTRRecord,
    VcfTypes.eh: _HarmonizeEHRecord, VcfTypes.popstr: _HarmonizePopSTRRecord,
}


def HarmonizeRecord(vcftype, vcfrecord):
    """Variant of a known caller -> TRRecord (tr_harmonizer.py:264-300)."""
    return _HARMONIZERS[_ToVCFType(vcftype)](vcfrecord)


class _FormatDict:
    """dict-like view of the record's FORMAT fields (tr_harmonizer.py:561-588)."""

    def __init__(self, record):
        self.record = record

    def __getitem__(self, key):
        return self.record.format(key)

    def __len__(self):
        return len(self.record.FORMAT)

    def __iter__(self):
        return iter(self.record.FORMAT)

    def __contains__(self, key):
        return key in self.record.FORMAT

    def keys(self):
        return self.record.FORMAT

    def get(self, key):
        return self.record.format(key)


class TRRecord:
    """Harmonised TR record (tr_harmonizer.py:591-1647): same constructor,
    attributes and methods; sample reductions run on the GPU."""

    def __init__(self, vcfrecord, ref_allele, alt_alleles, motif, record_id, quality_field, *,
                 harmonized_pos=None, full_alleles=None, ref_allele_length=None,
                 alt_allele_lengths=None, quality_score_transform=None):
        self.vcfrecord = vcfrecord
        self.ref_allele = ref_allele
        self.alt_alleles = alt_alleles
        self.motif = motif
        self.record_id = record_id
        self.chrom = vcfrecord.CHROM
        self.pos = harmonized_pos if harmonized_pos is not None else vcfrecord.POS
        self.info = dict(vcfrecord.INFO)
        self.format = _FormatDict(vcfrecord)
        self.full_alleles = full_alleles
        self.full_alleles_pos = vcfrecord.POS
        self.ref_allele_length = ref_allele_length
        self.alt_allele_lengths = alt_allele_lengths
        self.quality_field = quality_field
        self.quality_score_transform = quality_score_transform

        if full_alleles is not None and (alt_alleles is None or ref_allele is None):
            raise ValueError("Cannot set full alleles without setting regular alleles")
        if alt_allele_lengths is not None and alt_alleles is not None:
            raise ValueError("Must specify only the sequences or the lengths of the alt alleles, not both.")
        if ref_allele_length is not None and alt_allele_lengths is None:
            raise ValueError("If the ref allele is specified by length, the alt alleles must be too.")

        self.has_fabricated_ref_allele = ref_allele_length is not None
        if self.has_fabricated_ref_allele:
            self.ref_allele = utils.FabricateAllele(motif, ref_allele_length)
        else:
            self.ref_allele_length = len(ref_allele) / len(motif)
        # lengths may be fractional (partial repeats): round, do not truncate
        self.end_pos = round(self.pos + self.ref_allele_length * len(motif) - 1)
        self.full_alleles_end_pos = self.end_pos if full_alleles is None else \
            round(self.full_alleles_pos + len(self.full_alleles[0]) - 1)

        self.has_fabricated_alt_alleles = alt_allele_lengths is not None
        if self.has_fabricated_alt_alleles:
            self.alt_alleles = [utils.FabricateAllele(motif, n) for n in alt_allele_lengths]
        else:
            self.alt_allele_lengths = [len(a) / len(motif) for a in self.alt_alleles]

        all_lengths = [self.ref_allele_length] + list(self.alt_allele_lengths)
        self.min_allele_length = min(all_lengths)
        self.max_allele_length = max(all_lengths)
        try:
            self._CheckRecord()
        except ValueError as e:
            raise ValueError("Invalid TRRecord. TRRecord: {} Original record: {}".format(
                str(self), str(self.vcfrecord)), e)

    def _CheckRecord(self):
        """tr_harmonizer.py:775-808."""
        if len(self.alt_alleles) != len(self.vcfrecord.ALT):
            raise ValueError("Underlying record does not have the same number of alt alleles as given to the "
                             "TRRecord constructor. Underlying alt alleles: {},  constructor alt alleles: {}"
                             .format(self.vcfrecord.ALT, self.alt_alleles))
        if self.full_alleles:
            if len(self.full_alleles) != 2:
                raise ValueError("full_alleles doesn't have both a ref allele and alt alleles")
            full_ref, full_alts = self.full_alleles
            if len(full_alts) != len(self.alt_alleles):
                raise ValueError("Different number of full alternate alleles than normal alt alleles")
            if self.ref_allele not in full_ref:
                raise ValueError("could not find ref allele inside full ref allele")
            for i, (full_alt, alt) in enumerate(zip(full_alts, self.alt_alleles)):
                if alt not in full_alt:
                    raise ValueError("Could not find alt allele {} inside its full alt allele".format(i))

    # ---- shapes -----------------------------------------------------------------
    def GetMaxPloidy(self):
        return self.vcfrecord.ploidy

    def GetNumSamples(self):
        return self.vcfrecord.genotype.n_samples

    def GetGenotypeIndicies(self):
        """int array [S, ploidy+1], last column = phased (tr_harmonizer.py:829-862)."""
        if self.vcfrecord.genotype is None:
            return None
        return self.vcfrecord.genotype.array().astype(int)

    def GetCalledSamples(self, strict=True):
        """bool [S] (tr_harmonizer.py:864-897); a per-sample mask, computed where the
        genotype matrix lives (host) -- the locus-level count comes from the device."""
        g = self.GetGenotypeIndicies()
        if g is None:
            return None
        g = g[:, :-1]
        if strict:
            return ~np.any(g == -1, axis=1)
        return ~np.all((g == -1) | (g == -2), axis=1)

    def GetSamplePloidies(self):
        g = self.GetGenotypeIndicies()
        if g is None:
            return None
        return g.shape[1] - 1 - np.sum(g[:, :-1] == -2, axis=1)

    def GetCallRate(self, strict=True):
        """tr_harmonizer.py:921-946."""
        if self.vcfrecord.genotype is None:
            return None
        if not strict:
            called = self.GetCalledSamples(strict=False)
            return np.sum(called) / called.shape[0]
        st, _ = self._device_stats()
        n = st.locus_int[0, 0, L.LI_N_SAMPLES]
        return np.int64(st.locus_int[0, 0, L.LI_N_CALLED]) / n if n else np.float64('nan')

    # ---- genotype representations (host array transforms) ----------------------------
    def _string_array(self, idx_gts, seq_alleles):
        width = max(len(a) for a in seq_alleles)
        out = np.empty(idx_gts.shape, dtype="<U{}".format(width))
        phase = idx_gts[:, -1]
        out[:, -1][phase == 0] = '0'
        out[:, -1][phase == 1] = '1'
        body, gbody = out[:, :-1], idx_gts[:, :-1]
        for i, s in enumerate(seq_alleles):
            body[gbody == i] = s
        body[gbody == -1] = '.'
        body[gbody == -2] = ','
        return out

    def GetStringGenotypes(self):
        """tr_harmonizer.py:963-1017."""
        g = self.GetGenotypeIndicies()
        if g is None:
            return None
        if self.HasFabricatedAltAlleles():
            warnings.warn("String genotypes have been requested for a TRRecord generated by a caller which "
                          "only generates length genotypes, not string genotypes. Returning a fabricated "
                          "string genotype. Consider requesting length based genotypes instead.")
        return self._string_array(g, [self.ref_allele] + list(self.alt_alleles))

    def GetFullStringGenotypes(self):
        """tr_harmonizer.py:1019-1047."""
        if not self.HasFullStringGenotypes():
            return self.GetStringGenotypes()
        g = self.GetGenotypeIndicies()
        if g is None:
            return None
        return self._string_array(g, [self.full_alleles[0]] + list(self.full_alleles[1]))

    def GetLengthGenotypes(self):
        """tr_harmonizer.py:1210-1245."""
        g = self.GetGenotypeIndicies()
        if g is None:
            return None
        lut = np.array([self.ref_allele_length, *self.alt_allele_lengths, -2, -1])
        out = lut[g]
        out[:, -1] = g[:, -1]
        return out

    def _unique_mapping(self, key):
        first, mapping = {}, {}
        for i, a in enumerate([self.ref_allele] + list(self.alt_alleles)):
            k = key(a)
            mapping[i] = first.setdefault(k, i)
        return mapping

    def UniqueStringGenotypeMapping(self):
        """tr_harmonizer.py:1049-1082."""
        if not self.HasFullStringGenotypes():
            return {i: i for i in range(len(self.alt_alleles) + 1)}
        return self._unique_mapping(lambda a: a)

    def UniqueStringGenotypes(self):
        return set(self.UniqueStringGenotypeMapping().values())

    def UniqueLengthGenotypeMapping(self):
        """tr_harmonizer.py:1247-1273."""
        return self._unique_mapping(len)

    def UniqueLengthGenotypes(self):
        return set(self.UniqueLengthGenotypeMapping().values())

    def HasFullStringGenotypes(self):
        return self.full_alleles is not None

    def HasFabricatedRefAllele(self):
        return self.has_fabricated_ref_allele

    def HasFabricatedAltAlleles(self):
        return self.has_fabricated_alt_alleles

    # ---- reductions over samples: GPU ----------------------------------------------------
    def _device_stats(self, sample_index=None):
        """One-locus batch through trk_locus_stats; ``sample_index`` as one sample group.

        The result is kept per (genotype array, sample_index): GetAlleleCounts, GetAlleleFreqs and GetMaxAllele of one
        record cost one upload and one launch together, not one each (the reference recomputes per call,
        tr_harmonizer.py:1420-1575).  ``sample_index`` may be anything numpy takes as a row index (:1400-1401,
        :1488-1489): a boolean mask or unique indices become a sample group on the device; an index that repeats
        samples is expanded on the host first (the repeated rows are uploaded as samples of their own), so that a
        sample counts as often as it is named."""
        from .. import runtime
        from ..batch import pack_records
        gt_obj = self.vcfrecord._gt if hasattr(self.vcfrecord, '_gt') else None
        key = None
        if sample_index is not None:
            si = np.asarray(sample_index)
            key = (si.dtype.str, si.shape, si.tobytes())
        cache = getattr(self, '_stats_cache', None)
        if cache is not None and cache[0] is gt_obj and gt_obj is not None and key in cache[1]:
            return cache[1][key]
        masks, expand = None, None
        if sample_index is not None:
            if si.dtype == bool:
                masks = [si]
            else:
                idx = np.arange(self.GetNumSamples())[si]          # numpy's own index semantics (negatives, slices)
                if len(np.unique(idx)) == len(idx):
                    m = np.zeros(self.GetNumSamples(), dtype=bool)
                    m[idx] = True
                    masks = [m]
                else:
                    expand = idx
        hb = pack_records([self], masks)
        if expand is not None:
            hb.gt = np.ascontiguousarray(hb.gt[:, expand, :])
            hb.n_samples = hb.gt.shape[1]
        st = runtime.get_compute().locus_stats(hb)
        if st.locus_int[0, 0, L.LI_N_BAD]:
            raise IndexError("genotype index out of range for the alleles of record {}".format(str(self)))
        if gt_obj is not None:
            if cache is None or cache[0] is not gt_obj:
                self._stats_cache = cache = (gt_obj, {})
            if len(cache[1]) < 16:
                cache[1][key] = (st, hb)
        return st, hb

    def GetAlleleCounts(self, sample_index=None, *, uselength=True, index=False, fullgenotypes=False):
        """{allele: count} of called haplotypes (tr_harmonizer.py:1420-1499)."""
        if uselength and fullgenotypes:
            raise ValueError("Can't specify both uselength and fullgenotypes")
        if index and not uselength:
            raise ValueError("Specified uselength=False and index at the same time, these are mutually "
                             "exclusive options")
        if self.vcfrecord.genotype is None or self.GetNumSamples() == 0:
            return {}
        if not uselength and not fullgenotypes and self.HasFabricatedAltAlleles():
            warnings.warn("String genotypes have been requested for a TRRecord generated by a caller which "
                          "only generates length genotypes, not string genotypes.")
        st, hb = self._device_stats(sample_index)
        cnt = st.allele_count[0]
        if index:
            return {np.int64(i): np.int64(c) for i, c in enumerate(cnt) if c}
        if fullgenotypes and self.HasFullStringGenotypes():
            full = [self.full_alleles[0]] + list(self.full_alleles[1])
            acc = {}
            for i, c in enumerate(cnt):
                if c:
                    acc[full[i]] = acc.get(full[i], 0) + int(c)
            return {np.str_(k): np.int64(acc[k]) for k in sorted(acc)}
        keys, ranks = hb.class_keys(0, uselength)
        cc = np.zeros(len(keys), dtype=np.int64)
        np.add.at(cc, ranks, cnt)
        return {keys[c]: np.int64(cc[c]) for c in range(len(keys)) if cc[c]}

    def GetAlleleFreqs(self, sample_index=None, *, uselength=True, index=False, fullgenotypes=False):
        """tr_harmonizer.py:1501-1540."""
        counts = self.GetAlleleCounts(uselength=uselength, index=index, fullgenotypes=fullgenotypes,
                                      sample_index=sample_index)
        total = float(sum(counts.values()))
        return {k: v / total for k, v in counts.items()}

    def GetMaxAllele(self, sample_index=None):
        """tr_harmonizer.py:1542-1575."""
        if self.vcfrecord.genotype is None or self.GetNumSamples() == 0:
            return np.nan
        st, _ = self._device_stats(sample_index)
        return np.float64(st.locus_f64[0, 0, L.LF_THRESH])

    def GetGenotypeCounts(self, sample_index=None, uselength=True, index=False, fullgenotypes=False,
                          include_nocalls=False):
        """{sorted genotype tuple: count} (tr_harmonizer.py:1326-1418).

        API helper outside the statSTR/dumpSTR kernels: those consume only the
        two numbers the reference derives from this table (its total and the
        homozygote count, utils.py:326-333), which the device emits directly
        (TRK_LI_N_CALLED / TRK_LI_N_HOM_*).  The full table is tabulated here
        from the host genotype array."""
        if uselength and fullgenotypes:
            raise ValueError("Can't specify both uselength and fullgenotypes")
        if index and not uselength:
            raise ValueError("Specified uselength=False and index at the same time, these are mutually "
                             "exclusive options")
        if index:
            gts, nocall = self.GetGenotypeIndicies(), -1
        elif uselength:
            gts, nocall = self.GetLengthGenotypes(), -1
        elif fullgenotypes:
            gts, nocall = self.GetFullStringGenotypes(), '.'
        else:
            gts, nocall = self.GetStringGenotypes(), '.'
        if gts is None:
            return {}
        gts = np.sort(gts[:, :-1], axis=1)
        if sample_index is not None:
            gts = gts[sample_index, :]
        if gts.shape[0] == 0:
            return {}
        rows, counts = np.unique(gts, axis=0, return_counts=True)
        out = dict(zip(map(tuple, rows), counts))
        if not include_nocalls:
            out = {g: c for g, c in out.items() if nocall not in g}
        return out

    def GetDosages(self, dosagetype=TRDosageTypes.bestguess, strict=True):
        """Per-sample dosages (tr_harmonizer.py:1098-1208); a per-sample transform
        (annotaTR path), host numpy."""
        n = self.GetNumSamples()
        if n == 0:
            return None
        beagle = dosagetype in (TRDosageTypes.beagleap, TRDosageTypes.beagleap_norm)
        norm = dosagetype in (TRDosageTypes.bestguess_norm, TRDosageTypes.beagleap_norm)

        def problem(msg, raise_msg=None):
            if strict:
                raise ValueError(raise_msg or msg)
            common.WARNING(msg)
            return np.array([np.nan] * n)

        if beagle:
            fmt = self.vcfrecord.FORMAT
            if "AP1" not in fmt or "AP2" not in fmt or self.vcfrecord.format("AP1") is None \
                    or self.vcfrecord.format("AP2") is None:
                return problem("Requested Beagle dosages for record at {}:{} but AP1/AP2 fields not found."
                               .format(self.chrom, self.pos))
            ap1, ap2 = self.vcfrecord.format("AP1"), self.vcfrecord.format("AP2")
            if np.any(np.sum(ap1, axis=1) > 1.1) or np.any(np.sum(ap2, axis=1) > 1.1):
                return problem("{}:{} AP1 or AP2 field summing to more than 1 detected".format(self.chrom, self.pos))
            if np.any(ap1 < 0) or np.any(ap2 < 0):
                return problem("{}:{} Negative AP1 or AP2 fields detected".format(self.chrom, self.pos),
                               "Negative AP1 or AP2 fields detected")
            ref1 = np.clip(1 - np.sum(ap1, axis=1), 0, 1)
            ref2 = np.clip(1 - np.sum(ap2, axis=1), 0, 1)
            if len(self.alt_allele_lengths) > 0:
                cap = max(self.alt_allele_lengths)
                h1 = np.clip(np.dot(ap1, self.alt_allele_lengths), 0, cap)
                h2 = np.clip(np.dot(ap2, self.alt_allele_lengths), 0, cap)
            else:
                h1 = h2 = 0
            unnorm = (h1 + h2 + ref1 * self.ref_allele_length + ref2 * self.ref_allele_length).astype(np.float32)
        elif dosagetype in (TRDosageTypes.bestguess, TRDosageTypes.bestguess_norm):
            lengts = self.GetLengthGenotypes()
            fill = np.nan if norm else 0
            lengts[lengts == -1] = fill
            lengts[lengts == -2] = fill
            unnorm = lengts[:, :-1].sum(axis=1).astype(np.float32)
        else:
            raise ValueError("Unsupported dosagetype")
        if not norm:
            return unnorm
        if self.min_allele_length == self.max_allele_length:
            return np.zeros(n, dtype=np.float32)
        dos = (unnorm - 2 * self.min_allele_length) / (self.max_allele_length - self.min_allele_length)
        if np.any(dos >= 2.1) or np.any(dos <= -0.1):
            return problem("{}:{} Error normalizing dosages: value >=2.1 or <=-0.1 detected"
                           .format(self.chrom, self.pos))
        return np.clip(dos, 0, 2)

    # ---- quality scores ---------------------------------------------------------------
    def HasQualityScores(self):
        return self.quality_field is not None and self.quality_field in self.format

    def GetQualityScores(self):
        """tr_harmonizer.py:1592-1615."""
        if not self.HasQualityScores():
            raise TypeError("This TRRecord does not have a corresponding quality score field")
        val = self.format[self.quality_field]
        if self.quality_score_transform is None:
            return val
        return np.apply_along_axis(self.quality_score_transform, 0, val)

    def __str__(self):
        rid = self.record_id
        if rid is None:
            rid = "{}:{}".format(self.vcfrecord.CHROM, self.vcfrecord.POS)
        if self.HasFullStringGenotypes():
            return "{} {} {} ".format(rid, self.motif, self.full_alleles[0]) + ",".join(self.full_alleles[1])
        if self.HasFabricatedRefAllele():
            s = "{} {} n_reps:{} ".format(rid, self.motif, self.ref_allele_length)
        else:
            s = "{} {} {} ".format(rid, self.motif, self.ref_allele)
        if len(self.alt_alleles) == 0:
            return s + '.'
        if self.HasFabricatedAltAlleles():
            return s + ",".join("n_reps:" + str(n) for n in self.alt_allele_lengths)
        return s + ','.join(self.alt_alleles)


class TRRecordHarmonizer:
    """Iterator of TRRecords over a VCF reader (tr_harmonizer.py:1650-1779)."""

    def __init__(self, vcffile, vcftype="auto"):
        self.vcffile = vcffile
        self.vcftype = InferVCFType(vcffile, vcftype)
        self._record_idx = None

    def MayHaveImpureRepeats(self):
        return MayHaveImpureRepeats(self.vcftype)

    def HasLengthRefGenotype(self):
        return HasLengthRefGenotype(self.vcftype)

    def HasLengthAltGenotypes(self):
        return HasLengthAltGenotypes(self.vcftype)

    def HasQualityScore(self):
        """tr_harmonizer.py:1721-1749."""
        if self.vcftype == VcfTypes.gangstr:
            return 'FORMAT=<ID=Q,' in self.vcffile.raw_header
        if self.vcftype in (VcfTypes.hipstr, VcfTypes.longtr, VcfTypes.advntr):
            return not self.IsBeagleVCF()
        return False

    def IsBeagleVCF(self):
        return IsBeagleVCF(self.vcffile)

    def __iter__(self):
        return self

    def __next__(self):
        self._record_idx = 2 if self._record_idx is None else self._record_idx + 1
        try:
            record = next(self.vcffile)
        except StopIteration:
            raise
        except Exception:
            raise ValueError("Unable to parse the " + str(self._record_idx) + "th tandem repeat in the "
                             "provided VCF. Check that it is properly formatted.")
        return HarmonizeRecord(self.vcftype, record)


def GetDosagesBatch(records, dosagetype=TRDosageTypes.bestguess, strict=True):
    """``TRRecord.GetDosages`` (reference tr_harmonizer.py:1098-1208) for a list of records of one
    sample set in one device call (``trk_dosages``).  Returns a float32
    array ``[len(records), n_samples]``; errors follow the reference record by record: ValueError
    with its message when ``strict``, otherwise a warning and a row of nan."""
    from .. import runtime
    from ..batch import pack_records, stack_plane
    if not records:
        return np.zeros((0, 0), dtype=np.float32)
    beagle = dosagetype in (TRDosageTypes.beagleap, TRDosageTypes.beagleap_norm)
    norm = dosagetype in (TRDosageTypes.bestguess_norm, TRDosageTypes.beagleap_norm)
    n = records[0].GetNumSamples()
    ok = list(range(len(records)))
    out = np.full((len(records), n), np.nan, dtype=np.float32)

    def problem(i, msg, raise_msg=None):
        if strict:
            raise ValueError(raise_msg or msg)
        common.WARNING(msg)
        ok.remove(i)

    ap1 = ap2 = None
    if beagle:
        for i, r in enumerate(records):
            fmt = r.vcfrecord.FORMAT
            if "AP1" not in fmt or "AP2" not in fmt or r.vcfrecord.format("AP1") is None \
                    or r.vcfrecord.format("AP2") is None:
                problem(i, "Requested Beagle dosages for record at {}:{} but AP1/AP2 fields not found."
                        .format(r.chrom, r.pos))
        if not ok:
            return out
    recs = [records[i] for i in ok]
    hb = pack_records(recs)
    if beagle:
        ap1 = stack_plane([np.asarray(r.vcfrecord.format("AP1"), dtype=np.float32) for r in recs], np.float32)
        ap2 = stack_plane([np.asarray(r.vcfrecord.format("AP2"), dtype=np.float32) for r in recs], np.float32)
    dos, err = runtime.get_compute().dosages_batch(hb, dosagetype.name, ap1, ap2)
    for k, i in enumerate(list(ok)):
        r, e = records[i], int(err[k])
        if e & 1:
            problem(i, "{}:{} AP1 or AP2 field summing to more than 1 detected".format(r.chrom, r.pos))
        elif e & 2:
            problem(i, "{}:{} Negative AP1 or AP2 fields detected".format(r.chrom, r.pos),
                    "Negative AP1 or AP2 fields detected")
        elif norm and (e & 4):
            problem(i, "{}:{} Error normalizing dosages: value >=2.1 or <=-0.1 detected".format(r.chrom, r.pos))
        else:
            out[i] = dos[k]
    return out
