"""The compute seam between the Python host layer and libtrk.

``DeviceCompute`` uploads a HostBatch, runs the HIP kernels through the C ABI
and brings the (small) per-locus / per-sample results back as numpy arrays.
The host layer (tr_harmonizer / statSTR / dumpSTR mirrors) only talks to this
interface; the tests substitute an oracle-backed object with the same methods
to exercise the host logic on machines without a GPU -- the product itself has
no CPU implementation."""
from . import _knobs
import os

import numpy as np


class StatsHost:
    """Host copies of trk_stats_out."""

    def __init__(self, allele_count, locus_int, locus_f64):
        self.allele_count = allele_count   # [G, sumA] int32
        self.locus_int = locus_int         # [G, L, TRK_LI_COLS] int32
        self.locus_f64 = locus_f64         # [G, L, TRK_LF_COLS] float64


class CallHost:
    def __init__(self, gt_out, mask, sample_counters, totaldp, dp_missing, error):
        self.gt_out = gt_out
        self.mask = mask
        self.sample_counters = sample_counters
        self.totaldp = totaldp
        self.dp_missing = dp_missing
        self.error = error
        self.dev = None          # dumpstr_batch(keep_device=True): dict(mask8=, planes=[...], stride=) of DeviceArrays

    def release_device(self):
        d, self.dev = self.dev, None
        if d is not None:
            for a in [d.get('mask8')] + list(d.get('planes', [])):
                if a is not None and a.ptr is not None:
                    a.free()


class AssocHost:
    """Host copies of trk_assoc_out."""

    def __init__(self, locus_int, locus_f64, allele_count):
        self.locus_int = locus_int         # [L, AI_COLS] int32
        self.locus_f64 = locus_f64         # [L, AF_COLS] float64
        self.allele_count = allele_count   # [sumA] int32, tested samples only


class DeviceCompute:
    def __init__(self, engine=None, device=0, reserve_pair_gb=None):
        from .engine import Engine
        self.eng = engine if engine is not None else Engine(device, reserve_pair_gb=reserve_pair_gb)

    def host_buffer(self, nbytes):
        """Pinned staging memory for the native reader's batch arrays (Engine.host_buffer)."""
        return self.eng.host_buffer(nbytes)

    def host_release(self, arr):
        """Give a ``host_buffer`` back (the reader does when it is closed): the next reader reuses it."""
        self.eng.sync()                      # uploads from the buffer may still be queued
        self.eng.host_buffer_release(arr)

    @staticmethod
    def _n_pad(hb, rows=False):
        """Padding samples appended to every row (trk_batch.n_pad_samples): to a multiple of 4 so that the diploid
        rows are whole 16-byte chunks; ``rows`` (the statSTR / dumpSTR passes): to a multiple of TRK_ROW_ALIGN
        samples (default 32 = 128 bytes) from 512 samples on, so that every row of the tensor and of the FORMAT
        planes starts on a cache-line boundary (include/trk.h, trk_pad_rows: 3-4 % of the call-filter pass)."""
        S = hb.gt.shape[1]
        if not (hb.gt.shape[2] == 2 and S > 0 and _knobs.lab('TRK_PAD_SAMPLES', '1') != '0'):
            return 0
        align = 4
        if rows and S >= 512:
            align = max(4, int(_knobs.lab('TRK_ROW_ALIGN', '32')) & ~3)
        return (-S) % align

    @staticmethod
    def _pad(arr, n_pad, axis, value):
        if not n_pad:
            return arr
        arr = np.asarray(arr)
        shape = list(arr.shape)
        shape[axis] = n_pad
        return np.concatenate([arr, np.full(shape, value, dtype=arr.dtype)], axis=axis)

    def _upload(self, hb, pad=True, rows=False, allow_class_sort=False):
        # a per-locus ploidy table only when some locus really is of lower ploidy than the tensor
        # (the kernels' streaming paths need every column to be live)
        lp = hb.locus_ploidy
        if lp is not None and (hb.n_loci == 0 or bool(np.all(np.asarray(lp) == hb.ploidy))):
            lp = None
        n_pad = self._n_pad(hb, rows) if pad else 0
        gb = hb.group_bits
        # Sample groups: with four or more groups per pass (TRK_CLASS_SORT=1: always, =0: never) the columns are
        # gathered on the device into class order and counted range by range with the ungrouped streaming kernel
        # (DeviceBatch.sorted_by_class; the per-call group kernel takes 19-23 ms at 100k x 10k where this takes ~2).
        # Up to three groups keep the one-pass grouped kernel (k_locus_count_v2g), which needs no gather.
        lay = getattr(hb, 'class_layout', None)
        if lay is not None and allow_class_sort:
            # the columns arrive in class order already (the reader laid them out while it parsed the records,
            # trk_vcf_set_sample_map): no gather, the ungrouped streaming kernel per class range
            base = self.eng.make_batch(hb.gt, hb.allele_off, hb.len_class, hb.str_class, hb.len_class_value,
                                       max_alleles=hb.max_alleles)
            return base.with_class_layout(self.eng, lay, hb.n_groups)
        mode = _knobs.lab('TRK_CLASS_SORT', '')
        # (only trk_locus_stats reads a class-ordered batch: every other entry point needs the samples in their
        # own order and number, so the gather is an opt-in of locus_stats -- ADVICE r03)
        class_sort = (allow_class_sort and gb is not None and hb.gt.shape[2] == 2 and lp is None and hb.n_loci > 0 and mode != '0' and
                      (hb.n_groups >= 4 or mode == '1'))
        if class_sort:
            base = self.eng.make_batch(hb.gt, hb.allele_off, hb.len_class, hb.str_class, hb.len_class_value,
                                       max_alleles=hb.max_alleles)
            b = base.sorted_by_class(self.eng, gb, hb.n_groups)
            if b.arrays['gt'] is not base.arrays['gt']:
                base.arrays['gt'].free()
            return b
        if n_pad and gb is not None:
            gb = self._pad(gb, n_pad, 0, 0)          # padding samples belong to no group
        # (the padding columns are appended on the device: no host copy of the tensor)
        # (a genotype tensor that is on the device already -- parsed there, trk_parse_samples -- is taken over as it is)
        from .engine import DeviceArray
        gt_d = self.eng.pad_samples(hb.gt if isinstance(hb.gt, DeviceArray) else self.eng.upload(hb.gt, np.int16), n_pad)
        return self.eng.make_batch(gt_d, hb.allele_off, hb.len_class, hb.str_class,
                                   hb.len_class_value, locus_ploidy=lp, group_bits=gb,
                                   n_groups=hb.n_groups, max_alleles=hb.max_alleles, n_pad=n_pad)

    @staticmethod
    def _free(*objs):
        for o in objs:
            if o is None:
                continue
            if hasattr(o, 'arrays'):
                for a in o.arrays.values():
                    a.free()
            elif hasattr(o, 'free'):
                o.free()

    def locus_stats(self, hb, nalleles_thresh=0.01):
        b = self._upload(hb, rows=True, allow_class_sort=True)
        res = self.eng.locus_stats(b, nalleles_thresh=nalleles_thresh)
        out = StatsHost(res.allele_count.get(), res.locus_int.get(), res.locus_f64.get())
        self._free(b, res.allele_count, res.locus_int, res.locus_f64)
        return out

    supports_compact = True
    supports_class_layout = True      # locus_stats takes a HostBatch whose columns are in engine.class_layout order

    def dumpstr_batch(self, hb, planes, filters, dp_plane, locus_spec, nalleles_thresh=0.01, compact=False, keep_device=False):
        """Call filters -> masked genotypes -> locus statistics -> locus filters, all on the device.
        Returns (CallHost, StatsHost, locus_bits uint32[L], loc_counters int64[32]).  ``compact``: the caller
        rebuilds its records from the mask -- only the one-byte mask comes back (trk_call_out.filter_mask8; CallHost.mask
        is uint8 then, gt_out None), not the masked genotype tensor and the 32-bit mask."""
        eng = self.eng
        compact = compact and len(filters) <= 7
        b = self._upload(hb, rows=True)
        S = hb.gt.shape[1]
        n_pad = b.struct.n_samples - S      # the padding samples' FORMAT values are missing like their genotypes
        # (planes that are on the device already -- parsed there, trk_parse_samples -- are taken over, padded on the device)
        from .engine import DeviceArray
        dplanes = [eng.pad_samples(p, n_pad) if isinstance(p, DeviceArray) else eng.upload_plane(p, n_pad=n_pad) for p in planes]
        # counts of the unfiltered genotypes, corrected by the call-filter kernel for every call it
        # masks (dumpSTR.py:721-774 rebuilds the record; here no second pass over the tensor), then
        # the finaliser
        st = eng.locus_stats(b, nalleles_thresh=nalleles_thresh, count_only=True)
        cout = eng.alloc_call_out(b, len(filters), want_gt=not compact, want_mask=not compact, want_mask8=compact)
        call = eng.call_filters(b, dplanes, filters, dp_plane=dp_plane, out=cout, delta_stats=st)
        eng.locus_finalize(b, st, nalleles_thresh=nalleles_thresh)
        ext = None
        spec = dict(locus_spec)
        ext_host = spec.pop('extern_bits', None)
        if ext_host is not None:
            ext = eng.upload(np.ascontiguousarray(ext_host, dtype=np.uint32))
        bits, counters = eng.locus_filters(hb.n_loci, st, extern_bits=ext, **spec)
        totaldp = call.sample_totaldp.get()
        if dp_plane >= 0 and np.dtype(planes[dp_plane].dtype).kind == 'f':
            totaldp = call.sample_totaldp_f64.get()      # Float depth plane (ExpansionHunter's LC)
        ch = CallHost(None if compact else call.gt_out.get()[:, :S],
                      call.filter_mask8.get()[:, :S] if compact else call.filter_mask.get()[:, :S],
                      call.sample_counters.get()[:, :S], totaldp[:S], call.sample_dp_missing.get()[:S], call.error.get())
        sh = StatsHost(st.allele_count.get(), st.locus_int.get(), st.locus_f64.get())
        out = (ch, sh, bits.get(), counters.get())
        if keep_device and compact:
            # the one-byte mask and the planes stay on the device for the record writer's device half (trk_format_samples);
            # the caller gives them back (CallHost.release_device)
            ch.dev = dict(mask8=call.filter_mask8, planes=list(dplanes), stride=int(b.struct.n_samples))
            call.filter_mask8, dplanes = None, []
        self._free(b, *dplanes, call.gt_out, call.filter_mask, call.filter_mask8, call.sample_counters, call.sample_totaldp,
                   call.sample_totaldp_f64,
                   call.sample_dp_missing, call.error, st.allele_count, st.locus_int, st.locus_f64,
                   bits, counters, ext)
        return out

    def assoc_batch(self, hb, vec, sample_in, non_major_cutoff, precision=2, tables=None):
        """associaTR scan of one batch (trk_assoc_scan): vec [M, S] float64 (outcome, covariates),
        sample_in bool[S] or None; ``tables``: (allele_len, rlen_class) already made for the batch
        (synth.assoc_tables_from_classes: the batch pipeline has no per-locus Python lists).  Returns AssocHost."""
        from .synth import pack_assoc_tables
        eng = self.eng
        b = self._upload(hb)
        alen, rcls = tables if tables is not None else pack_assoc_tables(hb.allele_lens, precision)
        sin = None
        if sample_in is not None and not bool(np.all(sample_in)):
            sin = np.ascontiguousarray(sample_in, dtype=np.uint8)
        n_pad = self._n_pad(hb)
        if n_pad:   # padding samples are outside the regression set and carry zeros in every vector
            sin = self._pad(np.ones(hb.gt.shape[1], dtype=np.uint8) if sin is None else sin, n_pad, 0, 0)
        # the trait vectors are the same for every batch of a run: uploaded once and kept, keyed on their CONTENT
        # (shape + checksum -- an array reused with other values, or edited in place, is uploaded again); the small
        # sample mask travels with every call
        import zlib
        vec = np.ascontiguousarray(vec, dtype=np.float64)
        key = (vec.shape, n_pad, zlib.crc32(vec.view(np.uint8).reshape(-1)))
        if getattr(self, '_assoc_vec', (None, None))[0] != key:
            if getattr(self, '_assoc_vec', None) is not None:
                self._free(self._assoc_vec[1])
            self._assoc_vec = (key, eng.upload(np.ascontiguousarray(self._pad(vec, n_pad, 1, 0.0), dtype=np.float64)))
        vec_d = self._assoc_vec[1]
        sin_d = eng.upload(sin) if sin is not None else None
        alen_d, rcls_d = eng.upload(alen), eng.upload(rcls)
        res = eng.assoc_scan(b, vec_d, alen_d, rcls_d, sample_in=sin_d, non_major_cutoff=non_major_cutoff)
        out = AssocHost(res.locus_int.get(), res.locus_f64.get(), res.allele_count.get())
        self._free(b, alen_d, rcls_d, sin_d, res.locus_int, res.locus_f64, res.allele_count)
        return out

    def close(self):
        """Release what the object keeps on the device between calls."""
        if getattr(self, '_assoc_vec', None) is not None:
            self._free(self._assoc_vec[1])
            self._assoc_vec = None

    def assoc_dosage_batch(self, hb, vec, sample_in, ap1, ap2, precision=2):
        """associaTR --beagle-dosages scan of one batch (trk_assoc_scan_dosage): ap1/ap2 [L, S, K] float32.
        Returns (AssocHost, class_sums [sumA, 4], locus_sums [L, 6], (perm, dclass, dclass_value, best_class))."""
        from .synth import pack_assoc_tables, pack_dosage_tables
        eng = self.eng
        b = self._upload(hb, pad=False)     # the AP planes come as they are: per-call kernels, any row length
        alen, rcls = pack_assoc_tables(hb.allele_lens, precision)
        tabs = pack_dosage_tables(hb.allele_lens, precision)
        sin = None
        if sample_in is not None and not bool(np.all(sample_in)):
            sin = np.ascontiguousarray(sample_in, dtype=np.uint8)
        res, cs, ls = eng.assoc_scan_dosage(b, np.ascontiguousarray(vec, dtype=np.float64), alen, rcls,
                                            np.ascontiguousarray(ap1, dtype=np.float32),
                                            np.ascontiguousarray(ap2, dtype=np.float32), *tabs, sample_in=sin)
        out = (AssocHost(res.locus_int.get(), res.locus_f64.get(), res.allele_count.get()), cs.get(), ls.get(), tabs)
        self._free(b, res.locus_int, res.locus_f64, res.allele_count, cs, ls, *[x for x in res._keep if x is not None])
        return out

    def qc_batch(self, hb, quality=None, sample_index=None, ignore_no_call=False):
        """qcSTR's per-record loop (trtools/qcSTR/qcSTR.py:529-561) for a whole batch in one device pass
        (trk_qc_reduce): ``quality`` float32 [L, S] (GetQualityScores of every record, nan = missing),
        ``sample_index`` the boolean sample selection of qcSTR --samples.  Returns a dict of host arrays over ALL
        samples / loci of the batch: sample_calls, locus_calls and, with a plane, sample_qual_sum, sample_qual_n,
        locus_qual_sum, locus_qual_n (the reference's means are sum / n; rows of unselected samples are 0)."""
        eng = self.eng
        b = self._upload(hb, pad=False)
        sin = None if sample_index is None else np.ascontiguousarray(sample_index, dtype=bool).astype(np.uint8)
        res = eng.qc_reduce(b, None if quality is None else np.ascontiguousarray(quality, dtype=np.float32), sin,
                            ignore_no_call)
        keep = res.pop('_keep')
        out = {k: v.get() for k, v in res.items()}
        self._free(b, *res.values(), *[x for x in keep if x is not None])
        return out

    def dosages_batch(self, hb, dosage_type, ap1=None, ap2=None):
        """TRRecord.GetDosages for every record of a batch (trk_dosages): (float32 [L, S], int32 [L] error bits)."""
        from .synth import pack_assoc_tables
        eng = self.eng
        b = self._upload(hb, pad=False)     # the AP planes come as they are: per-call kernels, any row length
        alen, _ = pack_assoc_tables(hb.allele_lens, 2)
        out, err = eng.dosages(b, alen, dosage_type,
                               None if ap1 is None else np.ascontiguousarray(ap1, dtype=np.float32),
                               None if ap2 is None else np.ascontiguousarray(ap2, dtype=np.float32))
        res = (out.get(), err.get())
        self._free(b, out, err, *[x for x in out._keep if x is not None])
        return res
