"""Python face of the native VCF / BGZF reader (include/trk_vcf.h, csrc/trk_vcf.cpp).

``NativeVCFReader`` has the surface of ``vcfio.VCFReader`` (and therefore of the part of
``cyvcf2.VCF`` the hot path uses) but decodes batches of records in C++ threads straight into
numpy arrays: the genotype tensor, the phase bits and any FORMAT field selected with
``select_format``.  Records come out as ``vcfio.Variant`` objects whose genotype matrix and
selected FORMAT arrays are views into those batch arrays; other FORMAT fields are still
available (decoded lazily in Python from the record text)."""
from . import _knobs
import ctypes as C
import os
import sys

from struct import error as struct_error

import numpy as np

from . import _lib
from . import vcfio

KIND_INT, KIND_FLOAT, KIND_INT_RANGES, KIND_MINSUPP = 0, 1, 2, 3
MINSUPP_KEY = '__minsupp'


class _Batch(C.Structure):
    _fields_ = [('n_records', C.c_int32), ('max_ploidy', C.c_int32), ('gt', C.c_void_p),
                ('phased', C.c_void_p), ('locus_ploidy', C.c_void_p), ('planes', C.POINTER(C.c_void_p)),
                ('text', C.c_void_p), ('line_off', C.POINTER(C.c_int64)), ('line_end', C.POINTER(C.c_int64)),
                ('field_off', C.POINTER(C.c_int32)), ('gt_mapped', C.c_void_p)]


class _InflateHook(C.Structure):          # trk_vcf_inflate_hook (include/trk_vcf.h)
    _fields_ = [('user', C.c_void_p), ('seed', C.c_void_p), ('inflate', C.c_void_p), ('max_members', C.c_int32),
                ('submit', C.c_void_p), ('collect', C.c_void_p)]


class _Harmonized(C.Structure):
    _fields_ = [('n_records', C.c_int32), ('n_python', C.c_int32), ('n_alleles_total', C.c_int64),
                ('allele_off', C.POINTER(C.c_int32)), ('len_class', C.POINTER(C.c_uint16)),
                ('str_class', C.POINTER(C.c_uint16)), ('len_class_value', C.POINTER(C.c_double)),
                ('allele_len', C.POINTER(C.c_double)), ('pos', C.POINTER(C.c_int64)), ('end', C.POINTER(C.c_int64)),
                ('passing', C.POINTER(C.c_uint8)), ('status', C.POINTER(C.c_uint8)), ('keys', C.c_void_p),
                ('key_off', C.POINTER(C.c_int64)), ('n_str_classes', C.POINTER(C.c_int32)),
                ('n_len_classes', C.POINTER(C.c_int32)), ('hrun', C.POINTER(C.c_int32)),
                ('period', C.POINTER(C.c_int32)), ('tr_pos', C.POINTER(C.c_int64))]


class _StatRows(C.Structure):
    _fields_ = [('n_groups', C.c_int32), ('precision', C.c_int32), ('use_length', C.c_int32), ('flags', C.c_int32),
                ('allele_count', C.c_void_p), ('locus_int', C.c_void_p), ('locus_f64', C.c_void_p)]


VT_CODES = {'gangstr': 0, 'hipstr': 1, 'longtr': 1, 'advntr': 2, 'eh': 3, 'popstr': 4}
SS_FLAGS = dict(thresh=1, afreq=2, acount=4, nalleles=8, hwep=16, het=32, entropy=64, mean=128, mode=256, var=512,
                numcalled=1024)


def _np(ptr, n, dtype):
    """numpy view of n elements behind a ctypes pointer (memory owned by the reader)."""
    if n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).view(dtype)


class HarmonizedBatch:
    """trk_vcf_harmonize's tables as numpy views (valid until the reader's next batch)."""

    def __init__(self, hz):
        n, sa = hz.n_records, hz.n_alleles_total
        self.struct = hz
        self.n, self.n_python = n, hz.n_python
        self.allele_off = _np(hz.allele_off, n + 1, np.int32)
        self.len_class = _np(hz.len_class, sa, np.uint16)
        self.str_class = _np(hz.str_class, sa, np.uint16)
        self.len_class_value = _np(hz.len_class_value, sa, np.float64)
        self.allele_len = _np(hz.allele_len, sa, np.float64)
        self.pos = _np(hz.pos, n, np.int64)
        self.passing = _np(hz.passing, n, np.uint8)
        self.key_off = _np(hz.key_off, sa + 1, np.int64)
        self.hrun = _np(hz.hrun, n, np.int32)
        self.period = _np(hz.period, n, np.int32)
        self.tr_pos = _np(hz.tr_pos, n, np.int64)
        self.status = _np(hz.status, n, np.uint8)
        self.n_len_classes = _np(hz.n_len_classes, n, np.int32)

    def ref_keys(self):
        """The harmonised (trimmed, upper-cased) reference allele of every record, as str."""
        keys = C.string_at(self.struct.keys, int(self.key_off[-1])) if len(self.key_off) else b''
        at = self.allele_off[:-1].astype(np.int64) + self.str_class[self.allele_off[:-1]] if self.n else np.zeros(0, np.int64)
        lo, hi = self.key_off[at].tolist(), self.key_off[at + 1].tolist()
        return [keys[a:b].decode() for a, b in zip(lo, hi)]

    def lists(self):
        """(allele_lens, allele_strs) per locus as Python lists (tests' oracle stand-in only)."""
        keys = C.string_at(self.struct.keys, int(self.key_off[-1])) if len(self.key_off) else b''
        lens, strs = [], []
        for l in range(self.n):
            o, e = int(self.allele_off[l]), int(self.allele_off[l + 1])
            lens.append([float(x) for x in self.allele_len[o:e]])
            strs.append([keys[self.key_off[o + c]:self.key_off[o + c + 1]].decode() for c in self.str_class[o:e]])
        return lens, strs


class _CfValue(C.Structure):
    _fields_ = [('name', C.c_char_p), ('kind', C.c_int32), ('reserved', C.c_int32), ('plane_a', C.c_void_p),
                ('dtype_a', C.c_int32), ('ncol_a', C.c_int32), ('col_a', C.c_int32), ('col_a2', C.c_int32),
                ('plane_b', C.c_void_p), ('dtype_b', C.c_int32), ('ncol_b', C.c_int32), ('col_b', C.c_int32),
                ('pad_b', C.c_int32)]


class _DumpLines(C.Structure):
    _fields_ = [('n_samples', C.c_int32), ('ploidy', C.c_int32), ('n_filters', C.c_int32), ('n_format_keys', C.c_int32),
                ('n_threads', C.c_int32), ('reserved', C.c_int32), ('gt', C.c_void_p), ('phased', C.c_void_p),
                ('locus_ploidy', C.c_void_p), ('mask8', C.c_void_p), ('mask32', C.c_void_p),
                ('filters', C.POINTER(_CfValue)), ('heads', C.POINTER(C.c_char_p)),
                ('format_keys', C.POINTER(C.c_char_p)), ('format_kinds', C.POINTER(C.c_int32))]


class _DumpRecords(C.Structure):
    _fields_ = [('base', _DumpLines), ('keep', C.c_void_p), ('filter_text', C.POINTER(C.c_char_p)),
                ('hrun', C.c_void_p), ('have_stats', C.c_void_p), ('het', C.c_void_p), ('hwep', C.c_void_p),
                ('allele_count', C.c_void_p), ('allele_off', C.c_void_p), ('n_info_keys', C.c_int32),
                ('fast_path', C.c_int32), ('info_keys', C.POINTER(C.c_char_p)), ('info_kinds', C.POINTER(C.c_int32)),
                ('need_head', C.c_void_p), ('dev_regions', C.c_void_p), ('dev_region_off', C.c_void_p),
                ('dev_region_len', C.c_void_p), ('dev_flags', C.c_void_p),
                ('dev_wait', C.c_void_p), ('dev_wait_arg', C.c_void_p),
                ('dev_emit', C.c_void_p), ('dev_emit_arg', C.c_void_p)]


_EMIT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.c_int64, C.c_void_p)     # trk_vcf_dumpstr2.dev_emit


READER_QUEUE = 3       # the read-ahead thread's queue in device-parse mode (trk_thread_queue)
DEVICE_FORMAT = dict(records=0, left_to_host=0)     # records whose sample columns the device wrote / left to the host writer


class RawBatch:
    """One batch of records as the native reader decoded it: the genotype tensor and FORMAT planes as arrays, the
    record text still inside the reader.  No Python object per record unless ``records()`` is asked for."""

    def __init__(self, reader, b, gt, ph, lp, planes, gt_mapped=None):
        self.reader, self.b = reader, b
        self.n = b.n_records
        self._gt, self._phased, self.locus_ploidy = gt[:self.n], ph[:self.n], lp[:self.n]
        # the genotypes once more in the column order of set_sample_map (None without a map)
        self.gt_mapped = None if gt_mapped is None else gt_mapped[:self.n]
        self._planes = {k: planes[j][:self.n] for j, (k, _, _, _) in enumerate(reader._selected)}
        self._hz = None
        self.dev = None          # device-parsed batch (NativeVCFReader.device_parse): dict(gt=, planes={key: DeviceArray})

    # In device-parse mode the sample columns were parsed by trk_parse_samples and live in HBM (``dev``); the host
    # arrays are filled from them the first time somebody asks (the record writer's decode path, the per-record loop).
    def host_defer(self):
        """The device arrays are about to change hands (dumpSTR gives them to its device batch, which frees them when it is
        done): their MEMORY is held until this batch is released, so that the host copies can still be made if somebody
        asks -- 160 MB per batch at 17 000 x 5000 that only the records the device leaves to the host writer need.
        Returns False (nothing held) when the copies could not be plain ones: the caller copies now (``_host``)."""
        d = self.dev
        if d is None or d.get('on_host') or d.get('deferred') is not None:
            return d is not None and d.get('deferred') is not None
        pairs = [(self._gt, d['gt']), (self._phased, d['phased'])] + [(self._planes[k], a) for k, a in d['planes'].items()]
        if _knobs.lab('TRK_HOST_DEFER', '1') == '0' or not self.n or \
                not all(src is not None and src.ptr is not None and dst.flags['C_CONTIGUOUS'] and dst.nbytes == src.nbytes
                        for dst, src in pairs):
            return False
        d['deferred'] = (d['gt'].eng, [(dst, src.hold(), src.ptr, src.nbytes) for dst, src in pairs])
        return True

    def fetch_text(self):
        """Device-inflate mode (NativeVCFReader.device_inflate): the host's copy of the batch's text holds the heads of the
        lines only.  Whoever reads sample columns on the host -- the host parse of a batch the device flags, the record
        writers' host paths, the per-record objects -- brings the text over first (once): one copy out of the batch's
        text in HBM into the reader's buffer."""
        d = self.dev
        if d is None or not d.get('sparse_text') or d.get('text_on_host') or d.get('text') is None or d['text'].ptr is None:
            return self
        d['text_on_host'] = True
        eng = d['eng']
        eng._chk(eng.lib.trk_memcpy_d2h(eng.ctx, self.b.text + d['text_base'], d['text'].ptr, d['text_nbytes']))
        return self

    def _host(self):
        self.fetch_text()
        d = self.dev
        if d is not None and d.get('deferred') is not None:
            eng, items = d['deferred']
            d['deferred'], d['on_host'] = None, True
            for dst, src, ptr, nbytes in items:
                eng._chk(eng.lib.trk_memcpy_d2h(eng.ctx, dst.ctypes.data, ptr, nbytes))
                src.unhold()
        elif d is not None and not d.get('on_host'):
            d['on_host'] = True
            if self.n:
                eng = d['gt'].eng
                # straight into the reader's (pinned) arrays: the views are the contiguous leading records of them
                for dst, src in [(self._gt, d['gt']), (self._phased, d['phased'])] + \
                                [(self._planes[k], a) for k, a in d['planes'].items()]:
                    if dst.flags['C_CONTIGUOUS'] and dst.nbytes == src.nbytes:
                        eng._chk(eng.lib.trk_memcpy_d2h(eng.ctx, dst.ctypes.data, src.ptr, src.nbytes))
                    else:
                        dst[...] = src.get().reshape(dst.shape)
        return self

    gt = property(lambda self: self._host()._gt)
    phased = property(lambda self: self._host()._phased)
    planes = property(lambda self: self._host()._planes)

    def plane_arrays(self):
        """The host planes as OBJECTS (keys, shapes, addresses): in device-parse mode their contents may not have been
        copied yet (``host_defer``) -- reading values goes through ``planes``."""
        return self._planes

    def release_device(self):
        """Give the device-parsed arrays back (the caller took what it needs, or handed them to a DeviceBatch)."""
        if self.dev is not None and self.dev.get('deferred') is not None:
            for _dst, src, _ptr, _n in self.dev['deferred'][1]:      # nobody asked for the host copies
                src.unhold()
            self.dev['deferred'] = None
        d, self.dev = self.dev, None
        if d is not None:
            for a in [d.get('gt'), d.get('phased'), d.get('text'), d.get('smp_off'), d.get('line_end')] + list(d.get('planes', {}).values()):
                if a is not None and a.ptr is not None:
                    a.free()

    def harmonize(self, vcftype):
        if self._hz is not None and getattr(self, '_hz_type', None) == vcftype:
            return self._hz                  # (the read-ahead thread did it: NativeVCFReader.prefetch_harmonize)
        hz = _Harmonized()
        rc = self.reader._lib.trk_vcf_harmonize(self.reader._h, C.byref(self.b), VT_CODES[vcftype], C.byref(hz))
        if rc != 0:
            raise ValueError("trk_vcf_harmonize failed (%d)" % rc)
        self._hz = HarmonizedBatch(hz)
        self._hz_type = vcftype
        return self._hz

    def chrom_pos(self, l):
        f0, f1 = int(self.b.field_off[l * 10]), int(self.b.field_off[l * 10 + 1])
        return C.string_at(self.b.text + self.b.line_off[l] + f0, f1 - 1 - f0).decode(), int(self._hz.pos[l])

    def statstr_rows(self, st, flags, precision, use_length, skip=None):
        """statSTR's rows for this batch from the host copies of the device results (trk_vcf_statstr_rows)."""
        ac = np.ascontiguousarray(st.allele_count, dtype=np.int32)
        li = np.ascontiguousarray(st.locus_int, dtype=np.int32)
        lf = np.ascontiguousarray(st.locus_f64, dtype=np.float64)
        prm = _StatRows(ac.shape[0], int(precision), 1 if use_length else 0, int(flags), ac.ctypes.data,
                        li.ctypes.data, lf.ctypes.data)
        sk = None if skip is None else np.ascontiguousarray(skip, dtype=np.uint8)
        el, ek = C.c_int32(), C.c_int32()
        cap = max(1 << 16, self.n * 1024)     # (a row with many alleles runs to several hundred bytes)
        lib = self.reader._lib
        while True:
            buf = C.create_string_buffer(cap)
            n = lib.trk_vcf_statstr_rows(C.byref(self.b), C.byref(self._hz.struct), C.byref(prm),
                                         None if sk is None else sk.ctypes.data, buf, cap, C.byref(el), C.byref(ek))
            if n >= 0:
                break
            if n == -(1 << 63):
                raise ValueError("trk_vcf_statstr_rows: bad arguments")
            cap = -n + 64
        return buf.raw[:n], el.value, ek.value

    def _chrom_runs(self):
        """(first record of every run of records with one CHROM, that CHROM as str) -- by array operations on the text (a
        Python loop over the records was 1.4 ms per batch on dumpSTR's caller thread)."""
        n = self.n
        if n == 0:
            return np.zeros(0, np.int64), []
        fo = np.ctypeslib.as_array(self.b.field_off, shape=(n * 10,))
        lo = np.ctypeslib.as_array(self.b.line_off, shape=(n,))
        start = lo.astype(np.int64) + fo[0::10]
        ln = (fo[1::10] - 1 - fo[0::10]).astype(np.int64)
        L = max(int(ln.max()), 1)
        size = int((start + ln).max()) + 1
        text = np.frombuffer((C.c_uint8 * size).from_address(int(self.b.text)), dtype=np.uint8)
        col = np.arange(L, dtype=np.int64)[None, :]
        by = np.where(col < ln[:, None], text[np.minimum(start[:, None] + col, size - 1)], 0)
        change = np.ones(n, dtype=bool)
        change[1:] = (ln[1:] != ln[:-1]) | (by[1:] != by[:-1]).any(axis=1)
        first = np.flatnonzero(change)
        return first, [bytes(by[l, :ln[l]]).decode() for l in first]

    def chroms(self):
        """The distinct CHROM values of the batch, in order of first appearance."""
        out = []
        for d in self._chrom_runs()[1]:
            if d not in out:
                out.append(d)
        return out

    def chrom_column(self):
        """CHROM of every record of the batch (list of str)."""
        first, names = self._chrom_runs()
        out = []
        for k, d in enumerate(names):
            out.extend([d] * (int(first[k + 1] if k + 1 < len(first) else self.n) - int(first[k])))
        return out

    def format_columns(self):
        """The distinct FORMAT columns of the batch's records, as lists of keys (normally one).  Vectorised: records
        whose FORMAT bytes equal the first record's are recognised without touching Python strings."""
        b, n = self.b, self.n
        if n == 0:
            return []
        fo = np.ctypeslib.as_array(b.field_off, shape=(n * 10,)).reshape(n, 10)
        lo = np.ctypeslib.as_array(b.line_off, shape=(n,))
        le = np.ctypeslib.as_array(b.line_end, shape=(n,))
        start = lo + fo[:, 8]
        # field_off[9] is one past the tab that ends FORMAT (parse_record's own rule for the column's end)
        end = np.where(fo[:, 9] > fo[:, 8], lo + fo[:, 9] - 1, le)
        ln = np.maximum(end - start, 0)
        first = C.string_at(b.text + int(start[0]), int(ln[0])) if ln[0] > 0 else b''
        same = ln == ln[0]
        if ln[0] > 0 and same.any():
            text = np.ctypeslib.as_array(C.cast(b.text, C.POINTER(C.c_uint8)), shape=(int(le.max()),))
            ref = np.frombuffer(first, dtype=np.uint8)
            idx = np.flatnonzero(same)
            eq = (text[start[idx, None] + np.arange(ref.size)[None, :]] == ref[None, :]).all(axis=1)
            same[idx] = eq
        out = [first.decode().split(':') if first else []]
        seen = {first}
        for l in np.flatnonzero(~same):
            t = C.string_at(b.text + int(start[l]), int(ln[l])) if ln[l] > 0 else b''
            if t not in seen:
                seen.add(t)
                out.append(t.decode().split(':') if t else [])
        return out

    _MOTIF_KEY = {'gangstr': 'RU', 'advntr': 'RU', 'eh': 'RU', 'popstr': 'Motif'}

    def motifs(self, hz, vcftype):
        """TRRecord.motif of every record of the batch, as the harmonisers derive it (utils/tr_harmonizer.py): HipSTR / LongTR
        infer it from the trimmed reference allele -- sliced by the leading flank once more, as the reference does
        (tr_harmonizer.py:397) -- and INFO/PERIOD; the other callers carry it in INFO.  ``hz``: this batch's harmonised
        tables, ``vcftype``: the caller's name."""
        from .utils import utils
        if vcftype in ('hipstr', 'longtr'):
            lead = (hz.tr_pos - hz.pos).tolist()
            per = hz.period.tolist()
            return [utils.InferRepeatSequence(ref[ld:], p) for ref, ld, p in zip(hz.ref_keys(), lead, per)]
        key = self._MOTIF_KEY[vcftype] + '='
        out = []
        for l in range(self.n):
            info = self.head_fields(l)[7]
            val = next(item[len(key):] for item in info.split(';') if item.startswith(key))
            out.append(val.upper())
        return out

    def head_fields(self, l):
        """The eight leading columns and the FORMAT column of record l as text."""
        b = self.b
        f9 = int(b.field_off[l * 10 + 9])
        n_line = b.line_end[l] - b.line_off[l]
        end = f9 - 1 if 0 < f9 < n_line else n_line
        return C.string_at(b.text + b.line_off[l], end).decode().split('\t')

    def dumpstr_lines(self, heads, mask, cf_values, format_kinds, n_threads=0, out_ring=None, native=None):
        """trk_vcf_dumpstr_lines: the output records of the batch.  heads: list of str / None; mask: uint8 or uint32
        [n, S]; cf_values: list of (name, kind, (plane, col), (plane, col) | None); format_kinds: {FORMAT ID: 1 int /
        2 float / 4 string}.  Returns bytes, or None when a record is outside what the native writer covers.
        ``out_ring``: a dict the caller keeps for the run -- the output lands in one of TWO buffers held there, taken in
        turn, instead of a fresh array per batch (a fresh 100 MB array is 25 000 page faults inside the formatter's
        threads, which serialise in the kernel).  The caller must be done with a result before the call after the
        next one (dumpSTR's writer holds at most one block in flight).
        ``native`` (trk_vcf_dumpstr_records): the record heads are built by the library -- a dict with keep (uint8 [n]
        or None), filter_text (list of bytes / None per record, or None), hrun (int32 [n]), have_stats (uint8 [n]),
        het / hwep (float64 [n]), allele_count (int32), allele_off (int32 [n + 1]), info_types ({ID: (Type, Number)}),
        py_head (callable l -> str: the head of a record whose INFO column the native rewrite declines); ``heads`` is
        then ignored."""
        # (device-parse mode: the host copies may have been deferred -- host_defer; only the arrays' addresses are taken
        # here, and the copies are made below unless the device writes the columns of every record)
        S, P = self._gt.shape[1], self._gt.shape[2]
        gt = np.ascontiguousarray(self._gt)
        ph = np.ascontiguousarray(self._phased)
        lp = np.ascontiguousarray(self.locus_ploidy)
        mask = np.ascontiguousarray(mask)
        keep = [gt, ph, lp, mask]
        fv = (_CfValue * max(len(cf_values), 1))()
        for k, (name, kind, a, bsrc) in enumerate(cf_values):
            # kind 0: plane a, column a[1]; 1: a / b; 2: columns a[1] + a[2] of plane a; 3: GangSTR bad CI (a = REPCN,
            # b = the pre-parsed REPCI)
            pa = np.ascontiguousarray(a[0])
            keep.append(pa)
            fv[k].name, fv[k].kind = name.encode(), int(kind)
            fv[k].plane_a, fv[k].dtype_a = pa.ctypes.data, 1 if pa.dtype == np.float32 else 0
            fv[k].ncol_a, fv[k].col_a = (pa.shape[2] if pa.ndim == 3 else 1), int(a[1])
            fv[k].col_a2 = int(a[2]) if len(a) > 2 else 0
            if bsrc is not None:
                pb = np.ascontiguousarray(bsrc[0])
                keep.append(pb)
                fv[k].plane_b, fv[k].dtype_b = pb.ctypes.data, 1 if pb.dtype == np.float32 else 0
                fv[k].ncol_b, fv[k].col_b = (pb.shape[2] if pb.ndim == 3 else 1), int(bsrc[1])
        if native is None:
            harr = (C.c_char_p * max(self.n, 1))(*[None if h is None else h.encode() for h in heads])
        else:
            harr = None
        fk = list(format_kinds.items())
        karr = (C.c_char_p * max(len(fk), 1))(*[k.encode() for k, _ in fk])
        kk = (C.c_int32 * max(len(fk), 1))(*[int(v) for _, v in fk])
        prm = _DumpLines(S, P, len(cf_values), len(fk), int(n_threads), 0, gt.ctypes.data, ph.ctypes.data,
                         lp.ctypes.data, mask.ctypes.data if mask.dtype == np.uint8 else None,
                         mask.ctypes.data if mask.dtype == np.uint32 else None, fv, harr, karr, kk)
        lib = self.reader._lib
        err = C.c_int32()
        ext = regions = None
        if native is not None:
            arrs = {k: (None if native.get(k) is None else np.ascontiguousarray(native[k], dtype=dt))
                    for k, dt in (('keep', np.uint8), ('hrun', np.int32), ('have_stats', np.uint8), ('het', np.float64),
                                  ('hwep', np.float64), ('allele_count', np.int32), ('allele_off', np.int32))}
            keep.extend(arrs.values())
            ft = native.get('filter_text')
            ftarr = None if ft is None else (C.c_char_p * max(self.n, 1))(*ft)
            ik = list((native.get('info_types') or {}).items())
            kind_of = {'Integer': 1, 'Float': 2, 'Flag': 3}
            ikeys = (C.c_char_p * max(len(ik), 1))(*[k.encode() for k, _ in ik])
            ikinds = (C.c_int32 * max(len(ik), 1))(*[kind_of.get(t[0], 0) for _, t in ik])
            need = np.zeros(max(self.n, 1), dtype=np.uint8)
            ext = _DumpRecords(prm, None if arrs['keep'] is None else arrs['keep'].ctypes.data, ftarr,
                               arrs['hrun'].ctypes.data, arrs['have_stats'].ctypes.data, arrs['het'].ctypes.data,
                               arrs['hwep'].ctypes.data, arrs['allele_count'].ctypes.data, arrs['allele_off'].ctypes.data,
                               len(ik), 1, ikeys, ikinds, need.ctypes.data)
            keep.extend([ftarr, ikeys, ikinds, need])
            regions = self._device_regions(prm, mask, cf_values, S, out_ring, native.get('dev_call'), native.get('cf_plane_idx'))
            if regions is not None and regions.get('emit') is not None:
                # whole-record emit: the writer lays the batch out and calls back; the device puts the columns in place
                ext.dev_region_len, ext.dev_flags = regions['len'].ctypes.data, regions['flags'].ctypes.data
                ext.dev_emit = C.cast(regions['emit'], C.c_void_p).value
                keep.append(regions)
            elif regions is not None:
                ext.dev_regions, ext.dev_region_off = regions['buf'].ctypes.data, regions['off'].ctypes.data
                ext.dev_region_len, ext.dev_flags = regions['len'].ctypes.data, regions['flags'].ctypes.data
                if regions.get('wait') is not None:     # the download is in flight: the writer waits after the heads
                    ext.dev_wait, ext.dev_wait_arg = regions['wait']
                keep.append(regions)
        if regions is None or regions['flags'].any() or _knobs.lab('TRK_FMT_FAST', '1') == '0':
            self._host()        # the writer reads genotypes / values of some record: they have to be here
        # output bound: the input text, a third more for re-serialised numbers, and per call the FILTER column this
        # pass appends (':PASS' / ':NOCALL'; the longer '<name>_<value>' strings of the few filtered calls fit in
        # the slack).  The buffer is not touched beyond what is written, so a generous bound costs nothing -- a
        # tight one (x 1.3 alone) made the writer run twice on every batch whose calls mostly pass.
        cap = (int((int(self.b.line_end[self.n - 1]) - int(self.b.line_off[0])) * 1.35) + self.n * S * 12 + (1 << 16)
               if self.n else 16)
        slot = None
        if out_ring is not None:
            slot = out_ring['i'] = (out_ring.get('i', -1) + 1) % 2
        emit = regions is not None and regions.get('emit') is not None
        if emit:
            cap = int(regions['cap'])      # (a tight bound: the block is pinned memory, the copy's destination)
        try:
            while True:
                if emit:
                    ring, key = (out_ring if out_ring is not None else {}), ('emit', slot)
                    buf = ring.get(key)
                    if buf is None or buf.size < cap:
                        alloc = getattr(self.reader, '_alloc', None)     # (pinned where the reader has an engine's allocator)
                        buf = ring[key] = (alloc(cap + cap // 16 + (1 << 20)) if alloc is not None
                                           else np.empty(cap + cap // 16 + (1 << 20), dtype=np.uint8))
                        if alloc is not None:
                            self.reader._slabs = getattr(self.reader, '_slabs', []) + [buf]
                elif slot is None:
                    buf = np.empty(cap, dtype=np.uint8)      # not zero-filled; handed to the writer as a memoryview
                else:
                    buf = out_ring.get(slot)
                    if buf is None or buf.size < cap:
                        buf = out_ring[slot] = np.empty(cap + cap // 8, dtype=np.uint8)
                if ext is None:
                    n = lib.trk_vcf_dumpstr_lines(C.byref(self.b), C.byref(prm), buf.ctypes.data, cap, C.byref(err))
                else:
                    n = lib.trk_vcf_dumpstr_records(C.byref(self.b), C.byref(ext), buf.ctypes.data, cap, C.byref(err))
                if n >= 0:
                    return memoryview(buf)[:n]
                if emit and regions.get('error') is not None:
                    raise regions['error']
                if ext is not None and n == -(1 << 63) + 2:
                    # INFO columns the native rewrite leaves to Python: those heads come from the caller, once
                    if ext.base.heads:
                        return None
                    todo = np.flatnonzero(need[:self.n])
                    hl = [None] * max(self.n, 1)
                    for l in todo:
                        hl[int(l)] = native['py_head'](int(l)).encode()
                    harr = (C.c_char_p * max(self.n, 1))(*hl)
                    keep.append(harr)
                    ext.base.heads = harr
                    need[:] = 0
                    continue
                if n <= -(1 << 63) + 1:
                    return None
                cap = -n + 64
        finally:
            self._release_regions(regions)     # (the columns' download has completed; device buffers back)

    def _device_regions(self, prm, mask, cf_values, S, out_ring, dev_call=None, cf_plane_idx=None):
        """The sample columns of the batch's output records written ON THE DEVICE (trk_format_samples; round 4): needs the
        batch's text there (device-parse mode), the one-byte mask and plain-value filters.  Returns host arrays
        (buf, off, len, flags) for trk_vcf_dumpstr2.dev_*, or None (the host writer does everything).
        On by default where the batch was parsed on the device (TRK_DEVICE_FORMAT=0: off): byte-identical output; with the
        call-filter pass's mask and planes kept on the device 0.245-0.28 s against 0.28-0.30 on the 1 GB command line
        (profiles/r04_notes.md section 16)."""
        d = self.dev
        if (d is None or d.get('text') is None or _knobs.env('TRK_DEVICE_FORMAT', '1') != '1' or self.n == 0 or
                mask.dtype != np.uint8 or len(cf_values) > _lib.FORMAT_MAX_FILTERS or
                any(kind != 0 or bsrc is not None or int(a[1]) != 0 or len(name.encode()) > 31 or
                    (np.asarray(a[0]).ndim == 3 and np.asarray(a[0]).shape[2] != 1) for name, kind, a, bsrc in cf_values)):
            return None
        eng, n, lib = d['eng'], self.n, self.reader._lib
        import time as _t
        _tm = [_t.perf_counter()]
        kinds16, nf = np.zeros((n, 16), np.uint8), np.zeros(n, np.uint8)
        if lib.trk_vcf_format_kinds(C.byref(self.b), C.byref(prm), kinds16.ctypes.data, nf.ctypes.data) != 0 or not nf.any():
            return None
        tmp = []

        def up(x, dt):
            a = eng.upload(np.ascontiguousarray(x, dtype=dt), dt)
            tmp.append(a)
            return a
        fin = _lib.FormatIn()
        fin.text, fin.n_bytes = d['text'].ptr, d['text'].nbytes
        # the mask and the value planes: the call-filter pass's own device arrays when the caller kept them (rows padded
        # to `stride` samples), else uploaded again
        on_dev = dev_call is not None and cf_plane_idx is not None and len(cf_plane_idx) == len(cf_values)
        stride = int(dev_call['stride']) if on_dev else S
        fin.n_records, fin.n_samples, fin.mask_stride, fin.plane_stride = n, S, stride, stride
        fin.smp_off, fin.line_end = d['smp_off'].ptr, d['line_end'].ptr
        fin.field_kind, fin.n_fields = up(kinds16, np.uint8).ptr, up(nf, np.uint8).ptr
        fin.ploidy = up(self.locus_ploidy, np.uint8).ptr
        fin.mask8 = dev_call['mask8'].ptr if on_dev else up(mask, np.uint8).ptr
        fin.n_filters = len(cf_values)
        for k, (name, kind, a, bsrc) in enumerate(cf_values):
            pa = np.asarray(a[0])
            fin.filter_name[k].value = name.encode()
            fin.filter_plane[k] = dev_call['planes'][cf_plane_idx[k]].ptr if on_dev else up(pa.reshape(n, S), pa.dtype).ptr
            fin.filter_dtype[k] = 1 if pa.dtype == np.float32 else 0
        rec_len, flags = eng.empty((n,), np.uint32), eng.empty((n,), np.uint8)
        tmp += [rec_len, flags]
        fout = _lib.FormatOut()
        fout.rec_len, fout.flags = rec_len.ptr, flags.ptr
        try:
            _tm.append(_t.perf_counter())
            eng._chk(eng.lib.trk_format_samples(eng.ctx, C.byref(fin), C.byref(fout), 1))
            ln, fl = rec_len.get(), flags.get()
            _tm.append(_t.perf_counter())
            off = np.zeros(n, np.int64)
            np.cumsum(ln[:-1], out=off[1:])
            total = int(off[-1] + ln[-1])
            if total == 0:
                return None
            if (out_ring is not None and getattr(self.reader, '_alloc', None) is not None and
                    _knobs.lab('TRK_FMT_EMIT', '1') == '1'):
                # WHOLE-RECORD EMIT (round 5): pass 2 waits until the writer has built the heads and laid the batch out
                # (trk_vcf_dumpstr2.dev_emit); it then writes every record's columns at their FINAL offsets and one copy
                # brings the block to the pinned output buffer -- no gather of heads and columns on the host
                res = dict(buf=None, off=None, len=np.ascontiguousarray(ln, dtype=np.uint32),
                           flags=np.ascontiguousarray(fl, dtype=np.uint8), wait=None, held=(eng, tmp), error=None)
                held_tmp, tmp = tmp, []

                def emit_cb(_arg, rec_off_p, total_out, out_p):
                    try:
                        ro = np.ctypeslib.as_array(rec_off_p, shape=(n,)).copy()
                        ro[ro < 0] = 0
                        ro_d = eng.upload(ro, np.int64)
                        out_dd = eng.empty((int(total_out) + 16,), np.uint8)
                        held_tmp.extend([ro_d, out_dd])
                        fout.out, fout.out_off = out_dd.ptr, ro_d.ptr
                        eng._chk(eng.lib.trk_format_samples(eng.ctx, C.byref(fin), C.byref(fout), 2))
                        eng._chk(eng.lib.trk_memcpy_d2h(eng.ctx, out_p, out_dd.ptr, int(total_out)))
                        return 0
                    except Exception as e:      # (a ctypes callback must not raise: kept for the caller)
                        res['error'] = e
                        return 1
                res['emit'] = _EMIT_FN(emit_cb)
                res['keep'] = (fin, fout)
                # bound of the block: the columns, the heads (as read, doubled, + the INFO updates) and what the host
                # writer may make of the records the device left to it
                fo9 = np.ctypeslib.as_array(self.b.field_off, shape=(n, 10))[:, 9].astype(np.int64)
                ll = (np.ctypeslib.as_array(self.b.line_end, shape=(n,)) - np.ctypeslib.as_array(self.b.line_off, shape=(n,))).astype(np.int64)
                host = fl != 0
                res['cap'] = (int(ln[~host].sum()) + 2 * int(fo9.sum()) + 768 * n + int((ll[host] * 1.35).sum()) +
                              int(host.sum()) * (12 * S + 64) + (1 << 16))
                DEVICE_FORMAT['records'] += int((fl == 0).sum())
                DEVICE_FORMAT['left_to_host'] += int((fl != 0).sum())
                DEVICE_FORMAT['emitted'] = DEVICE_FORMAT.get('emitted', 0) + int((fl == 0).sum())
                return res
            out_d = eng.empty((total + 16,), np.uint8)
            tmp.append(out_d)
            fout.out, fout.out_off = out_d.ptr, up(off, np.int64).ptr
            eng._chk(eng.lib.trk_format_samples(eng.ctx, C.byref(fin), C.byref(fout), 2))
            eng.sync(); _tm.append(_t.perf_counter())
            # back into one of two pinned buffers kept for the run (the writer reads them before the batch after the next)
            ring = out_ring if out_ring is not None else {}
            key = ('dev', (ring.get('i', -1) + 1) % 2)      # (the slot dumpstr_lines moves to for THIS batch)
            buf = ring.get(key)
            if buf is None or buf.size < total:
                alloc = getattr(self.reader, '_alloc', None)
                buf = alloc(int(total * 1.25) + (1 << 20)) if alloc is not None else np.empty(int(total * 1.25) + (1 << 20), np.uint8)
                if alloc is not None:
                    self.reader._slabs = getattr(self.reader, '_slabs', []) + [buf]
                ring[key] = buf
            wait = held = None
            if _knobs.lab('TRK_FMT_ASYNC', '1') == '1' and getattr(self.reader, '_alloc', None) is not None:
                # (pinned destination: the copy runs beside the writer's head building; the writer waits through
                # dev_wait before it reads the columns, the device buffers are released after the writer's call)
                eng._chk(eng.lib.trk_memcpy_d2h_async(eng.ctx, buf.ctypes.data, out_d.ptr, total))
                wait = (C.cast(eng.lib.trk_sync, C.c_void_p).value, eng.ctx)
                held, tmp = (eng, tmp), []
            else:
                eng._chk(eng.lib.trk_memcpy_d2h(eng.ctx, buf.ctypes.data, out_d.ptr, total))
            _tm.append(_t.perf_counter())
            if _knobs.lab('TRK_FMT_TIMING'):
                print('[device format] setup %.1f ms, pass 1 %.1f, alloc + pass 2 %.1f, download of %.0f MB %.1f%s' % tuple(
                    [(b_ - a_) * 1e3 for a_, b_ in zip(_tm[:3], _tm[1:4])] + [total / 1e6, (_tm[4] - _tm[3]) * 1e3,
                                                                          ' (enqueued)' if wait else '']), file=sys.stderr)
        finally:
            if tmp:
                eng.sync()
                for a in tmp:
                    a.free()
        DEVICE_FORMAT['records'] += int((fl == 0).sum())
        DEVICE_FORMAT['left_to_host'] += int((fl != 0).sum())
        return dict(buf=buf, off=off, len=np.ascontiguousarray(ln, dtype=np.uint32), flags=np.ascontiguousarray(fl, dtype=np.uint8),
                    wait=wait, held=held)

    @staticmethod
    def _release_regions(regions):
        """After the writer's call (however it ended): the download has completed, the device buffers go back."""
        held = regions.get('held') if regions else None
        if held:
            regions['held'] = None
            eng, arrs = held
            eng.sync()
            for a in arrs:
                a.free()

    def iter_variants(self):
        """The batch's vcfio.Variant objects one at a time (a malformed line raises at ITS turn, as the per-record
        reader does)."""
        r = self.reader
        for line, g, native, tail in r._rows_of(self):
            v = vcfio.Variant(r, line, gt=g if r.n_samples else None, native=native, tail=tail)
            if v.CHROM not in r.contigs_declared and v.CHROM not in r.contigs_seen:
                r.contigs_seen.append(v.CHROM)
            yield v

    def records(self):
        """vcfio.Variant objects of the batch (the per-record path)."""
        out = [vcfio.Variant(self.reader, line, gt=g if self.reader.n_samples else None, native=native, tail=tail)
               for line, g, native, tail in self.reader._rows_of(self)]
        r = self.reader
        for v in out:          # contigs the file uses without declaring them (the writer adds their header lines)
            if v.CHROM not in r.contigs_declared and v.CHROM not in r.contigs_seen:
                r.contigs_seen.append(v.CHROM)
        return out


def dumpstr_writer_stats():
    """{'fast', 'decoded', 'caller_heads'}: output records written so far without decoding their sample columns / through
    the decode path / with a head built in Python (trk_vcf_dumpstr_stats)."""
    lib = _api()
    a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
    lib.trk_vcf_dumpstr_stats(C.byref(a), C.byref(b), C.byref(c))
    return dict(fast=a.value, decoded=b.value, caller_heads=c.value)


def _api():
    lib = _lib.load()
    if not getattr(lib, '_vcf_ready', False):
        vp = C.c_void_p
        lib.trk_vcf_open.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
        lib.trk_vcf_close.argtypes = [vp]
        lib.trk_vcf_close.restype = None
        lib.trk_vcf_last_error.argtypes = [vp]
        lib.trk_vcf_last_error.restype = C.c_char_p
        lib.trk_vcf_header.argtypes = [vp, C.POINTER(C.c_size_t)]
        lib.trk_vcf_header.restype = C.c_void_p
        lib.trk_vcf_set_sample_map.argtypes = [vp, vp, C.c_int32]
        lib.trk_vcf_n_samples.argtypes = [vp]
        lib.trk_vcf_sample_name.argtypes = [vp, C.c_int]
        lib.trk_vcf_sample_name.restype = C.c_char_p
        lib.trk_vcf_select_format.argtypes = [vp, C.c_char_p, C.c_int, C.c_int]
        lib.trk_vcf_read_batch.argtypes = [vp, C.c_int, C.c_int, C.POINTER(_Batch)]
        lib.trk_vcf_format_kinds.argtypes = [C.POINTER(_Batch), C.POINTER(_DumpLines), vp, vp]
        lib.trk_vcf_skip_samples.argtypes = [vp, C.c_int]
        lib.trk_vcf_set_inflate_hook.argtypes = [vp, C.POINTER(_InflateHook)]
        lib.trk_vcf_text_abs.argtypes = [vp]
        lib.trk_vcf_text_abs.restype = C.c_uint64
        lib.trk_vcf_set_text_buffers.argtypes = [vp, vp, vp, C.c_size_t]
        lib.trk_vcf_format_idx.argtypes = [vp, C.POINTER(C.c_int32)]
        lib.trk_vcf_format_idx.restype = C.c_void_p
        lib.trk_vcf_parse_samples.argtypes = [vp, C.POINTER(_Batch)]
        lib.trk_vcf_seek.argtypes = [vp, C.c_uint64]
        lib.trk_vcf_shard.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        lib.trk_vcf_counters.argtypes = [vp, C.POINTER(C.c_uint64)]
        lib.trk_vcf_counters.restype = None
        lib.trk_vcf_harmonize.argtypes = [vp, C.POINTER(_Batch), C.c_int, C.POINTER(_Harmonized)]
        lib.trk_vcf_statstr_rows.argtypes = [C.POINTER(_Batch), C.POINTER(_Harmonized), C.POINTER(_StatRows), vp,
                                             C.c_char_p, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        lib.trk_vcf_statstr_rows.restype = C.c_int64
        lib.trk_vcf_dumpstr_lines.argtypes = [C.POINTER(_Batch), C.POINTER(_DumpLines), vp, C.c_int64,
                                              C.POINTER(C.c_int32)]
        lib.trk_vcf_dumpstr_lines.restype = C.c_int64
        lib.trk_vcf_dumpstr_records.argtypes = [C.POINTER(_Batch), C.POINTER(_DumpRecords), vp, C.c_int64,
                                                C.POINTER(C.c_int32)]
        lib.trk_vcf_dumpstr_records.restype = C.c_int64
        lib.trk_vcf_dumpstr_stats.argtypes = [C.POINTER(C.c_int64)] * 3
        lib.trk_vcf_dumpstr_stats.restype = None
        lib._vcf_ready = True
    return lib


class NativeVCFReader(vcfio.VCFReader):
    """vcfio.VCFReader whose record decoding runs in the native reader."""

    def __init__(self, path, lazy=False, samples=None, n_threads=0, batch_records=None, max_ploidy=2):
        if not os.path.exists(path) or os.path.isdir(path):
            raise OSError("no such VCF: %s" % path)
        self._lib = _api()
        h = C.c_void_p()
        rc = self._lib.trk_vcf_open(path.encode(), int(n_threads), C.byref(h))
        if rc != 0:
            raise OSError(self._lib.trk_vcf_last_error(None).decode())
        self._h = h
        n = C.c_size_t()
        ptr = self._lib.trk_vcf_header(h, C.byref(n))
        text = C.string_at(ptr, n.value).decode()
        self.path = path
        self._fh = None
        self._header_lines = []
        self.samples = []
        self.info_types, self.format_types = {}, {}
        self.contigs_declared, self.contigs_seen = set(), []
        self.has_pass_filter = False
        for line in text.split('\n'):
            line = line.rstrip('\r')
            if line.startswith('##'):
                self._header_lines.append(line)
                self._register(line)
            elif line.startswith('#CHROM'):
                self._chrom_line = line
                self.samples = line.split('\t')[9:]
        self.n_samples = len(self.samples)
        self._region = None
        self._indexed_region = False
        self._selected = []          # (key, kind, ncol, dtype)
        self._max_ploidy = max_ploidy
        self._batch_records = batch_records
        self._rows = []
        self._row_i = 0
        self._eof = False

    def select_format(self, key, kind=None, ncol=1, alias=None):
        """Decode FORMAT field ``key`` natively (as ``alias`` in Variant.format when given, so that
        the original text stays available under ``key``).  kind defaults to the header's Type."""
        name = alias or key
        if any(k == name for k, _, _, _ in self._selected):
            return
        if kind is None:
            typ = self.format_types.get(key, ('String', '1'))[0]
            if typ == 'Integer':
                kind = KIND_INT
            elif typ == 'Float':
                kind = KIND_FLOAT
            else:
                raise ValueError("FORMAT field %s is not numeric; give an explicit kind" % key)
        rc = self._lib.trk_vcf_select_format(self._h, key.encode(), int(kind), int(ncol))
        if rc < 0:
            raise ValueError("cannot select FORMAT field %s" % key)
        self._selected.append((name, kind, ncol, np.float32 if kind == KIND_FLOAT else np.int32))

    def read_ahead(self, on=True):
        """Read batch n + 1 on a worker thread while the caller works on batch n (the native call releases the GIL;
        libtrk keeps a batch's text and line tables valid during the next read).  With a ring of array sets
        (``use_buffers``) the ring must hold two sets."""
        self._drop_pending()
        self._ahead = bool(on)
        return self

    def prefetch_harmonize(self, vcftype):
        """With read-ahead on, the reader's thread also harmonises the batch it read (trk_vcf_harmonize keeps two result
        sets: batch n's tables stay valid while batch n + 1 is harmonised).  Measured neutral on the 1 GB command lines
        (the reader's thread becomes the longer one: dumpSTR 0.30, statSTR 0.15-0.16 s either way) and, in round 5 with the
        inflate on the device and that thread the short one, slower (statSTR 0.115-0.123 against 0.103-0.117 s): not
        switched on."""
        self._prefetch_hz = vcftype
        return self

    def _drop_pending(self):
        """Wait for a read in flight and forget its batch (before a seek / shard / close)."""
        pend, self._pending = getattr(self, '_pending', None), None
        if pend is not None:
            pend[0].join()
            rb = pend[1].get('rb')
            if rb is not None:           # the batch nobody will take: its device-parsed arrays go back to the engine
                rb.release_device()

    def read_raw_batch(self, n_records=None):
        """Decode the next batch of records into arrays (RawBatch); ``.n == 0`` at the end of the file."""
        if not getattr(self, '_ahead', False):
            return self._read_raw_batch(n_records)
        import threading

        def start():
            box = {}

            def work():
                try:
                    eng = getattr(self, '_dev_eng', None)
                    if eng is not None and _knobs.lab('TRK_READER_QUEUE', '1') != '0':
                        # this thread's uploads and parse kernel on a queue of its own: they no longer sit in front of
                        # the caller's kernels (every batch is handed over after a blocking copy on that queue)
                        eng.thread_queue(READER_QUEUE)
                    rb = self._read_raw_batch(n_records)
                    if rb.n and getattr(self, '_prefetch_hz', None):
                        try:
                            rb.harmonize(self._prefetch_hz)      # (errors are the caller's to meet, on its own call)
                        except Exception:
                            rb._hz = None
                    box['rb'] = rb
                except BaseException as e:      # re-raised on the caller's thread
                    box['err'] = e
            t = threading.Thread(target=work, name='trk-vcf-read-ahead', daemon=True)
            t.start()
            return t, box
        if getattr(self, '_pending', None) is None:
            self._pending = start()
        t, box = self._pending
        t.join()
        self._pending = None
        if 'err' in box:
            raise box['err']
        rb = box['rb']
        if rb.n:
            self._pending = start()
        return rb

    def _pin_text(self):
        """Host-inflate mode WITHOUT read-ahead, from the second batch on: the reader's two text buffers continue in pinned
        memory of the engine's (trk_vcf_set_text_buffers), so that the upload of a batch's text is a plain DMA.  Sized from
        the first batch; a later batch that outgrows them moves the reader back to its own memory.  NOT in the default
        configuration: with read-ahead the previous batch's text is still in use when the next read starts (the move
        would pull it from under the caller), and with the device inflate (round 5, the default on bgzip'ed files) a
        batch's text never is on the host -- the compressed bytes are what crosses PCIe."""
        need = getattr(self, '_text_seen', 0)
        if not need or getattr(self, '_text_pinned', False) or getattr(self, '_alloc', None) is None or \
                getattr(self, '_ahead', False) or getattr(self, '_inflate_on', False):
            return
        self._text_pinned = True
        cap = (int(need * 1.25) + (64 << 20) + 4095) & ~4095
        try:
            a, b = self._alloc(cap), self._alloc(cap)
        except Exception:
            return
        self._slabs = getattr(self, '_slabs', []) + [a, b]
        self._lib.trk_vcf_set_text_buffers(self._h, a.ctypes.data, b.ctypes.data, cap)

    def _read_raw_batch(self, n_records=None):
        S = self.n_samples
        P = self._max_ploidy
        if getattr(self, '_dev_eng', None) is not None:
            self._pin_text()
        n = n_records or self._batch_records or max(1, min(4096, (1 << 22) // max(S, 1)))
        while True:
            # (a batch retried with a wider tensor gets arrays of its own: the ring keeps the file's usual shape)
            gt, ph, lp, planes, gtm = self._arrays(n, S, P, ring=(P == self._max_ploidy))
            parr = (C.c_void_p * max(len(planes), 1))(*[p.ctypes.data for p in planes])
            b = _Batch()
            b.gt, b.phased, b.locus_ploidy = gt.ctypes.data, ph.ctypes.data, lp.ctypes.data
            b.gt_mapped = None if gtm is None else gtm.ctypes.data
            b.planes = C.cast(parr, C.POINTER(C.c_void_p))
            rc = self._lib.trk_vcf_read_batch(self._h, n, P, C.byref(b))
            if rc == 5 and b'haplotypes' in self._lib.trk_vcf_last_error(self._h):
                # a genotype with more haplotypes than the tensor has columns (0/1/1 in a file read as diploid):
                # the reader has consumed nothing, decode the same lines again with a wider tensor (cyvcf2 sizes
                # its genotype array per record; the kernels take any ploidy up to 8)
                # Only THIS batch is decoded wider: one 0/1/1 genotype must not make every later batch of the file
                # twice as wide.  The device takes ploidy <= 8 (TRK_MAX_PLOIDY).
                if P >= 8:
                    raise ValueError("%s: %s -- more than 8 haplotypes per genotype are not supported"
                                     % (self.path, self._lib.trk_vcf_last_error(self._h).decode()))
                P = min(8, 2 * P)
                continue
            if rc != 0:
                raise ValueError(self._lib.trk_vcf_last_error(self._h).decode())
            break
        dev = None
        if getattr(self, '_dev_eng', None) is not None and b.n_records:
            dev, gt, ph, lp, planes, gtm, parr = self._parse_on_device(b, P, (gt, ph, lp, planes, gtm, parr))
        rb = RawBatch(self, b, gt, ph, lp, planes, gtm)
        rb._keep = parr
        rb.dev = dev
        return rb

    def device_parse(self, engine):
        """The sample columns of every batch parsed ON THE DEVICE (trk_parse_samples, include/trk.h; round 4): the native
        reader stops at the FORMAT keys (trk_vcf_skip_samples), the batch's text goes to the device once, the genotype
        tensor and the selected scalar planes come into being there (``RawBatch.dev``).  A batch in which the device
        flags any record is parsed by the host after all (trk_vcf_parse_samples): same arrays either way.  Needs
        scalar Integer / Float planes (at most four) and no sample map; ``None`` turns it off."""
        ok = engine is not None and not getattr(self, '_map_out', 0) and len(self._selected) <= _lib.PARSE_MAX_PLANES and \
            all(nc == 1 and kd in (KIND_INT, KIND_FLOAT) for _, kd, nc, _ in self._selected)
        self._dev_eng = engine if ok else None
        self._lib.trk_vcf_skip_samples(self._h, 1 if ok else 0)
        return bool(ok)

    def device_inflate(self, engine):
        """The file's BGZF members inflated ON THE DEVICE (round 5; trk_inflate_blocks / trk_inflate_hook, include/trk.h):
        the compressed bytes cross PCIe, the text comes into being in HBM where trk_parse_samples reads it, and the host
        gets the newlines and the heads of the lines (CHROM ... FORMAT) -- all it reads of a batch.  Needs ``device_parse``
        on, a bgzip'ed file read from start to end (no region, no shard).  Returns whether it is on."""
        if engine is None or getattr(self, '_dev_eng', None) is not engine or getattr(self, '_inflate_on', False):
            return bool(getattr(self, '_inflate_on', False))
        self._drop_pending()
        user, seed, infl = C.c_void_p(), C.c_void_p(), C.c_void_p()
        if engine.lib.trk_inflate_hook(engine.ctx, C.byref(user), C.byref(seed), C.byref(infl)) != 0:
            return False
        # (runs of 16 x the CUs members: a member is one wave's serial work of ~5 ms whatever else runs; the hook keeps
        # twelve of them on a CU at a time -- three workgroups, room for the other kernels' LDS -- and the reader never waits
        # for a run's kernel anyway; runs of 12 x the CUs measured the same: tools/inflate_probe.py, r05 notes section 6)
        hook = _InflateHook(user.value, seed.value, infl.value,
                            int(_knobs.lab('TRK_INFLATE_RUN_PER_CU', '16')) * int(getattr(engine, 'n_cu', 256)))
        if _knobs.lab('TRK_INFLATE_PIPELINE', '1') == '1':
            # two runs in flight: run k + 1 is read and uploaded behind the kernel of run k (trk_inflate_hook_async)
            sub, col = C.c_void_p(), C.c_void_p()
            if engine.lib.trk_inflate_hook_async(engine.ctx, C.byref(sub), C.byref(col)) == 0:
                hook.submit, hook.collect = sub.value, col.value
        if self._lib.trk_vcf_set_inflate_hook(self._h, C.byref(hook)) != 0:
            return False                     # (plain gzip / text, a shard, a region: the host inflates)
        self._inflate_hook = hook
        self._inflate_on = True
        return True

    def _parse_on_device(self, b, P, arrays):
        """The sample columns of the batch just read, on the device; returns (dev dict or None, the host arrays)."""
        gt, ph, lp, planes, gtm, parr = arrays
        eng, n, S = self._dev_eng, b.n_records, self.n_samples
        lo = np.ctypeslib.as_array(b.line_off, shape=(n,))
        le = np.ctypeslib.as_array(b.line_end, shape=(n,))
        fo9 = np.ctypeslib.as_array(b.field_off, shape=(n * 10,))[9::10]
        base = int(lo[0]) & ~15                         # the upload starts on a 16-byte boundary of the reader's buffer
        nbytes = int(le[-1]) + 1 - base
        self._text_seen = max(getattr(self, '_text_seen', 0), int(le[-1]) + 1)
        stride = C.c_int32()
        fptr = self._lib.trk_vcf_format_idx(self._h, C.byref(stride))
        fi = np.ctypeslib.as_array(C.cast(fptr, C.POINTER(C.c_int8)), shape=(n, stride.value)) if fptr else None
        out = None
        sparse = bool(getattr(self, '_inflate_on', False))
        if sparse and not (fi is not None and nbytes > 0 and S > 0):
            raise ValueError("%s: a batch without sample columns in device-inflate mode" % self.path)
        if fi is not None and nbytes > 0 and S > 0:
            td = eng.empty((nbytes + 32,), np.uint8)
            if sparse:
                # the text is in HBM already (the inflate hook's segments): the batch's span of it, device to device
                abs0 = int(self._lib.trk_vcf_text_abs(self._h))
                eng._chk(eng.lib.trk_inflate_text(eng.ctx, abs0 + base, nbytes, td.ptr, abs0 + base))
            else:
                eng._chk(eng.lib.trk_memcpy_h2d(eng.ctx, td.ptr, b.text + base, nbytes))
            kinds = ['f' if kd == KIND_FLOAT else 'i' for _, kd, _, _ in self._selected]
            so_d = eng.upload((lo + fo9 - base).astype(np.int64), np.int64)
            le_d = eng.upload((le - base).astype(np.int64), np.int64)
            out = eng.parse_samples(td, so_d, le_d, S, P,
                                    np.ascontiguousarray(fi[:, 0]),
                                    planes=[(np.ascontiguousarray(fi[:, 1 + j]), k) for j, k in enumerate(kinds)],
                                    want_phased=True)
            flags = out['flags'].get()
            if flags.any():
                if sparse:                        # the host parses this batch: its sample columns have to be there
                    eng._chk(eng.lib.trk_memcpy_d2h(eng.ctx, b.text + base, td.ptr, nbytes))
                with eng.idle_frees():            # (parse_samples waited for the device; .get() for this thread's queue)
                    for a in [out['gt'], out['phased'], out['locus_ploidy'], out['flags'], td, so_d, le_d] + out['planes']:
                        a.free()
                out = None
        if out is None:
            # something the device grammar does not cover (or an error the host reports in its own words): the host parses
            # the batch after all -- with a wider tensor when a call holds more alleles than this one has columns
            self.device_fallbacks = getattr(self, 'device_fallbacks', 0) + 1
            while True:
                rc = self._lib.trk_vcf_parse_samples(self._h, C.byref(b))
                if rc == 5 and b'haplotypes' in self._lib.trk_vcf_last_error(self._h) and P < 8:
                    P = min(8, 2 * P)
                    gt, ph, lp, planes, gtm = self._arrays(max(n, 1), S, P, ring=False)
                    parr = (C.c_void_p * max(len(planes), 1))(*[p.ctypes.data for p in planes])
                    b.gt, b.phased, b.locus_ploidy = gt.ctypes.data, ph.ctypes.data, lp.ctypes.data
                    b.gt_mapped = None if gtm is None else gtm.ctypes.data
                    b.planes = C.cast(parr, C.POINTER(C.c_void_p))
                    b.max_ploidy = P
                    continue
                if rc != 0:
                    raise ValueError(self._lib.trk_vcf_last_error(self._h).decode())
                break
            return None, gt, ph, lp, planes, gtm, parr
        lp[:n] = out['locus_ploidy'].get()
        with eng.idle_frees():
            out['locus_ploidy'].free()
            out['flags'].free()
        dev = dict(gt=out['gt'], phased=out['phased'], planes={k: a for (k, _, _, _), a in zip(self._selected, out['planes'])},
                   text=td, smp_off=so_d, line_end=le_d, eng=eng,     # (text and offsets stay: the record writer's device half)
                   sparse_text=sparse, text_base=base, text_nbytes=nbytes)
        return dev, gt, ph, lp, planes, gtm, parr

    def shard(self, rank, world):
        """Keep only this rank's contiguous share of the records (trk_vcf_shard: the compressed file cut at BGZF block
        boundaries, a record belongs to the rank in whose range its line starts).  Call before the first read.
        True when the file could be cut; False leaves the reader untouched (plain gzip: every rank reads everything
        and the callers deal batches out round-robin)."""
        if world <= 1:
            return True
        self._drop_pending()
        b, e = C.c_uint64(), C.c_uint64()
        if self._lib.trk_vcf_shard(self._h, int(rank), int(world), C.byref(b), C.byref(e)) != 0:
            return False
        self.shard_range = (b.value, e.value)
        self._rows, self._row_i, self._eof = [], 0, False
        return True

    def counters(self):
        """{inflated bytes, compressed bytes, BGZF blocks} read since ``shard`` (trk_vcf_counters)."""
        out = (C.c_uint64 * 3)()
        self._lib.trk_vcf_counters(self._h, out)
        return dict(inflated=int(out[0]), compressed=int(out[1]), blocks=int(out[2]))

    def use_buffers(self, allocator=None, ring=2, release=None):
        """Decode batches into a ring of ``ring`` preallocated array sets instead of fresh numpy arrays (a batch then
        stays valid until ``ring`` further batches have been read).  Fresh arrays cost a page fault per 4 KB inside
        the parser threads, which serialise in the kernel -- the reader ran at a fifth of its rate because of it.
        ``allocator(nbytes) -> uint8 array``: where the memory comes from (DeviceCompute.host_buffer: pinned pages,
        so that the upload is a plain DMA)."""
        self._ring, self._ring_i, self._ring_n, self._alloc = [], 0, int(ring), allocator
        self._release, self._slabs = release, []     # ``release(slab)``: called for every slab when the reader closes

    def set_sample_map(self, col_of_sample, n_out):
        """trk_vcf_set_sample_map: every batch also carries ``gt_mapped`` [n, n_out, P] -- sample s of the file in
        column ``col_of_sample[s]`` (-1: left out), unmapped columns no-calls.  ``None``: off."""
        if col_of_sample is None:
            self._lib.trk_vcf_set_sample_map(self._h, None, 0)
            self._map_out = 0
            return self
        m = np.ascontiguousarray(col_of_sample, dtype=np.int32)
        if m.shape != (self.n_samples,):
            raise ValueError("one column per sample of the file")
        if self._lib.trk_vcf_set_sample_map(self._h, m.ctypes.data, int(n_out)) != 0:
            raise ValueError(self._lib.trk_vcf_last_error(self._h).decode())
        self._map_out = int(n_out)
        return self

    def _arrays(self, n, S, P, ring=True):
        M = getattr(self, '_map_out', 0)
        if getattr(self, '_ring_n', 0) <= 0 or not ring:
            return (np.empty((n, S, P), dtype=np.int16), np.empty((n, S), dtype=np.uint8), np.empty(n, dtype=np.uint8),
                    [np.empty((n, S, nc), dtype=dt) for _, _, nc, dt in self._selected],
                    np.empty((n, M, P), dtype=np.int16) if M else None)
        key = (n, S, P, M, tuple((nc, np.dtype(dt).str) for _, _, nc, dt in self._selected))
        if len(self._ring) < self._ring_n or self._ring[self._ring_i % self._ring_n][0] != key:
            # one slab per ring slot (a pinned allocation costs milliseconds whatever its size), carved 64-byte aligned
            shapes = [((n, S, P), np.int16), ((n, S), np.uint8), ((n,), np.uint8)] + \
                     [((n, S, nc), dt) for _, _, nc, dt in self._selected] + ([((n, M, P), np.int16)] if M else [])
            sizes = [(int(np.prod(sh, dtype=np.int64)) * np.dtype(dt).itemsize + 63) & ~63 for sh, dt in shapes]
            total = max(sum(sizes), 64)
            if getattr(self, '_dev_eng', None) is not None:
                # device parse: these arrays are filled only if somebody asks for the host copies (RawBatch._host) or the
                # host has to parse a batch the device flags -- address space, no pinned pages, nothing touched (two
                # pinned slabs of ~220 MB were 40-80 ms of a fresh process's first batches for nothing)
                slab = np.empty(total, dtype=np.uint8)
            elif self._alloc is not None:
                slab = self._alloc(total)
                self._slabs.append(slab)
            else:
                slab = np.empty(total, dtype=np.uint8)
                slab[:] = 0                                      # touch the pages now, not in the parser threads
            cursor = [0]

            def take(shape, dt):
                nb = int(np.prod(shape, dtype=np.int64)) * np.dtype(dt).itemsize
                a = slab[cursor[0]:cursor[0] + nb].view(dt).reshape(shape)
                cursor[0] += (nb + 63) & ~63
                return a
            slot = (key, (take((n, S, P), np.int16), take((n, S), np.uint8), take((n,), np.uint8),
                          [take((n, S, nc), dt) for _, _, nc, dt in self._selected],
                          take((n, M, P), np.int16) if M else None))
            if len(self._ring) < self._ring_n:
                self._ring.append(slot)
                self._ring_i = len(self._ring) - 1
            else:
                self._ring[self._ring_i % self._ring_n] = slot
        arrs = self._ring[self._ring_i % self._ring_n][1]
        self._ring_i += 1
        return arrs

    def _rows_of(self, rb):
        b, S = rb.b, self.n_samples
        rb.fetch_text()
        rows = []
        for i in range(rb.n):
            # the nine fixed columns as text now; the sample columns (most of the line) stay bytes until a field
            # that was not decoded natively is asked for
            f9 = int(b.field_off[i * 10 + 9])
            n_line = b.line_end[i] - b.line_off[i]
            if 0 < f9 < n_line:
                line = C.string_at(b.text + b.line_off[i], f9 - 1).decode()
                tail = C.string_at(b.text + b.line_off[i] + f9, n_line - f9)
            else:
                line, tail = C.string_at(b.text + b.line_off[i], n_line).decode(), None
            pl = int(rb.locus_ploidy[i])
            g = np.empty((S, pl + 1), dtype=np.int16)
            g[:, :pl] = rb.gt[i, :, :pl]
            g[:, pl] = rb.phased[i]
            native = {k: p[i] for k, p in rb.planes.items()}
            rows.append((line, g, native, tail))
        return rows

    def _next_batch(self):
        rb = self.read_raw_batch()
        self._rows, self._row_i = self._rows_of(rb), 0
        self._eof = rb.n == 0
        self.last_batch = dict(gt=rb.gt, phased=rb.phased, locus_ploidy=rb.locus_ploidy, planes=rb.planes)

    def __next__(self):
        while True:
            if self._row_i >= len(self._rows):
                if self._eof:
                    raise StopIteration
                self._next_batch()
                if self._eof:
                    raise StopIteration
            line, g, native, tail = self._rows[self._row_i]
            self._row_i += 1
            v = vcfio.Variant(self, line, gt=g if self.n_samples else None, native=native, tail=tail)
            if v.CHROM not in self.contigs_declared and v.CHROM not in self.contigs_seen:
                self.contigs_seen.append(v.CHROM)
            if self._region is not None and not self._in_region(v):
                if self._indexed_region and self._past_region(v):
                    self._eof, self._rows = True, []     # sorted file: nothing further can overlap
                    raise StopIteration
                continue
            return v

    def __call__(self, region):
        """Region query ``chrom[:start-end]``.  With a tabix index next to a bgzipped file the reader
        seeks to the first 16 kb window the region touches (the index's linear table) and stops at
        the first record past the region; without one it scans the whole file."""
        vcfio.VCFReader.__call__(self, region)
        self._drop_pending()
        self._indexed_region = False
        tbi = self.path + '.tbi'
        if os.path.isfile(tbi) and _knobs.lab('TRK_TABIX', '1') != '0':
            from . import tabix
            try:
                idx = tabix.TabixIndex.load(tbi)
            except (OSError, ValueError, struct_error):
                return self
            chrom, start, _ = self._region
            off = idx.start_offset(chrom, start)
            if off is None or off < 0:
                self._eof, self._rows, self._row_i = True, [], 0   # sequence absent / region past its end
                return self
            if self._lib.trk_vcf_seek(self._h, off) == 0:
                self._rows, self._row_i, self._eof = [], 0, False
                self._indexed_region = True
        return self

    def region_keep(self, rb, hz):
        """The region query of ``__call__`` applied to a raw batch: (keep bool[n], done).  A record is in the region
        when its CHROM matches and [POS, POS + len(REF) - 1] meets [start, end] (``_in_region``); with an index the
        first record past the region ends the query (``done``; the records after it are dropped too)."""
        n = rb.n
        if self._region is None or n == 0:
            return np.ones(n, dtype=bool), False
        chrom, start, end = self._region
        same = np.array([c == chrom for c in rb.chrom_column()], dtype=bool)
        pos = hz.pos
        keep = same.copy()
        if start is not None:
            fo = np.ctypeslib.as_array(rb.b.field_off, shape=(n * 10,)).reshape(n, 10)
            ref_len = (fo[:, 4] - fo[:, 3] - 1).astype(np.int64)
            keep &= (pos + ref_len - 1 >= start) & (pos <= end)
        done = False
        if self._indexed_region:
            past = ~same
            if end is not None:
                past |= pos > end
            past &= ~keep
            if past.any():
                keep[int(np.flatnonzero(past)[0]):] = False
                done = True
        return keep, done

    def _past_region(self, v):
        chrom, start, end = self._region
        if v.CHROM != chrom:
            return True
        return end is not None and v.POS > end

    def close(self):
        if self._h is not None:
            self._drop_pending()
            self._lib.trk_vcf_close(self._h)
            self._h = None
        slabs, self._slabs, self._ring = getattr(self, '_slabs', []), [], []
        rel = getattr(self, '_release', None)
        if rel is not None:
            for slab in slabs:
                rel(slab)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
