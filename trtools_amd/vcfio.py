"""
Host-side VCF text reader / writer.

The reference delegates all VCF decoding to cyvcf2/htslib
(trtools/utils/utils.py:19-67 ``LoadSingleReader`` -> ``cyvcf2.VCF``); what
reaches the hot path is a handful of ``cyvcf2.Variant`` attributes
(SURVEY.md section 1, section 8b).  This module decodes VCF 4.x text (plain,
gzip or bgzip) into the same arrays with the same conventions, so that the
batch packer and the TRRecord facade see what they would see from cyvcf2:

* ``Variant.genotype.array()``  int16 ``[S, maxploidy+1]``; ``-1`` missing
  haplotype, ``-2`` ploidy padding, last column = phased bit
  (consumed at tr_harmonizer.py:860-862);
* ``Variant.format(key)``  Integer -> int32 ``[S,k]`` (missing ``INT_MIN``,
  short vectors padded with ``INT_MIN+1``), Float -> float32 (missing nan),
  String -> ``<U`` array ``[S]``; ``KeyError`` when the key is not in FORMAT
  (consumed through tr_harmonizer.py:561-588);
* ``Variant.INFO`` typed per the header (Integer -> int, Float -> float32
  rounded float, String -> str, multi-valued -> tuple, Flag -> True).

It is host plumbing (a native block-parallel parser is SURVEY.md section 8f
row 1), not part of the device hot path.
"""
from . import _knobs
import ctypes
import gzip
import os
import re

import numpy as np

INT_MISSING = -2147483648
INT_VECTOR_END = -2147483647

_HDR_RE = re.compile(r'^##(\w+)=<(.*)>\s*$')


def _split_header_fields(body):
    """Split ``ID=x,Number=1,Description="a, b"`` respecting quotes."""
    out = {}
    key, val, inq, cur = None, [], False, []
    i = 0
    n = len(body)
    while i < n:
        c = body[i]
        if key is None:
            if c == '=':
                key = ''.join(cur).strip()
                cur = []
            else:
                cur.append(c)
        else:
            if c == '"' and (i == 0 or body[i - 1] != '\\'):
                inq = not inq
            elif c == ',' and not inq:
                out[key] = ''.join(cur)
                key, cur = None, []
            else:
                cur.append(c)
        i += 1
    if key is not None:
        out[key] = ''.join(cur)
    return out


class _Genotype:
    """Stand-in for ``cyvcf2.Variant.genotype`` (array(), n_samples)."""

    def __init__(self, arr):
        self._arr = arr
        self.n_samples = arr.shape[0]

    def array(self):
        return self._arr


class _Info:
    """dict-like INFO with cyvcf2's ``get`` / ``[]`` / ``[]=`` / iteration."""

    def __init__(self, pairs):
        self._d = dict(pairs)
        self._order = [k for k, _ in pairs]

    def get(self, key, default=None):
        return self._d.get(key, default)

    def __getitem__(self, key):
        return self._d[key]

    def __setitem__(self, key, value):
        if key not in self._d:
            self._order.append(key)
        self._d[key] = value

    def __contains__(self, key):
        return key in self._d

    def __iter__(self):
        return iter([(k, self._d[k]) for k in self._order])

    def keys(self):
        return list(self._order)




class _CallFilterStruct(ctypes.Structure):
    _fields_ = [('mask', ctypes.c_void_p), ('n_filters', ctypes.c_int32), ('reserved', ctypes.c_int32),
                ('names', ctypes.POINTER(ctypes.c_char_p)), ('values', ctypes.POINTER(ctypes.c_void_p))]


def _addr(arr):
    """Address of a numpy array's buffer (cheaper than ``arr.ctypes.data``)."""
    return arr.__array_interface__['data'][0]


class CallFilterColumn:
    """dumpSTR's FORMAT/FILTER column of one record, kept as the call-filter mask until the record is written
    (dumpSTR.py:648-683): ``NOCALL`` (bit 31), ``PASS`` (0) or ``<name>_<%g value>`` of every fired filter, comma
    joined.  ``to_array()`` is the text as a numpy string array; the native serialiser writes it from the mask."""

    def __init__(self, mask, names, values):
        self.mask = np.ascontiguousarray(mask, dtype=np.uint32)
        self.names = list(names)
        self.values = [None if v is None else np.ascontiguousarray(v, dtype=np.float64) for v in values]

    def to_list(self):
        m = self.mask
        nocall = (m & np.uint32(0x80000000)) != 0
        out = np.full(len(m), 'PASS', dtype=object)
        out[nocall] = 'NOCALL'
        nk = len(self.names)
        for s in np.nonzero(~nocall & (m != 0))[0]:
            ms = int(m[s])
            out[s] = ','.join('%s_%s' % (self.names[k], '%g' % (np.nan if self.values[k] is None else self.values[k][s]))
                              for k in range(nk) if (ms >> k) & 1)
        return out.tolist()

    def to_array(self):
        return np.array(self.to_list()) if len(self.mask) else np.array([], dtype='<U1')

    def max_text(self):
        return max(8, sum(len(n.encode()) + 26 for n in self.names))

    def native_struct(self):
        nk = len(self.names)
        CallFilter = _CallFilterStruct
        names = (ctypes.c_char_p * max(nk, 1))(*[n.encode() for n in self.names])
        vals = (ctypes.c_void_p * max(nk, 1))(*[None if v is None else _addr(v) for v in self.values])
        st = CallFilter(_addr(self.mask), nk, 0, names, vals)
        return [st, names, vals, self]      # element 0 is the struct; the rest keeps its pointers alive

class _Decode(ctypes.Structure):
    _fields_ = [('kind', ctypes.c_int32), ('ncol', ctypes.c_int32), ('out', ctypes.c_void_p)]


_SERIALIZER = False


def _serializer():
    """(lib, Column struct) of the native record serialiser, or None when libtrk.so is not built -- text
    formatting is host IO, not the compute path, and falls back to the Python loop."""
    global _SERIALIZER
    if _SERIALIZER is False:
        _SERIALIZER = None
        if _knobs.lab('TRK_NATIVE_WRITER', '1') != '0':
            try:
                from . import _lib
                lib = _lib.load()

                class Column(ctypes.Structure):
                    _fields_ = [('kind', ctypes.c_int32), ('ncol', ctypes.c_int32), ('itemsize', ctypes.c_int32),
                                ('reserved', ctypes.c_int32), ('data', ctypes.c_void_p)]
                lib.trk_vcf_format_samples.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(Column),
                                                       ctypes.c_char_p, ctypes.c_int64]
                lib.trk_vcf_format_samples.restype = ctypes.c_int64
                lib.trk_vcf_decode_formats.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                                       ctypes.POINTER(_Decode), ctypes.c_int32]
                _SERIALIZER = (lib, Column)
            except (OSError, AttributeError, RuntimeError):
                _SERIALIZER = None
    return _SERIALIZER

class Variant:
    """One VCF record with the cyvcf2.Variant surface the hot path touches."""

    __slots__ = ('_reader', '_fields', 'CHROM', 'POS', 'ID', 'REF', 'ALT', 'QUAL',
                 '_filter', 'INFO', 'FORMAT', '_tail', '_samples_list', '_cols', '_fmt_cache',
                 '_gt', '_gtlist', 'ploidy', '_set_formats', '_info_dirty')

    def __init__(self, reader, line, gt=None, native=None, tail=None):
        """``gt`` (int16 [S, P+1], cyvcf2 layout) and ``native`` ({FORMAT key: array [S, k]}) are
        supplied by the native reader (trtools_amd.vcfnative); the sample columns are then only
        split in Python if a field that was not decoded natively is asked for.  With ``tail``
        (bytes of the sample columns) ``line`` holds the nine fixed columns only and the sample text
        is decoded on first use."""
        f = line.rstrip('\r\n').split('\t', 9)
        if tail is not None:
            f = f[:9] + [tail]
        self._reader = reader
        self._fields = f
        self.CHROM = f[0]
        self.POS = int(f[1])
        self.ID = None if f[2] == '.' else f[2]
        self.REF = f[3]
        self.ALT = [] if f[4] == '.' else f[4].split(',')
        self.QUAL = None if f[5] == '.' else float(f[5])
        self._filter = None if f[6] in ('.', 'PASS') else f[6]
        self.INFO = _Info(reader._parse_info(f[7]))
        if len(f) > 8:
            self.FORMAT = f[8].split(':')
            self._tail = f[9] if len(f) > 9 else ''
        else:
            self.FORMAT = []
            self._tail = ''
        self._samples_list = None
        self._cols = None
        self._fmt_cache = dict(native) if native else {}
        self._set_formats = {}
        self._gt = None
        self._gtlist = None
        self.ploidy = 2
        if gt is not None:
            self._gt = gt
            self.ploidy = gt.shape[1] - 1
        elif reader.n_samples and 'GT' in self.FORMAT:
            self._decode_gt()

    @property
    def _samples(self):
        if self._samples_list is None:
            if isinstance(self._tail, bytes):
                self._tail = self._tail.decode().rstrip('\r\n')
            self._samples_list = self._tail.split('\t') if self._tail else []
        return self._samples_list

    # ---- FILTER (cyvcf2: None when PASS or '.') ----
    @property
    def FILTER(self):
        return self._filter

    @FILTER.setter
    def FILTER(self, value):
        self._filter = value

    # ---- per-sample columns ----
    def _columns(self):
        if self._cols is None:
            nf = len(self.FORMAT)
            cols = [[] for _ in range(nf)]
            for s in self._samples:
                parts = s.split(':')
                np_ = len(parts)
                for i in range(nf):
                    cols[i].append(parts[i] if i < np_ else '.')
            self._cols = cols
        return self._cols

    def _decode_gt(self):
        gi = self.FORMAT.index('GT')
        if gi == 0:
            raw = [s.split(':', 1)[0] for s in self._samples]
        else:
            raw = self._columns()[gi]
        split = []
        maxp = 1
        for g in raw:
            phased = '|' in g
            al = g.replace('|', '/').split('/')
            if len(al) > maxp:
                maxp = len(al)
            split.append((al, phased))
        arr = np.full((len(raw), maxp + 1), -2, dtype=np.int16)
        for i, (al, phased) in enumerate(split):
            for j, a in enumerate(al):
                arr[i, j] = -1 if a == '.' else int(a)
            arr[i, maxp] = 1 if phased else 0
        self._gt = arr
        self.ploidy = maxp

    @property
    def genotype(self):
        if self._gt is None:
            return None
        return _Genotype(self._gt)

    @property
    def genotypes(self):
        """cyvcf2 ``Variant.genotypes``: list of [a0, a1, ..., phased].  Like cyvcf2 the list
        is cached, so ``v.genotypes[i] = ...; v.genotypes = v.genotypes`` updates the record."""
        if self._gt is None:
            return []
        if self._gtlist is None:
            p = self._gt.shape[1] - 1
            self._gtlist = [[int(x) for x in row[:p]] + [bool(row[p])] for row in self._gt]
        return self._gtlist

    @genotypes.setter
    def genotypes(self, value):
        p = max(len(v) - 1 for v in value)
        arr = np.full((len(value), p + 1), -2, dtype=np.int16)
        for i, v in enumerate(value):
            for j, a in enumerate(v[:-1]):
                arr[i, j] = a
            arr[i, p] = 1 if v[-1] else 0
        self._gt = arr
        self._gtlist = None
        self.ploidy = p

    def set_gt_array(self, arr):
        """Replace the genotype matrix (int16 [S, P+1], cyvcf2 layout)."""
        self._gt = np.asarray(arr, dtype=np.int16)
        self._gtlist = None
        self.ploidy = self._gt.shape[1] - 1

    def format(self, key):
        if key in self._set_formats:
            v = self._set_formats[key]
            if isinstance(v, CallFilterColumn):
                v = self._set_formats[key] = v.to_array()
            return v
        if key.startswith('__') and key in self._fmt_cache:   # pre-parsed plane of the native reader
            return self._fmt_cache[key]
        if key not in self.FORMAT:
            raise KeyError(key)
        if key in self._fmt_cache:
            return self._fmt_cache[key]
        if key != 'GT' and self._cols is None and self._decode_formats_native() and key in self._fmt_cache:
            return self._fmt_cache[key]
        col = self._columns()[self.FORMAT.index(key)]
        typ = self._reader.format_types.get(key, ('String', '1'))[0]
        if key == 'GT':
            out = np.array(col)
        elif typ == 'Integer':
            out = self._numeric(col, np.int32, INT_MISSING, INT_VECTOR_END, int)
        elif typ == 'Float':
            out = self._numeric(col, np.float32, np.nan, np.nan, float)
        else:
            out = np.array(col) if len(col) else np.array([], dtype='<U1')
        self._fmt_cache[key] = out
        return out

    def _decode_formats_native(self):
        """Decode every not yet cached Integer / Float / String FORMAT field of the record in one native pass over
        the sample columns (trk_vcf_decode_formats, include/trk_vcf.h) -- the arrays ``_numeric`` / ``np.array(col)``
        below build.  False when libtrk.so is not built or the text does not parse (the Python decoder then raises
        the error the reference's reader would)."""
        api = _serializer()
        if api is None:
            return False
        lib = api[0]
        tail = self._tail
        if self._samples_list is not None:
            tail = '\t'.join(self._samples_list)
        raw = tail if isinstance(tail, bytes) else tail.encode()
        n = self._reader.n_samples if self._gt is None else int(self._gt.shape[0])
        nf = len(self.FORMAT)
        if n == 0 or nf == 0:
            return False
        fields = (_Decode * nf)()
        want = []
        for i, key in enumerate(self.FORMAT):
            kind = -1
            if key != 'GT' and key not in self._fmt_cache:
                typ = self._reader.format_types.get(key, ('String', '1'))[0]
                kind = 1 if typ == 'Integer' else 2 if typ == 'Float' else 4
                want.append(i)
            fields[i].kind = kind
        if not want:
            return False
        if lib.trk_vcf_decode_formats(raw, len(raw), n, nf, fields, 0) != 0:
            return False
        outs = {}
        for i in want:
            k = fields[i].ncol
            if fields[i].kind == 4:
                arr = np.zeros(n, dtype='<U%d' % k)
            else:
                arr = np.empty((n, k), dtype=np.int32 if fields[i].kind == 1 else np.float32)
            outs[i] = arr
            fields[i].out = _addr(arr)
        if lib.trk_vcf_decode_formats(raw, len(raw), n, nf, fields, 1) != 0:
            return False
        for i in want:
            self._fmt_cache[self.FORMAT[i]] = outs[i]
        return True

    @staticmethod
    def _numeric(col, dtype, missing, pad, conv):
        n = len(col)
        simple = True
        for v in col:
            if ',' in v:
                simple = False
                break
        if simple:
            out = np.empty((n, 1), dtype=dtype)
            for i, v in enumerate(col):
                out[i, 0] = missing if v == '.' else conv(v)
            return out
        parts = [v.split(',') for v in col]
        k = max(len(p) for p in parts)
        out = np.full((n, k), pad, dtype=dtype)
        for i, p in enumerate(parts):
            for j, v in enumerate(p):
                out[i, j] = missing if v == '.' else conv(v)
        return out

    def set_format(self, key, arr):
        if key not in self.FORMAT:
            self.FORMAT = list(self.FORMAT) + [key]
        self._set_formats[key] = arr

    # ---- text ----
    def _format_value(self, key, i):
        if key == 'GT':
            row = self._gt[i]
            p = len(row) - 1
            sep = '|' if row[p] else '/'
            toks = ['.' if a == -1 else str(int(a)) for a in row[:p] if a != -2]
            return sep.join(toks) if toks else '.'
        arr = self.format(key)
        v = arr[i]
        if arr.dtype.kind in 'US':
            s = v.decode() if isinstance(v, bytes) else str(v)
            return s if s != '' else '.'
        toks = []
        if arr.dtype.kind == 'f' and np.all(np.isnan(np.atleast_1d(v))):
            return '.'          # a fully missing vector is a single '.' (htslib)
        for x in np.atleast_1d(v):
            if arr.dtype.kind == 'i':
                if x == INT_VECTOR_END:
                    break
                toks.append('.' if x == INT_MISSING else str(int(x)))
            else:
                toks.append('.' if np.isnan(x) else _fmt_float(float(x)))
        return ','.join(toks) if toks else '.'

    def _info_text(self):
        return info_text(self.INFO)

    def _info_text_old(self):
        toks = []
        for k, v in self.INFO:
            if v is True:
                toks.append(k)
            elif isinstance(v, (tuple, list)):
                toks.append(k + '=' + ','.join(_fmt_info(x) for x in v))
            else:
                toks.append(k + '=' + _fmt_info(v))
        return ';'.join(toks) if toks else '.'

    def __str__(self):
        return self.to_text()

    def to_text(self, native=True):
        """The record line.  native=False: format the sample columns with the Python loop (the definition the
        native serialiser is tested against)."""
        f = self._fields
        filt = 'PASS' if self._filter is None and f[6] != '.' else (self._filter or f[6])
        if self._filter is not None:
            filt = self._filter
        head = [self.CHROM, str(self.POS), self.ID or '.', self.REF,
                ','.join(self.ALT) if self.ALT else '.', f[5], filt, self._info_text()]
        if not self.FORMAT:
            return '\t'.join(head) + '\n'
        head.append(':'.join(self.FORMAT))
        body = self._samples_text_native() if native else None
        if body is not None:
            return '\t'.join(head) + body + '\n'
        n = len(self._samples)
        for i in range(n):
            head.append(':'.join(self._format_value(k, i) for k in self.FORMAT))
        return '\t'.join(head) + '\n'

    def _samples_text_native(self):
        """The per-sample columns ('\\t' + fields joined by ':' for every sample) from the typed FORMAT arrays,
        serialised by libtrk (trk_vcf_format_samples, include/trk_vcf.h) -- the same text the loop over
        ``_format_value`` builds.  None when a column is not a plain int32 / float32 / fixed-width string array
        (object arrays, other integer widths): the caller then formats in Python."""
        api = _serializer()
        if api is None or self._gt is None and 'GT' in self.FORMAT:
            return None
        lib, Column = api
        n = self.n_samples_hint()
        if n == 0:
            return None
        cols = (Column * len(self.FORMAT))()
        keep = []
        cap = 0
        for i, key in enumerate(self.FORMAT):
            lazy = self._set_formats.get(key)
            if key == 'GT':
                arr = np.ascontiguousarray(self._gt, dtype=np.int16)
                kind, ncol, item = 0, arr.shape[1], 0
                cap += n * (7 * ncol + 1)
            elif isinstance(lazy, CallFilterColumn):
                if len(lazy.mask) != n:
                    return None
                arr = lazy.native_struct()
                keep.append(arr)
                cols[i] = Column(5, 1, 0, 0, ctypes.addressof(arr[0]))
                cap += n * lazy.max_text()
                continue
            else:
                arr = self.format(key)
                if not isinstance(arr, np.ndarray) or arr.shape[:1] != (n,) or arr.ndim > 2:
                    return None
                dk = arr.dtype.kind
                if dk in 'if':
                    if arr.dtype != (np.int32 if dk == 'i' else np.float32):
                        return None
                    arr = np.ascontiguousarray(arr.reshape(n, -1))
                    if arr.shape[1] < 1:
                        return None
                    kind, ncol, item = (1 if dk == 'i' else 2), arr.shape[1], 0
                    cap += n * (17 * ncol + 1)
                elif dk in 'US' and arr.ndim == 1 and arr.dtype.itemsize > 0 and arr.dtype.isnative:
                    arr = np.ascontiguousarray(arr)
                    kind, ncol, item = (4 if dk == 'U' else 3), 1, arr.dtype.itemsize
                    cap += n * (item + 2)
                else:
                    return None
            keep.append(arr)
            cols[i] = Column(kind, ncol, item, 0, _addr(arr))
        cap += n + 16
        buf = ctypes.create_string_buffer(cap)
        got = lib.trk_vcf_format_samples(n, len(self.FORMAT), cols, buf, cap)
        if got < 0:
            return None
        return buf.raw[:got].decode()

    def n_samples_hint(self):
        if self._gt is not None:
            return int(self._gt.shape[0])
        return len(self._samples)


def _fmt_float(x):
    s = '%g' % x
    return s


def info_text(info):
    """The INFO column from an ``_Info`` (typed values re-serialised the way htslib writes them)."""
    toks = []
    for k, v in info:
        if v is True:
            toks.append(k)
        elif isinstance(v, (tuple, list)):
            toks.append(k + '=' + ','.join(_fmt_info(x) for x in v))
        else:
            toks.append(k + '=' + _fmt_info(v))
    return ';'.join(toks) if toks else '.'


def rewrite_info(reader, text, updates):
    """``info_text`` of the record's INFO column after ``info[k] = v`` for every (k, v) of ``updates`` -- what
    ``_Info(reader._parse_info(text))`` + the assignments + ``info_text`` give, without building the typed objects
    when the text already reads the way it would be written back: String values, bare flags and integers in canonical
    form pass through as they are.  Anything else (Float values, which htslib re-serialises from float32; '+5' or
    '007'; a key that occurs twice; a Flag with a value) takes the general path."""
    if text == '.' or text == '':
        return info_text(updates) if updates else '.'
    toks = text.split(';')
    types = reader.info_types
    upd = dict(updates)
    seen = set()
    out = []
    for tok in toks:
        if not tok:
            continue
        eq = tok.find('=')
        k = tok if eq < 0 else tok[:eq]
        if k in seen:
            return _rewrite_info_general(reader, text, updates)
        seen.add(k)
        if k in upd:
            out.append(k + '=' + _fmt_one(upd[k]))
            continue
        if eq < 0:
            out.append(tok)
            continue
        typ = types.get(k, ('String', '.'))[0]
        if typ == 'Integer':
            v = tok[eq + 1:]
            for x in (v.split(',') if ',' in v else (v,)):
                if not ((x.isdigit() and (x[0] != '0' or x == '0')) or
                        (x[:1] == '-' and x[1:].isdigit() and x[1:2] != '0') or x == '.'):
                    return _rewrite_info_general(reader, text, updates)
            out.append(tok)
        elif typ == 'Float' or typ == 'Flag':
            return _rewrite_info_general(reader, text, updates)
        else:
            out.append(tok)
    for k, v in updates:
        if k not in seen:
            seen.add(k)
            out.append(k + '=' + _fmt_one(upd[k]))
    return ';'.join(out) if out else '.'


def _fmt_one(v):
    if v is True:
        raise ValueError("flags are not set through rewrite_info")
    if isinstance(v, (tuple, list)):
        return ','.join(_fmt_info(x) for x in v)
    return _fmt_info(v)


def _rewrite_info_general(reader, text, updates):
    info = _Info(reader._parse_info(text))
    for k, v in updates:
        info[k] = v
    return info_text(info)


def _fmt_info(v):
    if v is None:
        return '.'
    if isinstance(v, float):
        return _fmt_float(v)
    return str(v)


class VCFReader:
    """Iterates ``Variant`` objects of a VCF file (plain / gzip / bgzip)."""

    def __init__(self, path, lazy=False, samples=None):
        if not os.path.exists(path) or os.path.isdir(path):
            raise OSError("no such VCF: %s" % path)
        self.path = path
        with open(path, 'rb') as fh:
            magic = fh.read(2)
        if magic == b'\x1f\x8b':
            self._fh = gzip.open(path, 'rt')
        else:
            self._fh = open(path, 'r')
        self._header_lines = []
        self.samples = []
        self.info_types = {}
        self.format_types = {}
        self.contigs_declared = set()
        self.contigs_seen = []      # CHROMs of parsed records that the header does not declare
        self.has_pass_filter = False
        self._pending = None
        saw_chrom = False
        for line in self._fh:
            if line.startswith('##'):
                self._header_lines.append(line.rstrip('\r\n'))
                self._register(line)
            elif line.startswith('#CHROM'):
                cols = line.rstrip('\r\n').split('\t')
                self.samples = cols[9:]
                self._chrom_line = line.rstrip('\r\n')
                saw_chrom = True
                break
            else:
                break
        if not saw_chrom:
            raise OSError("%s does not look like a VCF (no #CHROM line)" % path)
        self.n_samples = len(self.samples)
        self._region = None

    # -- header --
    def _register(self, line):
        m = _HDR_RE.match(line)
        if not m:
            return
        kind, body = m.group(1), m.group(2)
        d = _split_header_fields(body)
        if kind == 'INFO':
            self.info_types[d.get('ID')] = (d.get('Type'), d.get('Number'))
        elif kind == 'FORMAT':
            self.format_types[d.get('ID')] = (d.get('Type'), d.get('Number'))
        elif kind == 'contig':
            self.contigs_declared.add(d.get('ID'))
        elif kind == 'FILTER' and d.get('ID') == 'PASS':
            self.has_pass_filter = True

    @property
    def raw_header(self):
        return '\n'.join(self._header_lines + [self._chrom_line]) + '\n'

    def header_iter(self):
        for line in self._header_lines:
            m = _HDR_RE.match(line)
            if not m:
                continue
            d = _split_header_fields(m.group(2))
            d['HeaderType'] = m.group(1) if m.group(1) in ('INFO', 'FORMAT', 'FILTER') \
                else m.group(1)
            if 'Description' in d:
                d['Description'] = d['Description']
            yield _HeaderRec(d)

    def add_to_header(self, line):
        self._header_lines.append(line.rstrip('\n'))
        self._register(line)

    def add_info_to_header(self, d):
        self.add_to_header('##INFO=<ID={ID},Number={Number},Type={Type},'
                           'Description="{Description}">'.format(**d))

    def add_format_to_header(self, d):
        self.add_to_header('##FORMAT=<ID={ID},Number={Number},Type={Type},'
                           'Description="{Description}">'.format(**d))

    def add_filter_to_header(self, d):
        self.add_to_header('##FILTER=<ID={ID},Description="{Description}">'.format(**d))

    # -- INFO parsing --
    def _parse_info(self, text):
        pairs = []
        if text == '.' or text == '':
            return pairs
        for tok in text.split(';'):
            if not tok:
                continue
            if '=' not in tok:
                pairs.append((tok, True))
                continue
            k, v = tok.split('=', 1)
            typ = self.info_types.get(k, ('String', '.'))[0]
            if typ == 'Integer':
                vals = [int(x) if x != '.' else None for x in v.split(',')]
                pairs.append((k, vals[0] if len(vals) == 1 else tuple(vals)))
            elif typ == 'Float':
                vals = [float(np.float32(x)) if x != '.' else None for x in v.split(',')]
                pairs.append((k, vals[0] if len(vals) == 1 else tuple(vals)))
            elif typ == 'Flag':
                pairs.append((k, True))
            else:
                pairs.append((k, v))
        return pairs

    # -- iteration --
    def __iter__(self):
        return self

    def __next__(self):
        while True:
            line = self._fh.readline()
            if not line:
                raise StopIteration
            if line.strip() == '':
                continue
            v = Variant(self, line)
            if v.CHROM not in self.contigs_declared and v.CHROM not in self.contigs_seen:
                self.contigs_seen.append(v.CHROM)   # htslib adds a dummy ##contig line for these
            if self._region is not None and not self._in_region(v):
                continue
            return v

    def __call__(self, region):
        """Region query ``chrom[:start-end]`` (linear scan; 1-based inclusive)."""
        chrom, start, end = region, None, None
        if ':' in region:
            chrom, rng = region.split(':', 1)
            a, b = rng.replace(',', '').split('-')
            start, end = int(a), int(b)
        self._region = (chrom, start, end)
        return self

    def _in_region(self, v):
        chrom, start, end = self._region
        if v.CHROM != chrom:
            return False
        if start is None:
            return True
        vend = v.POS + len(v.REF) - 1
        return vend >= start and v.POS <= end

    def close(self):
        self._fh.close()


class _HeaderRec(dict):
    """``header_iter()`` element: dict with ``['HeaderType']`` etc. (cyvcf2 HREC)."""

    def info(self):
        return dict(self)


class VCFWriter:
    """Plain-text VCF writer (``cyvcf2.Writer`` stand-in: write_record, close).

    Like htslib the header is emitted with the first record (or on close) and completed with
    what htslib adds on its own: ``##FILTER=<ID=PASS,...>`` after the fileformat line and a bare
    ``##contig=<ID=...>`` line for every contig the records used without declaring it."""

    def __init__(self, path, template):
        self.path = path
        self._tmpl = template
        self._wrote_header = False
        self._pool = None        # one background thread for write_bytes (TRK_ASYNC_WRITE=0: none)
        self._pending = None
        # --zip: the records' places in the text are noted as they are written -- (sequence, interval, first byte, byte
        # behind the newline) -- and close() turns them into the tabix index (write_index): the reference runs `tabix`
        # over the finished file (dumpSTR.py:1347-1352), rounds 1-5 read it back and inflated it once more in Python
        self._recs = None
        self.wrote_index = False
        if path.endswith('.gz'):
            from .bgzf import BgzfWriter
            # --zip output is real bgzip (dumpSTR.py:1241-1245).  TRK_ZIP_LEVEL: the level of the host compressor
            # (libdeflate; default 6, what `bgzip` gives); TRK_DEVICE_DEFLATE=1: the record writer's blocks are deflated on
            # the GPU (trk_deflate_bgzf) when the process has a device engine
            eng = None
            if _knobs.env('TRK_DEVICE_DEFLATE', '0') == '1':
                from . import runtime
                try:
                    eng = getattr(runtime.get_compute(), 'eng', None)
                except Exception:
                    eng = None
            self._fh = BgzfWriter(path, level=int(_knobs.env('TRK_ZIP_LEVEL', '6')), engine=eng)
            self._recs = []
        else:
            self._fh = open(path, 'w')

    def _drain(self):
        """Wait for the block the writer thread holds (order of the output; its exception, if any, is raised here)."""
        f, self._pending = self._pending, None
        if f is not None:
            f.result()

    def _header(self):
        t = self._tmpl
        lines = list(t._header_lines)
        if not t.has_pass_filter:
            at = 1 if lines and lines[0].startswith('##fileformat') else 0
            lines.insert(at, '##FILTER=<ID=PASS,Description="All filters passed">')
        for c in t.contigs_seen:
            lines.append('##contig=<ID=%s>' % c)
        self._fh.write('\n'.join(lines + [t._chrom_line]) + '\n')
        self._wrote_header = True

    NOTE_NATIVE_MIN = 1 << 20     # bytes of a block from which its record places are found by libtrk

    def _note_block(self, data, at):
        """``_note`` for a block of the batch writer (a memoryview of ~150 MB): the newlines by libtrk (memchr on its worker
        pool), then the first eight columns of every line -- a slice of the line's head, never the sample columns."""
        import numpy as np
        from .bgzf import _native_lib
        from .tabix import record_interval
        lib = _native_lib()
        n = len(data)
        if lib is None or n < self.NOTE_NATIVE_MIN:
            return self._note(bytes(data), at)
        lib.trk_text_newlines.restype = ctypes.c_int64
        lib.trk_text_newlines.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
        arr = np.frombuffer(data, dtype=np.uint8)
        cap = max(1024, n // 64)
        while True:
            nl = np.empty(cap, dtype=np.int64)
            got = int(lib.trk_text_newlines(arr.ctypes.data, n, nl.ctypes.data, cap))
            if got <= cap:
                break
            cap = got
        # ... then the places and intervals of the record lines, by libtrk as well (a loop over a block's 1 700 lines here
        # held the interpreter's lock for 8 ms per block -- a third of the writer thread's time with the members made on the
        # device, and time the caller's thread waited for the lock)
        lib.trk_text_record_places.restype = ctypes.c_int64
        lib.trk_text_record_places.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                                               ctypes.c_void_p, ctypes.c_size_t]
        cap = got + 1
        rows = np.empty((cap, 8), dtype=np.int64)
        k = int(lib.trk_text_record_places(arr.ctypes.data, n, nl.ctypes.data, got, rows.ctypes.data, cap))
        rows = rows[:k]
        mv = memoryview(data)
        names = []
        for r in np.flatnonzero(rows[:, 6]).tolist():              # where the sequence changes: its name
            names.append(bytes(mv[rows[r, 4]:rows[r, 4] + rows[r, 5]]).decode())
        for r in np.flatnonzero(rows[:, 7]).tolist():              # a head the scanner leaves to this side
            e = int(rows[r, 1])
            if e > int(rows[r, 0]) and mv[e - 1] == 10:
                e -= 1
            _, rows[r, 2], rows[r, 3] = record_interval(bytes(mv[rows[r, 0]:e]))
        which = np.cumsum(rows[:, 6]) - 1                          # index into names
        rows[:, 0] += at
        rows[:, 1] += at
        self._recs.append((names, np.column_stack((which, rows[:, 2], rows[:, 3], rows[:, 0], rows[:, 1]))))

    def _note(self, data, at):
        """The record lines of ``data`` (bytes-like, whole lines), which will lie at byte ``at`` of the text."""
        from .tabix import record_interval
        recs, n, pos = self._recs, len(data), 0
        find = data.find
        while pos < n:
            nl = find(b'\n', pos)
            if nl < 0:
                nl = n
            if nl > pos and data[pos:pos + 1] != b'#':
                # the first eight columns lie in front of the eighth tab: no copy of a 60 KB line for them
                t = pos
                for _ in range(8):
                    t = find(b'\t', t, nl) + 1
                    if t == 0:
                        t = nl + 1
                        break
                chrom, beg, end = record_interval(bytes(data[pos:t - 1]))
                recs.append((chrom, beg, end, at + pos, at + min(nl + 1, n)))
            pos = nl + 1

    def write_record(self, variant):
        self._drain()
        if not self._wrote_header:
            self._header()
        text = str(variant)
        if self._recs is not None:
            text = text.encode()
            self._note(text, self._fh.text_bytes)
        self._fh.write(text)

    def write_text(self, text):
        """Already formatted record lines (merged shards of a multi-process run)."""
        self._drain()
        if not self._wrote_header:
            self._header()
        if self._recs is not None:
            text = text.encode() if isinstance(text, str) else text
            self._note(text, self._fh.text_bytes)
        self._fh.write(text)

    def write_bytes(self, data):
        """Already formatted record lines as bytes / a memoryview (the batch record writer): no decode - encode
        round trip through the text layer.  The block is written (and, for --zip, compressed) by a background thread
        while the caller goes on to its next batch; at most one block is in flight, so the order of the output and the
        memory held are those of the synchronous writer."""
        self._drain()
        if not self._wrote_header:
            self._header()
        import sys
        import time
        timing = bool(_knobs.lab('TRK_WRITE_TIMING'))
        raw = getattr(self._fh, 'buffer', None)
        if raw is not None:                      # text file: flush what the text layer holds, then the raw bytes
            self._fh.flush()
            job = lambda: raw.write(data)
        else:
            block = data            # (bytes, a bytearray or a memoryview of the record writer's block: no copy)
            at = self._fh.text_bytes

            def job():
                t0 = time.perf_counter()
                self._note_block(block, at)     # (on the writer thread, beside the caller's next batch)
                t1 = time.perf_counter()
                self._fh.write(block)
                if timing:
                    print('[writer] places noted in %.1f ms, members made and written in %.1f ms' % ((t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3), file=sys.stderr)
        if timing:
            inner = job

            def job():
                t = time.perf_counter()
                inner()
                print('[writer] block of %.0f MB written in %.1f ms' % (len(data) / 1e6, (time.perf_counter() - t) * 1e3), file=sys.stderr)
        if _knobs.lab('TRK_ASYNC_WRITE', '1') == '0':
            job()
            return
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(1)
        self._pending = self._pool.submit(job)

    def close(self):
        try:
            self._drain()
        finally:
            if self._pool is not None:
                self._pool.shutdown(wait=True)
                self._pool = None
        if not self._wrote_header:
            self._header()
        self._fh.close()

    def write_index(self):
        """``<path>.tbi`` from the places noted while writing (after close(); --zip only).  ValueError when the records
        were not sorted by position -- what `tabix` refuses."""
        from . import tabix
        tb = tabix.TabixBuilder(self.path)
        vo = self._fh.voffset
        for rec in self._recs:
            if len(rec) == 2:                    # a block of the batch writer: (sequence names, rows)
                names, rows = rec
                for w, beg, end, a, b in rows.tolist():
                    tb.add(names[w], beg, end, vo(a), vo(b))
            else:
                chrom, beg, end, a, b = rec
                tb.add(chrom, beg, end, vo(a), vo(b))
        idx = tb.finish()
        tabix.write(idx, self.path + '.tbi')
        self.wrote_index = True
        return idx
