"""Python face of the native VCF / BGZF reader (include/trk_vcf.h, csrc/trk_vcf.cpp).

``NativeVCFReader`` has the surface of ``vcfio.VCFReader`` (and therefore of the part of
``cyvcf2.VCF`` the hot path uses) but decodes batches of records in C++ threads straight into
numpy arrays: the genotype tensor, the phase bits and any FORMAT field selected with
``select_format``.  Records come out as ``vcfio.Variant`` objects whose genotype matrix and
selected FORMAT arrays are views into those batch arrays; other FORMAT fields are still
available (decoded lazily in Python from the record text)."""
import ctypes as C
import os

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
                ('field_off', C.POINTER(C.c_int32))]


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
        lib.trk_vcf_n_samples.argtypes = [vp]
        lib.trk_vcf_sample_name.argtypes = [vp, C.c_int]
        lib.trk_vcf_sample_name.restype = C.c_char_p
        lib.trk_vcf_select_format.argtypes = [vp, C.c_char_p, C.c_int, C.c_int]
        lib.trk_vcf_read_batch.argtypes = [vp, C.c_int, C.c_int, C.POINTER(_Batch)]
        lib.trk_vcf_seek.argtypes = [vp, C.c_uint64]
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

    def _next_batch(self):
        S = self.n_samples
        P = self._max_ploidy
        n = self._batch_records or max(1, min(4096, (1 << 22) // max(S, 1)))
        while True:
            gt = np.empty((n, S, P), dtype=np.int16)
            ph = np.empty((n, S), dtype=np.uint8)
            lp = np.empty(n, dtype=np.uint8)
            planes = [np.empty((n, S, nc), dtype=dt) for _, _, nc, dt in self._selected]
            parr = (C.c_void_p * max(len(planes), 1))(*[p.ctypes.data for p in planes])
            b = _Batch()
            b.gt, b.phased, b.locus_ploidy = gt.ctypes.data, ph.ctypes.data, lp.ctypes.data
            b.planes = C.cast(parr, C.POINTER(C.c_void_p))
            rc = self._lib.trk_vcf_read_batch(self._h, n, P, C.byref(b))
            if rc == 5 and b'haplotypes' in self._lib.trk_vcf_last_error(self._h):
                # a genotype with more haplotypes than the tensor has columns (0/1/1 in a file read as diploid):
                # the reader has consumed nothing, decode the same lines again with a wider tensor (cyvcf2 sizes
                # its genotype array per record; the kernels take any ploidy up to 8)
                if P >= 64:
                    raise ValueError("a record of %s has more than 64 haplotypes per genotype" % self.path)
                P = self._max_ploidy = 2 * P
                continue
            if rc != 0:
                raise ValueError(self._lib.trk_vcf_last_error(self._h).decode())
            break
        m = b.n_records
        rows = []
        for i in range(m):
            # the nine fixed columns as text now; the sample columns (most of the line) stay bytes until a field
            # that was not decoded natively is asked for
            f9 = int(b.field_off[i * 10 + 9])
            n_line = b.line_end[i] - b.line_off[i]
            if 0 < f9 < n_line:
                line = C.string_at(b.text + b.line_off[i], f9 - 1).decode()
                tail = C.string_at(b.text + b.line_off[i] + f9, n_line - f9)
            else:
                line, tail = C.string_at(b.text + b.line_off[i], n_line).decode(), None
            pl = int(lp[i])
            g = np.empty((S, pl + 1), dtype=np.int16)
            g[:, :pl] = gt[i, :, :pl]
            g[:, pl] = ph[i]
            native = {k: planes[j][i] for j, (k, _, _, _) in enumerate(self._selected)}
            rows.append((line, g, native, tail))
        self._rows, self._row_i = rows, 0
        self._eof = m == 0
        self.last_batch = dict(gt=gt[:m], phased=ph[:m], locus_ploidy=lp[:m],
                               planes={k: planes[j][:m] for j, (k, _, _, _) in enumerate(self._selected)})

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
        self._indexed_region = False
        tbi = self.path + '.tbi'
        if os.path.isfile(tbi) and os.environ.get('TRK_TABIX', '1') != '0':
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

    def _past_region(self, v):
        chrom, start, end = self._region
        if v.CHROM != chrom:
            return True
        return end is not None and v.POS > end

    def close(self):
        if self._h is not None:
            self._lib.trk_vcf_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
