"""BGZF writer (blocked gzip with the 'BC' extra field, SAM spec section 4.1) so that dumpSTR --zip output and large
synthetic test inputs are real bgzip files that htslib tools and the native reader's block-parallel inflate accept.

Round 6: the members are compressed by libtrk (``trk_bgzf_compress``, include/trk_vcf.h: libdeflate on the caller-side
worker pool of the library) -- the zlib members made on a thread pool of the interpreter (rounds 1-5, still here as the
definition and for a process that cannot load libtrk) ran at 45-100 MB/s: 15-30 s for the 1.5 GB a dumpSTR run of a
second's work writes.  Either way a member holds 0xff00 bytes of text, so the two paths cut a stream into the same
members and differ in the DEFLATE bytes only."""
import ctypes as C
import os
import struct
import zlib

_EOF = bytes.fromhex('1f8b08040000000000ff0600424302001b0003000000000000000000')
BLOCK = 0xff00
DEVICE_MEMBER = 16384     # bytes of text per member the device makes (include/trk.h: TRK_DEFLATE_MEMBER)


def _compress_block(args):
    raw, level = args
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    comp = c.compress(raw) + c.flush()
    bsize = len(comp) + 25
    hdr = struct.pack('<BBBBIBBHBBHH', 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6, ord('B'), ord('C'), 2, bsize)
    return hdr + comp + struct.pack('<II', zlib.crc32(raw) & 0xffffffff, len(raw))


_native = False       # False: not looked for yet; None: not there


def _native_lib():
    """libtrk's trk_bgzf_* entries, or None (no library, or a lab process that asks for the Python members:
    TRK_BGZF_PYTHON=1)."""
    global _native
    if _native is False:
        _native = None
        try:
            from . import _knobs, _lib
            if _knobs.lab('TRK_BGZF_PYTHON') != '1':
                lib = _lib.load()
                lib.trk_bgzf_bound.argtypes = [C.c_size_t]
                lib.trk_bgzf_bound.restype = C.c_size_t
                lib.trk_bgzf_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                                  C.POINTER(C.c_size_t)]
                lib.trk_bgzf_compress.restype = C.c_int
                lib.trk_bgzf_member_offsets.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
                lib.trk_bgzf_member_offsets.restype = C.c_int64
                _native = lib
        except Exception:          # (no libtrk.so here: the Python members)
            _native = None
    return _native


def _address(data):
    """(object that keeps the buffer alive, address of its first byte) of bytes / bytearray / a memoryview."""
    import numpy as np
    arr = np.frombuffer(data, dtype=np.uint8)
    return arr, arr.ctypes.data


class BgzfWriter:
    """``write`` collects text; whole members leave ``CHUNK`` bytes at a time -- one native call that compresses its
    members on the library's worker pool (ctypes drops the GIL), or ``BATCH`` zlib members on a thread pool of the
    interpreter when the library is not there -- and are written in order."""
    BATCH = 64
    CHUNK = 256 * BLOCK            # ~16 MB of text per native call

    def __init__(self, path, level=6, threads=None, engine=None):
        """``engine``: a device engine -- blocks of at least DEVICE_MIN bytes are deflated on the GPU (trk_deflate_bgzf:
        the text goes up, the members come down; the host computes the CRCs), smaller writes and the stream's tail by the
        host compressor (whose members hold 0xff00 bytes of text; the device's hold DEVICE_MEMBER)."""
        self._engine = engine
        from . import _knobs
        self._timing = bool(_knobs.lab('TRK_WRITE_TIMING'))
        self._fh = open(path, 'wb')
        self._buf = bytearray()
        self._level = level
        self._pending = []
        self._threads = threads
        self._lib = _native_lib()
        self._out = None           # the native path's output buffer, reused
        self._coff = [0]           # compressed offset of member k; the last entry: behind the members written so far
        self._toff = [0]           # ... and the offset in the TEXT of its first byte (members need not be of one size:
                                   # the device's hold DEVICE_MEMBER bytes, the host's BLOCK)
        self.text_bytes = 0        # bytes of text handed to write() so far
        n = threads if threads is not None else min(16, os.cpu_count() or 1)
        self._pool = None
        if n > 1 and self._lib is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=n)

    # ---- native members -------------------------------------------------------------------------------------------
    def _native_emit(self, data, at, n):
        """Members of data[at : at + n] (bytes or bytearray; n a multiple of BLOCK, or the stream's tail) through
        trk_bgzf_compress, written out.  No copy of the text: the library reads the caller's bytes where they lie."""
        lib = self._lib
        need = lib.trk_bgzf_bound(n)
        if self._out is None or len(self._out) < need:
            self._out = bytearray(need)
        hold, base = _address(data)
        dst = (C.c_char * len(self._out)).from_buffer(self._out)
        got = C.c_size_t(0)
        try:
            rc = lib.trk_bgzf_compress(C.c_void_p(base + at), n, int(self._level), int(self._threads or 0), dst,
                                       len(self._out), C.byref(got))
        finally:
            del hold, dst          # (a bytearray cannot be resized while a view of it lives)
        if rc != 0:
            raise OSError("trk_bgzf_compress failed (%d)" % rc)
        self._members(self._out, got.value, n, BLOCK)
        self._fh.write(memoryview(self._out)[:got.value])

    def _members(self, out, end, n_text, member):
        """The members of out[0 : end) -- ``n_text`` bytes of text, ``member`` per member -- join the offset tables.  Where
        they begin: libtrk walks the chain of their BSIZE fields (a block of the device's has 9 000 members: not a loop for
        the interpreter, whose lock the caller's thread waits for meanwhile)."""
        import numpy as np
        coff, toff = self._coff, self._toff
        base, tbase = coff[-1], toff[-1]
        lib = self._lib
        if lib is not None and end:
            cap = n_text // member + 2
            offs = np.empty(cap, dtype=np.uint64)
            hold, addr = _address(out)
            try:
                k = int(lib.trk_bgzf_member_offsets(C.c_void_p(addr), end, offs.ctypes.data, cap))
            finally:
                del hold
            if k < 0 or k > cap:
                raise OSError("BgzfWriter: the members made do not form a chain")
            starts = offs[:k].astype(np.int64)
            coff.extend((base + starts[1:]).tolist())
            coff.append(base + end)
            toff.extend((tbase + np.minimum(np.arange(1, k + 1, dtype=np.int64) * member, n_text)).tolist())
            return
        pos, k = 0, 0
        while pos < end:           # the members' sizes (BSIZE at byte 16 of each): where member k + 1 begins
            pos += (out[pos + 16] | (out[pos + 17] << 8)) + 1
            k += 1
            coff.append(base + pos)
            toff.append(tbase + min(k * member, n_text))

    DEVICE_MIN = 8 << 20

    def _device_emit(self, data, at, n):
        """Members of data[at : at + n] made on the device (DEVICE_MEMBER bytes of text each)."""
        import time
        t0 = time.perf_counter()
        hold, base = _address(data)
        try:
            out = self._engine.deflate_bgzf(None, address=base + at, nbytes=n)
        finally:
            del hold
        t1 = time.perf_counter()
        self._members(out, len(out), n, DEVICE_MEMBER)
        t2 = time.perf_counter()
        self._fh.write(out)
        if self._timing:
            import sys
            print('[bgzf] %.0f MB: device deflate %.1f ms, member table %.1f ms, write of %.0f MB %.1f ms' % (
                n / 1e6, (t1 - t0) * 1e3, (t2 - t1) * 1e3, len(out) / 1e6, (time.perf_counter() - t2) * 1e3), file=sys.stderr)

    def _emit_chunks(self, data, at, n):
        if self._engine is not None and n >= self.DEVICE_MIN:      # (any length: a call's last member is as long as what is left)
            try:
                self._device_emit(data, at, n)
                return
            except Exception as e:        # (out of device memory, a context that went away): the host makes the members from here on
                import sys
                sys.stderr.write("BgzfWriter: the device deflate failed (%s); the host compressor takes over\n" % e)
                self._engine = None
        for o in range(0, n, self.CHUNK):
            self._native_emit(data, at + o, min(self.CHUNK, n - o))

    def _native_write(self, data):
        if not isinstance(data, (bytes, bytearray, memoryview)):
            data = bytes(data)
        if len(data) >= self.CHUNK:
            # a block of a batch writer (150 MB): what is kept of the text before it is topped up to whole members and
            # leaves, then the block's own whole members go straight from the caller's bytes (no copy); its tail is kept
            at = 0
            if self._buf:
                at = -len(self._buf) % BLOCK
                self._buf += data[:at]
                part, self._buf = self._buf, bytearray()
                self._emit_chunks(part, 0, len(part))
            whole = (len(data) - at) // BLOCK * BLOCK
            self._emit_chunks(data, at, whole)
            self._buf += data[at + whole:]
            return
        self._buf += data
        if len(self._buf) >= self.CHUNK:
            whole = len(self._buf) // BLOCK * BLOCK
            part = self._buf
            self._buf = bytearray(part[whole:])
            self._emit_chunks(part, 0, whole)

    def voffset(self, text_off):
        """BGZF virtual offset of byte ``text_off`` of the text, as htslib's bgzf_tell reports it while reading: the
        member's offset in the file << 16 | the byte's offset in the member's text; the position behind the last byte of
        a member is the START of the next one.  After close() only (every member has been written)."""
        if text_off >= self.text_bytes:
            return self._coff[-1] << 16
        import bisect
        k = bisect.bisect_right(self._toff, text_off) - 1
        return (self._coff[k] << 16) | (text_off - self._toff[k])

    def write(self, data):
        if isinstance(data, str):
            data = data.encode()
        self.text_bytes += len(data)
        if self._lib is not None:
            self._native_write(data)
            return
        self._buf += data
        if len(self._buf) >= BLOCK:
            n = len(self._buf) // BLOCK
            view = bytes(self._buf[:n * BLOCK])
            del self._buf[:n * BLOCK]
            for i in range(n):
                self._pending.append(view[i * BLOCK:(i + 1) * BLOCK])
            if len(self._pending) >= self.BATCH:
                self._flush_blocks()

    def _flush_blocks(self):
        jobs = [(raw, self._level) for raw in self._pending]
        self._pending = []
        out = self._pool.map(_compress_block, jobs) if self._pool is not None else map(_compress_block, jobs)
        for blk, (raw, _) in zip(out, jobs):
            self._fh.write(blk)
            self._coff.append(self._coff[-1] + len(blk))
            self._toff.append(self._toff[-1] + len(raw))

    def close(self):
        if self._lib is not None:
            if self._buf:
                part, self._buf = self._buf, bytearray()
                self._native_emit(part, 0, len(part))
        else:
            if self._buf:
                self._pending.append(bytes(self._buf))
                self._buf = bytearray()
            self._flush_blocks()
        if self._pool is not None:
            self._pool.shutdown()
        self._fh.write(_EOF)
        self._fh.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
