"""Minimal BGZF writer (blocked gzip with the 'BC' extra field, SAM spec section 4.1) so that
dumpSTR --zip output and large synthetic test inputs are real bgzip files that htslib tools and
the native reader's block-parallel inflate accept."""
import struct
import zlib

_EOF = bytes.fromhex('1f8b08040000000000ff0600424302001b0003000000000000000000')
BLOCK = 0xff00


class BgzfWriter:
    def __init__(self, path, level=6):
        self._fh = open(path, 'wb')
        self._buf = bytearray()
        self._level = level

    def write(self, data):
        if isinstance(data, str):
            data = data.encode()
        self._buf += data
        while len(self._buf) >= BLOCK:
            self._block(bytes(self._buf[:BLOCK]))
            del self._buf[:BLOCK]

    def _block(self, raw):
        c = zlib.compressobj(self._level, zlib.DEFLATED, -15)
        comp = c.compress(raw) + c.flush()
        bsize = len(comp) + 25
        hdr = struct.pack('<BBBBIBBHBBHH', 0x1f, 0x8b, 8, 4, 0, 0, 0xff, 6, ord('B'), ord('C'), 2, bsize)
        self._fh.write(hdr + comp + struct.pack('<II', zlib.crc32(raw) & 0xffffffff, len(raw)))

    def close(self):
        if self._buf:
            self._block(bytes(self._buf))
            self._buf = bytearray()
        self._fh.write(_EOF)
        self._fh.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
