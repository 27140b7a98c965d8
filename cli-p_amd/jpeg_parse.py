"""Marker walk of a baseline JPEG file for the device decoder (csrc/jpeg.hip; SURVEY.md §8(f) next-1, reference
build-index.py:47). numpy only - the decode workers (decode_worker.py) import this file by path, without torch.

`parse` lets through what the device decodes - 8-bit baseline / extended-sequential Huffman, one interleaved scan, grey or
YCbCr with luma sampling 1x1 / 2x1 / 2x2 and 1x1 chroma, with or without restart intervals - and raises `Unsupported` for
everything else (progressive, CMYK / RGB-coded, 12-bit, arithmetic coding, odd sampling, not a JPEG): those files stay with Pillow.
That is a choice of decoder per file format, made on the host from the file's own header.
"""
import re

import numpy as np

TABLE_BYTES = 288          # a raw table: DHT's 16 counts + up to 256 symbols (zero padded to 272) + class (0 DC, 1 AC) + 15 zero bytes
MAX_STREAM = 1 << 28
MAX_INTERVALS = 1 << 16
_NATURAL = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14,
                     21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53,
                     60, 61, 54, 47, 55, 62, 63])
_INV_NATURAL = np.argsort(_NATURAL)


_MARKER = re.compile(rb"\xff[^\x00]")


class Unsupported(Exception):
    """Not a file for the device decoder; Pillow's path takes it."""


class Parsed:
    __slots__ = ("width", "height", "ncomp", "hs", "vs", "quant", "tables", "stream", "ri", "starts", "stuffed")
    # ri: restart interval in MCUs (0: none); starts: uint32 byte offsets into `stream` of the restart intervals (None without);
    # stuffed: 1 = `stream` is the file's segment as it is, 0xFF00 stuffing included (parse(keep_stuffing=True))

    def mcus(self):
        hs, vs = (self.hs, self.vs) if self.ncomp == 3 else (1, 1)
        return -(-self.width // (8 * hs)) * -(-self.height // (8 * vs))

    def blocks(self):
        return self.mcus() * (self.hs * self.vs + 2 if self.ncomp == 3 else 1)


def parse(data, keep_stuffing=False):
    """JPEG file bytes -> Parsed (header fields, quantisation steps in natural order, the six Huffman tables a scan can
    name as raw 288-byte records, the entropy-coded segment without byte stuffing). Raises Unsupported.
    keep_stuffing: a file without restart intervals that ends with its EOI marker is only SLICED - the segment keeps its stuffing
    (Parsed.stuffed = 1) and the device removes it and looks for markers inside (csrc/jpeg.hip jpeg_unstuff_kernel): no pass over
    the data on the host at all."""
    if len(data) < 4 or data[0] != 0xFF or data[1] != 0xD8:
        raise Unsupported("not a JPEG file")
    n = len(data)
    i = 2
    qt = {}
    huff = {}
    frame = None
    ri = 0
    jfif = adobe = False
    adobe_tf = 0
    while True:
        if i + 4 > n or data[i] != 0xFF:
            raise Unsupported("marker expected")
        m = data[i + 1]
        if m == 0xFF:
            i += 1
            continue
        if m == 0xD8 or m == 0x01 or 0xD0 <= m <= 0xD7:
            i += 2
            continue
        L = (data[i + 2] << 8) | data[i + 3]
        if L < 2 or i + 2 + L > n:
            raise Unsupported("truncated segment")
        if m == 0xDB:
            k = i + 4
            while k < i + 2 + L:
                if data[k] >> 4 or k + 65 > i + 2 + L:
                    raise Unsupported("quantisation table")
                qt[data[k] & 15] = data[k + 1:k + 65]
                k += 65
        elif m == 0xC0 or m == 0xC1:
            if frame is not None or L < 11 or data[i + 4] != 8:
                raise Unsupported("frame header")
            nf = data[i + 9]
            if L != 8 + 3 * nf:
                raise Unsupported("frame header")
            frame = ((data[i + 5] << 8) | data[i + 6], (data[i + 7] << 8) | data[i + 8],
                     [(data[i + 10 + 3 * c], data[i + 11 + 3 * c] >> 4, data[i + 11 + 3 * c] & 15, data[i + 12 + 3 * c]) for c in range(nf)])
        elif 0xC2 <= m <= 0xCF and m != 0xC4 and m != 0xC8 and m != 0xCC:
            raise Unsupported("not a baseline frame")
        elif m == 0xCC:
            raise Unsupported("arithmetic coding")
        elif m == 0xC4:
            k = i + 4
            while k < i + 2 + L:
                if k + 17 > i + 2 + L:
                    raise Unsupported("Huffman table")
                cnt = sum(data[k + 1:k + 17])
                if cnt > 256 or k + 17 + cnt > i + 2 + L or (data[k] >> 4) > 1 or (data[k] & 15) > 3:
                    raise Unsupported("Huffman table")
                if (data[k] >> 4) == 0 and cnt and max(data[k + 17:k + 17 + cnt]) > 15:
                    raise Unsupported("DC Huffman table")               # libjpeg refuses such a table (jdhuff.c)
                huff[data[k]] = bytes(data[k + 1:k + 17 + cnt]).ljust(272, b"\0") + bytes([data[k] >> 4]) + b"\0" * 15
                k += 17 + cnt
        elif m == 0xDD:
            if L != 4:
                raise Unsupported("restart interval")
            ri = (data[i + 4] << 8) | data[i + 5]
        elif m == 0xE0 and data[i + 4:i + 9] == b"JFIF\0":
            jfif = True
        elif m == 0xEE and L >= 14 and data[i + 4:i + 9] == b"Adobe":
            adobe, adobe_tf = True, data[i + 15]
        elif m == 0xDA:
            break
        i += 2 + L
    if frame is None:
        raise Unsupported("no frame header")
    height, width, comps = frame
    nc = len(comps)
    if width == 0 or height == 0 or nc not in (1, 3):
        raise Unsupported("frame")
    ns = data[i + 4]
    if ns != nc or L != 6 + 2 * ns or data[i + 5 + 2 * ns] != 0 or data[i + 6 + 2 * ns] != 63 or data[i + 7 + 2 * ns] != 0:
        raise Unsupported("scan header")
    out = Parsed()
    out.width, out.height, out.ncomp = width, height, nc
    tables = []
    quant = np.zeros((3, 64), np.uint8)
    for c in range(nc):
        cid, td_ta = data[i + 5 + 2 * c], data[i + 6 + 2 * c]
        if cid != comps[c][0]:
            raise Unsupported("scan order")
        dc, ac = huff.get(td_ta >> 4), huff.get(0x10 | (td_ta & 15))
        q = qt.get(comps[c][3])
        if dc is None or ac is None or q is None:
            raise Unsupported("missing table")
        tables += [dc, ac]
        quant[c] = np.frombuffer(q, np.uint8)[_INV_NATURAL]
    while len(tables) < 6:
        tables += tables[:2]
    if nc == 3:
        ids = (comps[0][0], comps[1][0], comps[2][0])
        # jdapimin.c default_decompress_parms: which three-component files are YCbCr
        ycc = True if jfif else (adobe_tf != 0) if adobe else ids != (0x52, 0x47, 0x42)
        if not ycc:
            raise Unsupported("RGB-coded JPEG")
        if (comps[1][1], comps[1][2], comps[2][1], comps[2][2]) != (1, 1, 1, 1) or (comps[0][1], comps[0][2]) not in ((1, 1), (2, 1), (2, 2)):
            raise Unsupported("sampling factors")
        out.hs, out.vs = comps[0][1], comps[0][2]
        if out.hs == 2 and (width + 1) // 2 <= 2:
            raise Unsupported("too narrow for fancy upsampling")
    else:
        out.hs = out.vs = 1
    out.quant, out.tables = quant, tables
    # the entropy-coded segment ends at the first marker that is not a stuffed 0xFF00 (C-speed searches: a photo's segment
    # holds thousands of stuffed bytes)
    i += 2 + L
    out.ri, out.starts, out.stuffed = ri, None, 0
    if keep_stuffing and not ri and n - i >= 2 and data[n - 2] == 0xFF and data[n - 1] == 0xD9:
        out.stream = data[i:n - 2]
        out.stuffed = 1
    elif ri:
        # restart intervals: RSTn markers, numbered 0..7 in turn (jdmarker.c read_restart_marker), separate them; every
        # interval starts on a byte with fresh DC predictions - an independent chain for the device. The markers are dropped.
        want = -(-out.mcus() // ri)
        if want > MAX_INTERVALS:
            raise Unsupported("too many restart intervals")
        parts, count, seg = [], 0, i
        for m_ in _MARKER.finditer(data, i):
            j = m_.start()
            nxt = data[j + 1]
            if nxt == 0xD0 + (count & 7) and count + 1 < want:
                parts.append(data[seg:j].replace(b"\xff\x00", b"\xff"))
                count += 1
                seg = j + 2
                continue
            if nxt != 0xD9 or count + 1 != want:
                raise Unsupported("marker inside the scan")
            parts.append(data[seg:j].replace(b"\xff\x00", b"\xff"))
            break
        else:
            raise Unsupported("no end of image")
        lens = np.fromiter((len(x) for x in parts), dtype=np.int64, count=len(parts))
        out.starts = (np.cumsum(lens) - lens).astype(np.uint32)
        out.stream = b"".join(parts)
    else:
        m_ = _MARKER.search(data, i)
        if m_ is None:
            raise Unsupported("no end of image")
        j = m_.start()
        nxt = data[j + 1]
        if nxt == 0xFF:                                      # fill bytes in front of a marker: rare, walk them
            while j + 1 < n and data[j + 1] == 0xFF:
                j += 1
            if j + 1 >= n:
                raise Unsupported("no end of image")
            nxt = data[j + 1]
            if nxt == 0:
                raise Unsupported("fill bytes inside the scan")
        if nxt != 0xD9:
            raise Unsupported("marker inside the scan")         # further scans
        out.stream = data[i:j].replace(b"\xff\x00", b"\xff")
    if len(out.stream) >= MAX_STREAM:
        raise Unsupported("too large")
    return out
