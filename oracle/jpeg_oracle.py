"""jpeg_oracle.py — TEST INFRASTRUCTURE, not product code.

CPU restatement (plain Python / numpy, integer arithmetic) of what `Image.open(tfn)` (reference build-index.py:47, the
first step of `transform(...)` at :48) computes for a baseline JPEG file: the pixels Pillow hands to the transform.

The arithmetic lives in third-party dependencies that are ABSENT from /root/reference: Pillow (unpinned in the
reference's setup.sh) and, inside it, libjpeg-turbo (this image: Pillow 12.2.0, libjpeg-turbo with the v6b API,
`PIL.features.version("jpg") == "6.2"`). This file restates their published algorithm for the subset the device
decoder takes (csrc/jpeg.hip): 8-bit baseline / extended-sequential Huffman JPEG (SOF0 / SOF1), one interleaved scan,
1 or 3 components, luma sampling 1x1, 2x1 or 2x2 with 1x1 chroma, with or without restart intervals -
  * jdhuff.c   canonical Huffman decode, EXTEND, DC prediction, jpeg_natural_order;
  * jidctint.c `jpeg_idct_islow` (JDCT_ISLOW is Pillow's method), dequantisation folded in, output range-limited;
  * jdsample.c `h2v1_fancy_upsample` / `h2v2_fancy_upsample` (do_fancy_upsampling is libjpeg's default) with
    jdmainct.c's context rows (edge rows duplicated at the top and the bottom of the image);
  * jdcolor.c  `ycc_rgb_convert`'s 16-bit fixed-point tables; grayscale -> RGB as `Image.convert("RGB")` replicates it.
PINNED against Pillow itself, which is installed here and on the GPU box: tests/test_jpeg.py decodes seeded images of
every supported sampling, odd sizes included, with both and demands identical bytes.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np

NATURAL = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14,
                    21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53,
                    60, 61, 54, 47, 55, 62, 63], dtype=np.int32)      # jutils.c jpeg_natural_order


class Unsupported(Exception):
    """A JPEG the device decoder does not take (progressive, CMYK, ...): Pillow's path decodes it."""


def parse(data):
    """Marker walk -> dict(width, height, comps=[(id, h, v, tq)], qt={id: 64 ints in zigzag order},
    huff={(cls, id): (bits[16], vals)}, scan=[(comp index, td, ta)], stream=unstuffed entropy-coded bytes, ri=restart interval
    in MCUs (0: none), starts=[byte offset in stream of every restart interval])."""
    if data[:2] != b"\xff\xd8":
        raise Unsupported("no SOI")
    i, n = 2, len(data)
    qt, huff, comps, scan, width, height, ri = {}, {}, None, None, 0, 0, 0
    jfif = adobe = False
    adobe_tf = 0
    while True:
        if i + 4 > n or data[i] != 0xFF:
            raise Unsupported("marker expected")
        m = data[i + 1]
        if m == 0xFF:
            i += 1
            continue
        L = int.from_bytes(data[i + 2:i + 4], "big")
        seg = data[i + 4:i + 2 + L]
        if len(seg) != L - 2:
            raise Unsupported("truncated segment")
        if m == 0xDB:
            k = 0
            while k < len(seg):
                if seg[k] >> 4:
                    raise Unsupported("16-bit quantisation table")
                qt[seg[k] & 15] = list(seg[k + 1:k + 65])
                k += 65
        elif m in (0xC0, 0xC1):
            if seg[0] != 8:
                raise Unsupported("precision")
            height, width = int.from_bytes(seg[1:3], "big"), int.from_bytes(seg[3:5], "big")
            comps = [(seg[6 + 3 * c], seg[7 + 3 * c] >> 4, seg[7 + 3 * c] & 15, seg[8 + 3 * c]) for c in range(seg[5])]
        elif 0xC2 <= m <= 0xCF and m not in (0xC4, 0xC8, 0xCC):
            raise Unsupported("not a baseline frame")
        elif m == 0xC4:
            k = 0
            while k < len(seg):
                bits = list(seg[k + 1:k + 17])
                cnt = sum(bits)
                huff[(seg[k] >> 4, seg[k] & 15)] = (bits, list(seg[k + 17:k + 17 + cnt]))
                k += 17 + cnt
        elif m == 0xDD:
            ri = int.from_bytes(seg[:2], "big")
        elif m == 0xE0 and seg[:5] == b"JFIF\0":
            jfif = True
        elif m == 0xEE and seg[:5] == b"Adobe" and len(seg) >= 12:
            adobe, adobe_tf = True, seg[11]
        elif m == 0xDA:
            ns = seg[0]
            scan = [(seg[1 + 2 * c], seg[2 + 2 * c] >> 4, seg[2 + 2 * c] & 15) for c in range(ns)]
            i += 2 + L
            break
        i += 2 + L
    if comps is None or width == 0 or height == 0:
        raise Unsupported("no frame")
    if len(comps) not in (1, 3) or len(scan) != len(comps) or [s[0] for s in scan] != [c[0] for c in comps]:
        raise Unsupported("components / scans")
    if len(comps) == 3:
        ids = tuple(c[0] for c in comps)
        # jdapimin.c default_decompress_parms: which 3-component files are YCbCr
        ycc = True if jfif else (adobe_tf == 1) if adobe else ids != (0x52, 0x47, 0x42)
        if adobe and not jfif and adobe_tf not in (0, 1):
            ycc = True
        if not ycc:
            raise Unsupported("RGB-coded JPEG")
        if (comps[1][1], comps[1][2], comps[2][1], comps[2][2]) != (1, 1, 1, 1) or (comps[0][1], comps[0][2]) not in ((1, 1), (2, 1), (2, 2)):
            raise Unsupported("sampling factors")
    # entropy-coded segment: up to the first marker that is not a stuffed 0xFF00; RSTn markers (jdmarker.c read_restart_marker:
    # numbered 0..7 in turn) separate the restart intervals and are dropped
    parts, j, count = [], i, 0
    while True:
        j = data.find(b"\xff", j)
        if j < 0 or j + 1 >= n:
            raise Unsupported("no EOI")
        nxt = data[j + 1]
        if nxt == 0:
            j += 2
            continue
        if ri and nxt == 0xD0 + (count & 7):
            parts.append(data[i:j].replace(b"\xff\x00", b"\xff"))
            count += 1
            i = j = j + 2
            continue
        if nxt != 0xD9:
            raise Unsupported("marker inside the scan")
        parts.append(data[i:j].replace(b"\xff\x00", b"\xff"))
        break
    starts = [0]
    for part in parts[:-1]:
        starts.append(starts[-1] + len(part))
    return dict(width=width, height=height, comps=comps, qt=qt, huff=huff, scan=scan, stream=b"".join(parts), ri=ri, starts=starts)


def _huff_lut(bits, vals):
    """16-bit look-up: code length and symbol for every 16-bit window (jdhuff.c jpeg_make_d_derived_tbl's code
    assignment: canonical, shortest first)."""
    ln = np.zeros(65536, np.int32)
    sy = np.zeros(65536, np.int32)
    code, k = 0, 0
    for length in range(1, 17):
        for _ in range(bits[length - 1]):
            lo = code << (16 - length)
            ln[lo:lo + (1 << (16 - length))] = length
            sy[lo:lo + (1 << (16 - length))] = vals[k]
            code += 1
            k += 1
        code <<= 1
    return ln.tolist(), sy.tolist()


def decode_coefficients(info):
    """-> int32 [blocks][64] in natural order, blocks in scan (MCU) order; DC values already predicted."""
    comps, scan = info["comps"], info["scan"]
    hmax = max(c[1] for c in comps)
    vmax = max(c[2] for c in comps)
    mx = -(-info["width"] // (8 * hmax))
    my = -(-info["height"] // (8 * vmax))
    order = []
    for ci, c in enumerate(comps):
        order += [ci] * (c[1] * c[2] if len(comps) > 1 else 1)
    luts = {k: _huff_lut(*v) for k, v in info["huff"].items()}
    s = info["stream"] + b"\0" * 8
    win = [int.from_bytes(s[k:k + 4], "big") for k in range(len(s) - 3)]
    nbits = 8 * len(info["stream"])
    nblk = mx * my * len(order)
    out = np.zeros((nblk, 64), np.int32)
    pred = [0] * len(comps)
    p = 0
    nat = NATURAL.tolist()
    ri, starts = info.get("ri", 0), info.get("starts", [0])
    if ri and len(starts) != -(-(mx * my) // ri):
        raise Unsupported("restart intervals")
    for b in range(nblk):
        if ri and b % (ri * len(order)) == 0:            # jdhuff.c process_restart: byte-aligned start, DC predictions reset
            p = 8 * starts[b // (ri * len(order))]
            pred = [0] * len(comps)
        ci = order[b % len(order)]
        dl, ds = luts[(0, scan[ci][1])]
        al, as_ = luts[(1, scan[ci][2])]
        w = (win[p >> 3] >> (16 - (p & 7))) & 0xFFFF
        if dl[w] == 0:
            raise Unsupported("bad Huffman code")
        p += dl[w]
        sz = ds[w]
        if sz:
            v = (win[p >> 3] >> (32 - (p & 7) - sz)) & ((1 << sz) - 1)
            p += sz
            if v < (1 << (sz - 1)):
                v -= (1 << sz) - 1
            pred[ci] += v
        row = out[b]
        row[0] = pred[ci]
        k = 1
        while k < 64:
            w = (win[p >> 3] >> (16 - (p & 7))) & 0xFFFF
            if al[w] == 0:
                raise Unsupported("bad Huffman code")
            p += al[w]
            r, sz = as_[w] >> 4, as_[w] & 15
            if sz:
                k += r
                v = (win[p >> 3] >> (32 - (p & 7) - sz)) & ((1 << sz) - 1)
                p += sz
                if v < (1 << (sz - 1)):
                    v -= (1 << sz) - 1
                row[nat[k] if k < 64 else 63] = v
                k += 1
            elif r == 15:
                k += 16
            else:
                break
        if p > nbits:
            raise Unsupported("entropy-coded data ends early")
    return out, (mx, my, hmax, vmax, order)


def idct_islow(coef, q):
    """jidctint.c jpeg_idct_islow on [n][64] natural-order coefficients with the 64 quantisation steps q (natural order)
    -> uint8 [n][8][8]."""
    C = dict(f0298=2446, f0390=3196, f0541=4433, f0765=6270, f0899=7373, f1175=9633, f1501=12299, f1847=15137, f1961=16069,
             f2053=16819, f2562=20995, f3072=25172)
    x = (coef.astype(np.int64) * q.astype(np.int64)[None, :]).reshape(-1, 8, 8)

    def one_d(v, shift):                       # v[..., 8] along the last axis
        z2, z3 = v[..., 2], v[..., 6]
        z1 = (z2 + z3) * C["f0541"]
        t2 = z1 - z3 * C["f1847"]
        t3 = z1 + z2 * C["f0765"]
        t0 = (v[..., 0] + v[..., 4]) << 13
        t1 = (v[..., 0] - v[..., 4]) << 13
        t10, t13, t11, t12 = t0 + t3, t0 - t3, t1 + t2, t1 - t2
        o0, o1, o2, o3 = v[..., 7], v[..., 5], v[..., 3], v[..., 1]
        z1, z2, z3, z4 = o0 + o3, o1 + o2, o0 + o2, o1 + o3
        z5 = (z3 + z4) * C["f1175"]
        o0, o1, o2, o3 = o0 * C["f0298"], o1 * C["f2053"], o2 * C["f3072"], o3 * C["f1501"]
        z1, z2, z3, z4 = -z1 * C["f0899"], -z2 * C["f2562"], -z3 * C["f1961"] + z5, -z4 * C["f0390"] + z5
        o0, o1, o2, o3 = o0 + z1 + z3, o1 + z2 + z4, o2 + z2 + z3, o3 + z1 + z4
        r = np.stack([t10 + o3, t11 + o2, t12 + o1, t13 + o0, t13 - o0, t12 - o1, t11 - o2, t10 - o3], axis=-1)
        return (r + (1 << (shift - 1))) >> shift

    ws = one_d(x.transpose(0, 2, 1), 11).transpose(0, 2, 1)      # pass 1: columns, CONST_BITS - PASS1_BITS
    px = one_d(ws, 18)                                           # pass 2: rows, CONST_BITS + PASS1_BITS + 3
    return np.clip(px + 128, 0, 255).astype(np.uint8)


def _planes(info):
    coef, (mx, my, hmax, vmax, order) = decode_coefficients(info)
    comps = info["comps"]
    planes = []
    bpm = len(order)
    first = 0
    for ci, c in enumerate(comps):
        h, v = (c[1], c[2]) if len(comps) > 1 else (1, 1)
        q = np.zeros(64, np.int64)
        q[NATURAL] = info["qt"][c[3]]
        idx = (np.arange(mx * my)[:, None] * bpm + first + np.arange(h * v)[None, :]).reshape(-1)
        px = idct_islow(coef[idx], q).reshape(my, mx, v, h, 8, 8)
        planes.append(px.transpose(0, 2, 4, 1, 3, 5).reshape(my * v * 8, mx * h * 8))
        first += h * v
    return planes, hmax, vmax


def _h2v1_fancy(p, dw):
    """jdsample.c h2v1_fancy_upsample on rows p[:, :dw] -> [rows][2 dw]"""
    p = p[:, :dw].astype(np.int32)
    out = np.empty((p.shape[0], 2 * dw), np.int32)
    left = np.concatenate([p[:, :1], p[:, :-1]], axis=1)
    right = np.concatenate([p[:, 1:], p[:, -1:]], axis=1)
    out[:, 0::2] = (3 * p + left + 1) >> 2
    out[:, 1::2] = (3 * p + right + 2) >> 2
    out[:, 0] = p[:, 0]
    out[:, -1] = p[:, -1]
    return out


def _h2v2_fancy(p, dw, dh):
    """jdsample.c h2v2_fancy_upsample with jdmainct.c's context rows -> [2 dh][2 dw]"""
    p = p[:dh, :dw].astype(np.int32)
    up = np.concatenate([p[:1], p[:-1]], axis=0)
    dn = np.concatenate([p[1:], p[-1:]], axis=0)
    out = np.empty((2 * dh, 2 * dw), np.int32)
    for v, other in ((0, up), (1, dn)):
        s = 3 * p + other                                   # thiscolsum per column
        left = np.concatenate([s[:, :1], s[:, :-1]], axis=1)
        right = np.concatenate([s[:, 1:], s[:, -1:]], axis=1)
        row = np.empty((dh, 2 * dw), np.int32)
        row[:, 0::2] = (3 * s + left + 8) >> 4
        row[:, 1::2] = (3 * s + right + 7) >> 4
        row[:, 0] = (4 * s[:, 0] + 8) >> 4
        row[:, -1] = (4 * s[:, -1] + 7) >> 4
        out[v::2] = row
    return out


def decode(data):
    """JPEG file bytes -> uint8 [H][W][3], the array of Image.open(...).convert("RGB")."""
    info = parse(data)
    W, H = info["width"], info["height"]
    planes, hmax, vmax = _planes(info)
    if len(planes) == 1:
        y = planes[0][:H, :W]
        return np.stack([y, y, y], axis=-1)
    y = planes[0][:H, :W].astype(np.int32)
    dw, dh = -(-W // hmax), -(-H // vmax)
    ch = []
    for p in planes[1:]:
        if (hmax, vmax) == (1, 1):
            c = p.astype(np.int32)
        elif (hmax, vmax) == (2, 1):
            c = _h2v1_fancy(p, dw) if dw > 2 else np.repeat(p[:, :dw].astype(np.int32), 2, axis=1)
        else:
            c = _h2v2_fancy(p, dw, dh) if dw > 2 else np.repeat(np.repeat(p[:dh, :dw].astype(np.int32), 2, axis=0), 2, axis=1)
        ch.append(c[:H, :W])
    cb, cr = ch[0] - 128, ch[1] - 128
    r = y + ((91881 * cr + 32768) >> 16)
    g = y + ((-22554 * cb + 32768 - 46802 * cr) >> 16)
    b = y + ((116130 * cb + 32768) >> 16)
    return np.clip(np.stack([r, g, b], axis=-1), 0, 255).astype(np.uint8)
