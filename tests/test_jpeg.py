"""The JPEG decode in front of the transform (reference build-index.py:47): the CPU restatement
(oracle/jpeg_oracle.py) against Pillow itself and against the committed Pillow outputs, and the host parser."""
import io
import os

import numpy as np
import pytest
from PIL import Image

import clipmi
from clipmi import jpeg_parse
from oracle import jpeg_oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "jpeg_cases.npz")


def golden_cases():
    d = np.load(GOLDEN)
    return [(d[f"file_{i}"].tobytes(), d[f"rgb_{i}"]) for i in range(int(d["n"]))]


def smooth(rng, h, w):
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([127 + 100 * np.sin(xx / 9.0 + yy / 17.0), 127 + 100 * np.cos(xx / 13.0 - yy / 7.0), (xx * 3 + yy * 2) % 256], -1)
    return np.clip(base + rng.normal(0, 12, (h, w, 3)), 0, 255).astype(np.uint8)


def encode(a, **kw):
    buf = io.BytesIO()
    Image.fromarray(a).save(buf, format="JPEG", **kw)
    return buf.getvalue()


def test_oracle_equals_the_committed_pillow_pixels():
    for blob, rgb in golden_cases():
        assert np.array_equal(jpeg_oracle.decode(blob), rgb)


def test_oracle_equals_pillow_live():
    """Pillow is the reference's own decoder and is installed here: every supported sampling, odd sizes, three qualities,
    noise (no end-of-block symbols) and smooth content, grey, optimised tables."""
    rng = np.random.default_rng(11)
    n = 0
    for (h, w) in [(224, 224), (31, 45), (9, 8), (120, 77)]:
        for sub in (0, 1, 2):
            for q in (95, 60):
                for a in (smooth(rng, h, w), rng.integers(0, 256, (h, w, 3), dtype=np.uint8)):
                    blob = encode(a, quality=q, subsampling=sub)
                    assert np.array_equal(jpeg_oracle.decode(blob), np.asarray(Image.open(io.BytesIO(blob)).convert("RGB")))
                    n += 1
    blob = encode(smooth(rng, 50, 70)[..., 1], quality=80, optimize=True)
    assert np.array_equal(jpeg_oracle.decode(blob), np.asarray(Image.open(io.BytesIO(blob)).convert("RGB")))
    assert n == 48
    for sub in (0, 1, 2):                                  # restart intervals (DRI + RSTn markers)
        for kw in (dict(restart_marker_blocks=1), dict(restart_marker_blocks=5), dict(restart_marker_rows=1)):
            blob = encode(smooth(rng, 70, 100), quality=85, subsampling=sub, **kw)
            assert b"\xff\xdd" in blob
            assert np.array_equal(jpeg_oracle.decode(blob), np.asarray(Image.open(io.BytesIO(blob)).convert("RGB")))


def test_parser_reads_what_pillow_reads():
    rng = np.random.default_rng(3)
    for sub, (hs, vs) in ((0, (1, 1)), (1, (2, 1)), (2, (2, 2))):
        blob = encode(smooth(rng, 41, 67), quality=88, subsampling=sub)
        p = jpeg_parse.parse(blob)
        im = Image.open(io.BytesIO(blob))
        assert (p.width, p.height, p.ncomp, p.hs, p.vs) == (im.size[0], im.size[1], 3, hs, vs)
        # quantisation steps: Pillow (>= 8.3) reports them de-zigzagged, the order the parser hands the device
        for c, tq in enumerate((0, 1, 1)):
            assert p.quant[c].tolist() == list(im.quantization[tq])
        assert len(p.tables) == 6 and all(len(t) == jpeg_parse.TABLE_BYTES for t in p.tables)
        assert p.blocks() == -(-67 // (8 * hs)) * -(-41 // (8 * vs)) * (hs * vs + 2)
    blob = encode(smooth(rng, 60, 90), quality=90)
    a, b = jpeg_parse.parse(blob), jpeg_parse.parse(blob, keep_stuffing=True)
    assert (a.stuffed, b.stuffed) == (0, 1) and b.stream.replace(b"\xff\x00", b"\xff") == a.stream
    assert jpeg_parse.parse(blob + b"trailing bytes", keep_stuffing=True).stuffed == 0      # no EOI at the end: the host looks for it
    g = jpeg_parse.parse(encode(smooth(rng, 20, 30)[..., 0], quality=70))
    assert (g.ncomp, g.hs, g.vs, g.blocks()) == (1, 1, 1, 3 * 4)


def test_parser_leaves_everything_else_to_pillow():
    rng = np.random.default_rng(4)
    a = smooth(rng, 64, 64)
    with pytest.raises(jpeg_parse.Unsupported):
        jpeg_parse.parse(encode(a, quality=80, progressive=True))
    buf = io.BytesIO()
    Image.fromarray(a).convert("CMYK").save(buf, format="JPEG")
    with pytest.raises(jpeg_parse.Unsupported):
        jpeg_parse.parse(buf.getvalue())
    buf = io.BytesIO()
    Image.fromarray(a).save(buf, format="PNG")
    with pytest.raises(jpeg_parse.Unsupported):
        jpeg_parse.parse(buf.getvalue())
    blob = encode(a, quality=80)
    with pytest.raises(jpeg_parse.Unsupported):
        jpeg_parse.parse(blob[:len(blob) // 2])               # no end-of-image marker: Pillow's error handling applies
    with pytest.raises(jpeg_parse.Unsupported):
        jpeg_parse.parse(encode(a[:, :4], quality=80, subsampling=2))      # 2 chroma columns: libjpeg upsamples without the filter
    with pytest.raises(jpeg_parse.Unsupported):
        jpeg_parse.parse(b"")
    blob = encode(a, quality=80, restart_marker_blocks=4)            # restart intervals are taken ...
    p = jpeg_parse.parse(blob)
    assert p.ri == 4 and len(p.starts) == -(-p.mcus() // 4) and p.starts[0] == 0 and b"\xff\xd0" not in p.stream[:int(p.starts[1]) + 2]
    k = blob.index(b"\xff\xd1")
    with pytest.raises(jpeg_parse.Unsupported):                       # ... unless their markers are out of turn
        jpeg_parse.parse(blob[:k + 1] + b"\xd3" + blob[k + 2:])


def test_pack_lays_out_aligned_streams_and_shares_tables():
    from clipmi import jpeg
    rng = np.random.default_rng(6)
    items = [jpeg_parse.parse(encode(smooth(rng, 30 + 7 * k, 50 + k), quality=90, subsampling=k % 3)) for k in range(5)]
    recs, tables, streams, out_bytes, total_blocks, max_blocks, max_pixels = jpeg.pack(items)
    assert tables.shape == (4, jpeg_parse.TABLE_BYTES)        # Pillow's standard tables: luma / chroma x DC / AC
    assert total_blocks == sum(it.blocks() for it in items) and max_blocks == max(it.blocks() for it in items)
    for r, it in zip(recs, items):
        o, n = int(r["stream_off"]), int(r["stream_bytes"])
        assert o % 16 == 0 and streams[o:o + n].tobytes() == it.stream and not streams[o + n:o + n + 16].any()
        assert int(r["out_off"]) % 16 == 0
    assert out_bytes >= sum(it.width * it.height * 3 for it in items)


def test_worker_regions_and_the_records_built_from_them(tmp_path):
    """What a decode worker lays out for a JPEG file (decode_worker.stage_jpeg) and what the parent builds out of a batch's
    regions (pipeline.jpeg_records) is, field for field, what jpeg.pack builds from the parsed files: sizes, sampling,
    quantisation steps, table indices into the distinct tables, the segment's bytes, restart intervals, the resize plan."""
    from clipmi import jpeg, pipeline
    from clipmi import decode_worker as dw
    rng = np.random.default_rng(12)
    specs = [(224, 224, dict(quality=95, subsampling=2)), (300, 500, dict(quality=85, subsampling=1)),
             (120, 90, dict(quality=70, subsampling=0, optimize=True)), (260, 340, dict(quality=88, restart_marker_rows=1)),
             (96, 64, dict(quality=80))]
    n, cap, n_px = len(specs) + 2, 256 << 10, 224
    big = np.zeros(n * cap, np.uint8)
    slots, items = [], []
    for k, (h, w, kw) in enumerate(specs):
        a = smooth(rng, h, w) if k != 4 else smooth(rng, h, w)[..., 0]
        path = str(tmp_path / f"f{k}.jpg")
        Image.fromarray(a).save(path, format="JPEG", **kw)
        slot = k + (1 if k >= 2 else 0)                       # slot 2 holds no JPEG file (a PNG, say): a gap in the batch
        got = dw.stage_jpeg(path, n_px, big[slot * cap:(slot + 1) * cap])
        assert got[:2] == (w, h) and 0 < got[2] <= cap
        slots.append(slot)
        items.append(jpeg_parse.parse(open(path, "rb").read(), keep_stuffing=True))
    assert dw.stage_jpeg(str(tmp_path / "f1.jpg"), n_px, big[6 * cap:6 * cap + 4096])[2] < 0          # too small a region: says what it needs
    with pytest.raises(jpeg_parse.Unsupported):
        Image.fromarray(smooth(rng, 50, 50)).save(str(tmp_path / "p.jpg"), format="JPEG", progressive=True)
        dw.stage_jpeg(str(tmp_path / "p.jpg"), n_px, big[6 * cap:7 * cap])
    comp = np.arange(n)
    recs, tables, jobs, out_sz, blocks, nt = pipeline.jpeg_records(big, n, cap, slots, comp, n_px)
    ref, rtab, rstreams, _, total_blocks, _, _ = jpeg.pack(items)
    assert nt == len(rtab) and int(blocks.sum()) == total_blocks
    tables = tables.reshape(nt, jpeg_parse.TABLE_BYTES)
    for k, (r, q, it, slot) in enumerate(zip(recs, ref, items, slots)):
        for f in ("stream_bytes", "width", "height", "ncomp", "hs", "vs", "restart_interval", "n_intervals", "stuffed", "coef_off", "out_off"):
            assert int(r[f]) == int(q[f]), f
        assert np.array_equal(r["quant"], q["quant"])
        for c in range(3):                                  # the same tables, whatever their numbering
            assert tables[r["dc_tbl"][c]].tobytes() == rtab[q["dc_tbl"][c]].tobytes()
            assert tables[r["ac_tbl"][c]].tobytes() == rtab[q["ac_tbl"][c]].tobytes()
        o, nb = int(r["stream_off"]), int(r["stream_bytes"])
        assert slot * cap <= o and o + nb + 16 <= (slot + 1) * cap and o % 16 == 0
        assert big[o:o + nb].tobytes() == it.stream and not big[o + nb:o + nb + 16].any()
        if it.ri:
            io_ = int(r["intervals_off"])
            assert np.array_equal(np.frombuffer(big, np.uint32, count=len(it.starts), offset=io_), it.starts)
        plan = dw.resize_plan(it.width, it.height, n_px)
        j = jobs[k]
        assert (int(j["w"]), int(j["h"]), int(j["r0"]), int(j["nrows"]), int(j["need_h"]), int(j["need_v"]), int(j["left"]), int(j["top"]),
                int(j["hk"]), int(j["vk"])) == (it.width, it.height, plan["r0"], plan["nrows"], plan["need_h"], plan["need_v"], plan["left"],
                                                plan["top"], plan["hk"], plan["vk"])
        assert int(j["src_off"]) == int(r["out_off"]) and int(j["out_index"]) == slot
        hc = np.frombuffer(big, np.int32, count=plan["hcoef"].size, offset=4 * int(j["hcoef_off"]))
        vc = np.frombuffer(big, np.int32, count=plan["vcoef"].size, offset=4 * int(j["vcoef_off"]))
        assert np.array_equal(hc, plan["hcoef"]) and np.array_equal(vc, plan["vcoef"])
    assert np.array_equal(jobs["tmp_off"], np.cumsum(jobs["nrows"].astype(np.int64) * n_px * 3) - jobs["nrows"].astype(np.int64) * n_px * 3)
