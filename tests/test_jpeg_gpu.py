"""csrc/jpeg.hip against Pillow itself (the reference's decoder, build-index.py:47) and the committed Pillow pixels:
bit-exact RGB for every file the host parser lets through; corrupt files are reported, not mis-decoded."""
import io

import numpy as np
import pytest
import torch
from PIL import Image

import clipmi
from clipmi import jpeg, jpeg_parse
from test_jpeg import encode, golden_cases, smooth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def pillow(blob):
    return np.asarray(Image.open(io.BytesIO(blob)).convert("RGB"))


def test_golden_files_decode_to_the_committed_pillow_pixels():
    cases = golden_cases()
    got = jpeg.decode_files([b for b, _ in cases], DEV)
    for (blob, rgb), g in zip(cases, got):
        assert g is not None and np.array_equal(g, rgb)


def test_device_decode_equals_pillow_live():
    """One batch with every supported sampling x three qualities x smooth / noise content, odd sizes, an image of several
    chunks of subsequences (1 100 x 1 500), grey and optimised-table files, a 2 x 2-block image."""
    rng = np.random.default_rng(21)
    blobs = []
    for (h, w) in [(224, 224), (37, 53), (8, 8), (17, 16), (100, 75), (64, 129), (5, 7), (480, 640), (1100, 1500)]:
        for sub in (0, 1, 2):
            for q in (95, 75, 30):
                blobs.append(encode(smooth(rng, h, w), quality=q, subsampling=sub))
                if h * w < 100000:
                    blobs.append(encode(rng.integers(0, 256, (h, w, 3), dtype=np.uint8), quality=q, subsampling=sub))
    blobs.append(encode(smooth(rng, 300, 200)[..., 0], quality=85, optimize=True))
    blobs.append(encode(smooth(rng, 300, 200), quality=85, optimize=True))
    blobs.append(encode(np.zeros((16, 16, 3), np.uint8), quality=50))
    for (h, w) in [(64, 96), (37, 53), (480, 640), (224, 224)]:         # restart intervals: 1 MCU, a few, a row of MCUs, many rows
        for sub in (0, 1, 2):
            for kw in (dict(restart_marker_blocks=1), dict(restart_marker_blocks=7), dict(restart_marker_rows=1), dict(restart_marker_rows=3)):
                a = smooth(rng, h, w) if (h + sub) % 2 else rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
                blobs.append(encode(a, quality=88, subsampling=sub, **kw))
                assert jpeg_parse.parse(blobs[-1]).ri > 0
    got = jpeg.decode_files(blobs, DEV)
    raw = jpeg.decode_files(blobs, DEV, keep_stuffing=True)          # the pipeline's form: byte stuffing removed on the device
    assert sum(jpeg_parse.parse(b, keep_stuffing=True).stuffed for b in blobs) >= 140          # (not the restart-interval files)
    for b, g, r in zip(blobs, got, raw):
        assert g is not None and r is not None
        ref = pillow(b)
        assert np.array_equal(g, ref) and np.array_equal(r, ref)


def test_noise_image_of_many_subsequences_converges():
    """A noise image has no end-of-block symbols: the subsequences do not re-synchronise and the decode degenerates to the
    serial chain, one subsequence per round, across several chunks of 256 - same pixels."""
    rng = np.random.default_rng(22)
    blob = encode(rng.integers(0, 256, (400, 600, 3), dtype=np.uint8), quality=98, subsampling=0)
    assert len(jpeg_parse.parse(blob).stream) > 3 * 256 * 128
    (g,) = jpeg.decode_files([blob], DEV)
    assert np.array_equal(g, pillow(blob))


def test_corrupt_entropy_data_is_reported_or_decoded_as_pillow_does():
    rng = np.random.default_rng(23)
    blob = encode(smooth(rng, 96, 128), quality=85)
    p = jpeg_parse.parse(blob)
    short = jpeg_parse.parse(blob)
    short.stream = p.stream[:len(p.stream) // 2]                 # the data ends early
    out, recs, status = jpeg.decode_device([p, short], DEV)
    assert status.cpu().tolist() == [0, 2]
    start = blob.index(b"\xff\xda") + 14
    n_checked = 0
    for k in range(24):
        bad = bytearray(blob)
        pos = start + int(rng.integers(0, len(blob) - start - 4))
        bad[pos] ^= 1 << int(rng.integers(0, 8))
        bad = bytes(bad)
        try:
            q = jpeg_parse.parse(bad)
        except jpeg_parse.Unsupported:
            continue
        (g,) = jpeg.decode_files([bad], DEV)
        if g is None:
            continue                                             # reported: the file goes to Pillow
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            try:
                ref = pillow(bad)
            except Exception:
                continue                                         # Pillow refuses the file; the device decoded what was there
        if np.array_equal(g, ref):
            n_checked += 1
    assert n_checked >= 4


def test_marker_inside_a_stuffed_segment_is_reported():
    """A segment handed over with its stuffing is searched for markers on the device: 0xFF followed by anything but 0x00."""
    rng = np.random.default_rng(26)
    blob = encode(smooth(rng, 120, 160), quality=85)
    p = jpeg_parse.parse(blob, keep_stuffing=True)
    assert p.stuffed == 1 and p.stream.count(b"\xff\x00") > 0
    q = jpeg_parse.parse(blob, keep_stuffing=True)
    k = len(q.stream) // 2
    while q.stream[k - 1] == 0xFF or q.stream[k] == 0xFF:
        k += 1
    q.stream = q.stream[:k] + b"\xff\xd3" + q.stream[k + 2:]
    r = jpeg_parse.parse(blob, keep_stuffing=True)
    r.stream = r.stream[:-1] + b"\xff"                                  # a 0xFF with nothing behind it
    out, recs, status = jpeg.decode_device([p, q, r], DEV)
    assert status.cpu().tolist() == [0, 3, 3]
    (g,) = jpeg.decode_files([blob], DEV, keep_stuffing=True)
    assert np.array_equal(g, pillow(blob))


def test_restart_interval_that_ends_early_is_reported():
    rng = np.random.default_rng(25)
    blob = encode(smooth(rng, 96, 128), quality=85, restart_marker_rows=1)
    p = jpeg_parse.parse(blob)
    q = jpeg_parse.parse(blob)
    cut = int(q.starts[2])
    q.stream = q.stream[:cut - 40] + q.stream[cut:]                 # the second interval loses its last 40 bytes
    q.starts = np.concatenate([q.starts[:2], q.starts[2:] - 40]).astype(np.uint32)
    out, recs, status = jpeg.decode_device([p, q], DEV)
    st = status.cpu().tolist()
    assert st[0] == 0 and st[1] != 0


def test_decoded_pixels_feed_the_resize_kernel():
    """The decoder's output is the layout clipmi_resize_crop_rgb8 takes: decode + resize on the device equals the host
    transform's pixels (decode_worker.load_uint8 = Pillow decode + Pillow resize + crop)."""
    import os
    import tempfile
    from clipmi import resize
    from clipmi.decode_worker import load_uint8
    rng = np.random.default_rng(24)
    blobs = [encode(smooth(rng, h, w), quality=90, subsampling=s) for (h, w, s) in [(300, 400, 2), (224, 224, 2), (250, 224, 1), (500, 333, 0)]]
    imgs = jpeg.decode_files(blobs, DEV)
    got = resize.resize_crop_device(imgs, 224, DEV).cpu().numpy()
    with tempfile.TemporaryDirectory() as d:
        for k, b in enumerate(blobs):
            path = os.path.join(d, f"{k}.jpg")
            with open(path, "wb") as f:
                f.write(b)
            assert np.array_equal(got[k], load_uint8(path, 224))
