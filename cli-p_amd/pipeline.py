"""Host image pipeline for the build side: threaded decode + the CLIP
transform's geometry on the host, pinned staging, asynchronous H2D, encode on the GPU in batches.

Reference per-image sequence (build-index.py:47-51): Image.open -> transform -> unsqueeze(0).to(device)
-> encode_image -> /norm -> .cpu(). Here the same images travel as uint8 [B,3,R,R] (resize / centre
crop / RGB done with Pillow exactly as `transform` does; the /255, -mean, /std tail is fused into the
device patch kernel), B at a time, decode of batch i+1 overlapping the GPU work of batch i.
Failures are per file (build-index.py:55-58): a file that does not decode is reported, not fatal.
"""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch


from .decode_worker import load_uint8  # noqa: E402,F401  (the Pillow part of the transform lives beside its worker script)


class DecodePool:
    """Worker PROCESSES for the Pillow part of the transform (decode_worker.py): decode throughput that scales with
    the host cores also for small images, where threads are bound by the GIL. The workers are plain `python
    decode_worker.py` children started HERE - create the pool BEFORE the process initialises the GPU (indexer.main
    does): a process that has touched the GPU should not spawn programs on this platform. Pixels come back through
    shared memory (workers write their slots; no pickling, no copies through pipes): two segments used in turn, so that
    a batch can be decoded while the previous one is still being copied out."""

    def __init__(self, workers):
        import os
        import subprocess
        import sys
        self.n = max(1, int(workers))
        script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "decode_worker.py")
        # own session: a Ctrl-C at the terminal reaches the ranks (which finish their round: indexer.StopFlag), not the
        # workers - a worker that exited on SIGINT used to turn its current file into a "failed" file that was then
        # written to skip_db and never retried. The workers also ignore SIGINT themselves and leave at EOF on stdin.
        import signal
        import threading
        restore = None
        if threading.current_thread() is threading.main_thread():
            restore = signal.signal(signal.SIGINT, signal.SIG_IGN)      # inherited across exec: ignored from the first instruction
        try:
            self.procs = [subprocess.Popen([sys.executable, script], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                           start_new_session=True)
                          for _ in range(self.n)]
        finally:
            if restore is not None:
                signal.signal(signal.SIGINT, restore)
        self.lost = set()           # files a worker DIED on: failed for this run, but not proven undecodable (never skip_db)
        self._all_procs = list(self.procs)
        self.threads = ThreadPoolExecutor(max_workers=self.n)
        self.segs = [None, None, None, None]      # 0, 1: the n_px x n_px slots of two batches in turn; 2, 3: their full-size regions
        self.jpeg_wanted = 0                      # largest region a JPEG file asked for and did not get (encode_files sizes by it)
        self.jpeg_cap_hint = 0                    # the region size the last encode_files call ended with

    def _segment(self, nbytes, which=0):
        from multiprocessing import shared_memory
        seg = self.segs[which]
        if seg is None or seg.size < nbytes:
            self._drop_segment(which)
            seg = self.segs[which] = shared_memory.SharedMemory(create=True, size=int(nbytes))
        return seg

    @staticmethod
    def shm_room():
        """Bytes of /dev/shm THIS process may count on for its segments (/dev/shm is a tmpfs whose pages exist only once
        written: a segment larger than what is free is created without complaint and kills the writer later; containers
        default to 64 MB). The ranks of one node all see the same free space before any of them has written a page, so
        each takes its share: free / LOCAL_WORLD_SIZE."""
        import os
        try:
            st = os.statvfs("/dev/shm")
            return st.f_bavail * st.f_frsize // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
        except (OSError, ValueError):
            return 0

    def pin_segment(self, which):
        """Page-lock segment `which` for the GPU (hipHostRegister through torch's runtime handle) so that it can be copied
        to the device where it lies - no packing copy in this process. True when it is (already) locked."""
        seg = self.segs[which]
        if seg is None:
            return False
        reg = self.__dict__.setdefault("_pinned", {})
        if reg.get(which) == seg.name:
            return True
        try:
            import ctypes
            addr = ctypes.addressof(ctypes.c_char.from_buffer(seg.buf))
            rc = torch.cuda.cudart().cudaHostRegister(addr, seg.size, 0)
            ok = int(rc) == 0
        except Exception:
            ok = False
        if ok:
            reg[which] = seg.name
            self.__dict__.setdefault("_pinned_addr", {})[which] = addr
        return ok

    def _unpin_segment(self, which):
        reg = self.__dict__.get("_pinned", {})
        if which in reg:
            try:
                torch.cuda.cudart().cudaHostUnregister(self._pinned_addr[which])
            except Exception:
                pass
            reg.pop(which, None)

    def _drop_segment(self, which):
        """Unlink first (always possible), then unmap (refused while a caller still holds a view: the mapping then goes
        with the last reference)."""
        seg = self.segs[which]
        if seg is None:
            return
        self._unpin_segment(which)
        try:
            seg.unlink()
        except Exception:
            pass
        try:
            seg.close()
        except BufferError:
            seg.close = lambda: None          # a view is still alive: the mapping goes with it; keep __del__ quiet
        self.segs[which] = None

    def _run(self, w, jobs, n_px, name, seg, big=None):
        """Worker w decodes its share of the batch: (slot, path) pairs -> (slot, status) with status False (failed), True
        (the transform's pixels are in the slot) or (kind, w, h, bytes): the image is in its region of the big segment -
        kind 2 at full size, for the resize on the device; kind 3 as a parsed JPEG file, for the decode on the device. If the worker process dies (a file that crashes the decoder, an OOM kill), that
        file is reported as failed and the rest of the share - and of every later batch - is decoded in this process: no
        program is spawned once the GPU may have been initialised."""
        import struct
        p = self.procs[w]
        ok = []
        per = 3 * n_px * n_px
        bname, bcap, bmode = (big[0], big[1], big[2]) if big else (b"-", 0, 0)
        if p is not None:
            # the whole share in one write: this thread sleeps in read() while the worker decodes (one request per round
            # trip kept 16 parent threads busy handing the GIL around)
            # the path travels hex-encoded: a file name may hold '\n' or '\t' (legal on Linux, and os.listdir returns them) -
            # raw, such a name split into two request lines and shifted every later reply of the worker by one
            req = b"".join(b"%d\t%s\t%d\t%s\t%d\t%d\t%d\t" % (n_px, name, slot * per, bname, slot * bcap, bcap, bmode) +
                           path.encode("utf-8", "surrogateescape").hex().encode() + b"\n" for slot, path in jobs)
            try:
                p.stdin.write(req)
                p.stdin.flush()
                # one read for the whole share: the worker answers every file with a 17-byte record (status + <iiq), and a
                # short read means it died - the number of whole records says on which file (a read per file kept the 16
                # parent threads handing the GIL around: 5 ms per 435-image batch)
                raw = p.stdout.read(17 * len(jobs))
                for k in range(len(raw) // 17):
                    st = raw[17 * k:17 * k + 1]
                    if st == b"2" or st == b"3":
                        ok.append((jobs[k][0], (int(st),) + struct.unpack_from("<iiq", raw, 17 * k + 1)))
                    elif st == b"5":                       # Pillow decoded it; a larger region would have taken the file itself
                        self.jpeg_wanted = max(self.jpeg_wanted, struct.unpack_from("<iiq", raw, 17 * k + 1)[2])
                        ok.append((jobs[k][0], True))
                    else:
                        ok.append((jobs[k][0], st == b"1"))
            except (BrokenPipeError, OSError):
                pass
            if len(ok) < len(jobs):                # the worker died on file len(ok): that one failed, the rest in-process
                self.procs[w] = None
                self.lost.add(jobs[len(ok)][1])    # ... for this run only: the caller must not record it as undecodable
                ok.append((jobs[len(ok)][0], False))
        for slot, path in jobs[len(ok):]:
            try:
                load_uint8(path, n_px, out=np.frombuffer(seg.buf, dtype=np.uint8, count=per, offset=slot * per).reshape(3, n_px, n_px))
                ok.append((slot, True))
            except KeyboardInterrupt:
                raise
            except Exception:
                ok.append((slot, False))
        return ok

    def decode(self, paths, n_px, copy=True, segment=0, full_cap=0, full_mode=1):
        """-> (uint8 array [n_ok,3,n_px,n_px], ok_paths, failed_paths), file order kept. copy=False returns a VIEW of
        the pool's shared-memory segment `segment` (all slots, plus a boolean mask of the good ones instead of the
        compacted array): valid until the next decode() into the same segment - encode_files copies it straight into
        pinned memory. One decode() at a time (the workers take one request stream).
        full_cap > 0 (with copy=False): 8-bit RGB images that need resampling and fit full_cap bytes are delivered at FULL
        size in a second segment, one region of full_cap bytes per slot (tmpfs pages exist only where written), for the
        resize on the device; the result is then ((slots view, good mask, big view, {slot: (kind, w, h, bytes)}), ok, bad).
        full_mode: what a region may take - bit 0 full-size pixels (kind 2), bit 1 baseline JPEG files parsed for the decode on
        the device (kind 3: decode_worker.stage_jpeg)."""
        n = len(paths)
        per = 3 * n_px * n_px
        seg = self._segment(max(1, n * per), segment)
        name = seg.name.encode()
        big = bseg = None
        if full_cap > 0 and not copy:
            full_cap = (int(full_cap) + 15) // 16 * 16
            bseg = self._segment(max(1, n * full_cap), 2 + segment)
            big = (bseg.name.encode(), full_cap, int(full_mode))
        live = [w for w in range(self.n) if self.procs[w] is not None] or [0]
        futs = [self.threads.submit(self._run, w, [(i, paths[i]) for i in range(k, n, len(live))], n_px, name, seg, big)
                for k, w in enumerate(live) if k < n]
        good = np.zeros(n, dtype=bool)
        full = {}
        for f in futs:
            for slot, st in f.result():
                good[slot] = bool(st)
                if isinstance(st, tuple):
                    full[slot] = st
        arr = np.frombuffer(seg.buf, dtype=np.uint8, count=n * per).reshape(n, 3, n_px, n_px)
        ok = [p for p, g_ in zip(paths, good) if g_]
        bad = [p for p, g_ in zip(paths, good) if not g_]
        if not copy:
            if big is not None:
                return (arr, good, np.frombuffer(bseg.buf, dtype=np.uint8, count=n * full_cap), full), ok, bad
            return (arr, good), ok, bad
        out = arr[good].copy() if not good.all() else arr.copy()
        return out, ok, bad

    def close(self):
        procs = [p for p in getattr(self, "_all_procs", self.procs) if p is not None]
        for p in procs:
            try:
                p.stdin.close()
            except Exception:
                pass
        for p in procs:
            try:
                p.wait(timeout=5)
            except Exception:
                p.kill()
        self.threads.shutdown(wait=False)
        for which in range(4):
            self._drop_segment(which)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def _load_safe(args):
    path, n_px = args
    try:
        return load_uint8(path, n_px)
    except KeyboardInterrupt:
        raise
    except Exception:
        return None


def jpeg_records(bigview, n, cap, slots, comp, n_px):
    """The device decoder's records out of a batch's regions of the big segment (decode_worker.stage_jpeg wrote them): every field
    comes out of the regions' headers as one strided numpy gather - no Python per image except the table-set look-up.
    bigview: the segment (uint8, n regions of cap bytes); slots: the regions that hold a parsed JPEG file; comp: slot -> row of the
    batch's tensor. -> (clipmi_jpeg_image records with offsets into the segment, distinct raw Huffman tables uint8, clipmi_resize_job
    records whose sources are the decoder's outputs laid out back to back, output bytes per image, blocks per image, number of tables)"""
    from . import jpeg as J
    from .decode_worker import JPEG_COEF_OFF, JPEG_HDR_INTS, JPEG_QUANT_OFF, JPEG_TABLES_OFF
    from .resize import JOB
    slots = np.asarray(slots, dtype=np.int64)
    n3 = len(slots)
    st = np.lib.stride_tricks.as_strided
    H = st(bigview[:4 * JPEG_HDR_INTS].view(np.int32), shape=(n, JPEG_HDR_INTS), strides=(cap, 4))[slots].astype(np.int64)
    w, h, blocks, nrows = H[:, 1], H[:, 2], H[:, 7], H[:, 9]
    recs = np.zeros(n3, dtype=J.IMAGE)
    out_sz = (w * h * 3 + 15) // 16 * 16
    out_off = np.cumsum(out_sz) - out_sz
    recs["stream_off"], recs["coef_off"], recs["out_off"] = slots * cap + H[:, 18], np.cumsum(blocks) - blocks, out_off
    recs["stream_bytes"], recs["width"], recs["height"] = H[:, 6], w, h
    recs["ncomp"], recs["hs"], recs["vs"] = H[:, 3], H[:, 4], H[:, 5]
    recs["restart_interval"], recs["n_intervals"], recs["intervals_off"] = H[:, 20], H[:, 21], slots * cap + H[:, 22]
    recs["stuffed"] = H[:, 23]
    recs["quant"] = st(bigview[JPEG_QUANT_OFF:], shape=(n, 192), strides=(cap, 1))[slots].reshape(n3, 3, 64)
    # the Huffman tables: distinct six-table sets first (files of one encoder share theirs), then distinct tables
    tabs = st(bigview[JPEG_TABLES_OFF:], shape=(n, 6 * J.TABLE_BYTES), strides=(cap, 1))[slots]
    sets, pool_t, set_idx = {}, {}, np.zeros((n3, 6), np.int32)
    for k in range(n3):
        key = tabs[k].tobytes()
        idx = sets.get(key)
        if idx is None:
            idx = sets[key] = [pool_t.setdefault(key[t * J.TABLE_BYTES:(t + 1) * J.TABLE_BYTES], len(pool_t)) for t in range(6)]
        set_idx[k] = idx
    recs["dc_tbl"], recs["ac_tbl"] = set_idx[:, 0::2], set_idx[:, 1::2]
    tables = np.frombuffer(b"".join(pool_t), np.uint8)
    jobs = np.zeros(n3, dtype=JOB)
    jobs["src_off"], jobs["w"], jobs["h"], jobs["r0"], jobs["nrows"], jobs["out_index"] = out_off, w, h, H[:, 8], nrows, np.asarray(comp)[slots]
    jobs["need_h"], jobs["need_v"], jobs["left"], jobs["top"], jobs["hk"], jobs["vk"] = (H[:, 10], H[:, 11], H[:, 12], H[:, 13],
                                                                                      H[:, 14], H[:, 15])
    jobs["hcoef_off"] = (slots * cap + JPEG_COEF_OFF) // 4
    jobs["vcoef_off"] = jobs["hcoef_off"] + H[:, 16]
    tmp = nrows * n_px * 3
    jobs["tmp_off"] = np.cumsum(tmp) - tmp
    return recs, tables, jobs, out_sz, blocks, len(pool_t)


def encode_files(model, paths, batch=256, workers=8, pool=None, device_resize_mb=None, device_jpeg_kb=None, stats=None,
                 jpeg_group_mb=32768):
    """Generator over batches: yields (ok_paths, features f32 [n,E] numpy normalised, failed_paths).
    Decode runs in the worker processes of `pool` (a DecodePool) when given, else on `workers` threads (Pillow
    releases the GIL while decoding, which is enough for large photos and not for small images).
    device_resize_mb (default $CLIPMI_DEVICE_RESIZE_MB, 0 = off; needs `pool` and a GPU): 8-bit RGB images of up to that
    many MB decoded travel at full size and are resized + cropped by clipmi_resize_crop_rgb8 - the same pixels, with the
    workers left to decode only (Pillow's bicubic resize is half of a photo-sized file's host time).
    device_jpeg_kb (default $CLIPMI_DEVICE_JPEG_KB, else 8192; 0 = off; needs `pool` and a GPU): baseline JPEG files of up to that
    many KB are not decoded on the host at all - a worker reads the file, walks its markers and removes the byte stuffing
    (jpeg_parse.py), and clipmi_jpeg_decode_rgb8 + clipmi_resize_crop_rgb8 produce the transform's pixels in HBM, the same
    bytes as Pillow's. Every other file (progressive, PNG, CMYK ...) and every file the device reports corrupt takes the
    Pillow path as before.
    jpeg_group_mb: the device decodes a batch's JPEG files in groups whose decoded form (~22 bytes per pixel) stays under that
    many MB of HBM - one group for a batch of thumbnails, several for a batch of photos.
    stats: a dict that receives the seconds each of the three pipelined stages was busy (decode_s: worker processes, copy_s:
    shared memory -> device incl. the decode / resize kernels, encode_s) and the files that took the device decoder (jpeg_files)."""
    import os
    import time
    n_px = model.visual.input_resolution
    dev = model.device
    use_gpu = dev.type == "cuda"
    copy_stream = None
    if use_gpu:
        from . import _lib as _l
        copy_stream = _l.copy_stream(dev)
    if device_resize_mb is None:
        device_resize_mb = float(os.environ.get("CLIPMI_DEVICE_RESIZE_MB", "0"))
    if pool is not None and pool.shm_room() < 2 * batch * 3 * n_px * n_px + (64 << 20):
        print(f"(shared memory too small for decode workers' batches: {pool.shm_room() >> 20} MB free in /dev/shm; "
              f"decoding on {workers} threads)")
        pool = None
    if device_jpeg_kb is None:
        device_jpeg_kb = float(os.environ.get("CLIPMI_DEVICE_JPEG_KB", "8192"))
    on_device = use_gpu and pool is not None
    resize_cap = int(device_resize_mb * (1 << 20)) if on_device else 0
    jpeg_cap = int(device_jpeg_kb * 1024) if on_device else 0
    jpeg_group_bytes = int(jpeg_group_mb) << 20

    def room_for(cap):
        return pool.shm_room() >= 2 * batch * (3 * n_px * n_px + cap) + (256 << 20)

    if resize_cap and not room_for(resize_cap):
        resize_cap = 0
    if jpeg_cap:
        # the largest region /dev/shm has room for (two segments of `batch` regions beside the n_px slots), found once per pool:
        # regions start small and grow with the files, so a large configured size costs nothing until files of that size come
        if not getattr(pool, "jpeg_fit_cap", 0):
            room = pool.shm_room() - 2 * batch * 3 * n_px * n_px - (256 << 20)
            pool.jpeg_fit_cap = max(1, room // (2 * batch) // 65536 * 65536)
        jpeg_cap = min(jpeg_cap, pool.jpeg_fit_cap)
        if jpeg_cap < (64 << 10):
            jpeg_cap = 0
    # regions are copied to the device whole, so JPEG regions start small (or where the pool's last call ended) and follow the
    # files: a file that does not fit is decoded by Pillow this once and says what it would have needed (DecodePool.jpeg_wanted)
    jpeg_now = min(jpeg_cap, pool.jpeg_cap_hint or (128 << 10)) if jpeg_cap else 0
    full_cap = [max(resize_cap, jpeg_now)]               # bytes per region of the big segment; [0]: mutable (may be switched off)
    full_mode = (1 if resize_cap else 0) | (2 if jpeg_cap else 0)

    # three pinned staging buffers used in turn (GPU): batch i may still be in its H2D copy while batch i+1 is filled;
    # a buffer is reused only after the copy that read it has finished. Pixels go shared memory -> pinned -> device:
    # ONE host copy (the first version copied out of the segment, then again into freshly pinned memory: 55 ms per
    # 435-image batch against 23 ms of decode on 16 workers).
    ring = []

    def staging(n):
        slot = ring.pop(0) if len(ring) >= 3 else {"buf": None, "ev": None}
        if slot["ev"] is not None:
            slot["ev"].synchronize()
        if slot["buf"] is None or slot["buf"].shape[0] < n:
            slot["buf"] = torch.empty((max(n, batch), 3, n_px, n_px), dtype=torch.uint8).pin_memory()
            slot["np"] = slot["buf"].numpy()              # the same pinned bytes as a numpy array
        ring.append(slot)
        return slot

    big_ring = []

    def big_staging(nbytes):
        slot = big_ring.pop(0) if len(big_ring) >= 3 else {"buf": None, "ev": None}
        if slot["ev"] is not None:
            slot["ev"].synchronize()
        if slot["buf"] is None or slot["buf"].numel() < nbytes:
            slot["buf"] = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8).pin_memory()
            slot["np"] = slot["buf"].numpy()
        big_ring.append(slot)
        return slot

    def device_stage(devt, bigview, full, good):
        """The batch's regions of the big segment -> their rows of devt, on the copy stream behind devt's own copy: ONE H2D copy
        of the segment where it lies (it is page-locked: no packing copy on the host - packing 1 GB per batch of photo-sized
        images with one thread was slower than Pillow's resize), then clipmi_resize_crop_rgb8 for the full-size images (kind 2)
        and clipmi_jpeg_decode_rgb8 + clipmi_resize_crop_rgb8 for the parsed JPEG files (kind 3).
        The copy stream carries the copies only; the kernels go to the process's ONE side stream (_lib.side_stream: this ROCm gives a
        process three hardware queues) behind an event, so that the next batch's copy runs beside this batch's kernels instead of
        behind them, and the consumer finds them queued in front of its encode step.
        -> (event behind the kernels, pending: what the consumer checks afterwards - the decoder's per-file status). Returns when the
        segment has been copied (it is decoded into again two batches later)."""
        from . import _lib
        from . import jpeg as J
        from .decode_worker import JPEG_COEF_OFF, JPEG_HDR_INTS, JPEG_QUANT_OFF, JPEG_TABLES_OFF, PLAN_INTS
        from .resize import JOB
        L = _lib.lib()
        n = len(good)
        comp = np.cumsum(good) - 1                           # slot -> row of devt
        cap = bigview.size // n
        used = (max(full) + 1) * cap
        e2 = sorted((s_, v) for s_, v in full.items() if v[0] == 2)
        e3 = np.array(sorted(s_ for s_, v in full.items() if v[0] == 3), dtype=np.int64)
        if not pool.pin_segment(2 + seg_index[0]):
            # the segment could not be page-locked (locked-memory limit?): this batch goes through a pinned copy of it, the
            # following ones take the host path
            full_cap[0] = 0
            slot = big_staging(used)
            np.copyto(slot["np"][:used], bigview[:used])
            src = slot["buf"][:used]
        else:
            slot = None
            src = torch.from_numpy(bigview[:used])
        pending = {"keep": [], "status": None, "launch": []}
        with torch.cuda.stream(copy_stream):
            dbig = src.to(dev, non_blocking=True)
            base = dbig.data_ptr()
            if e2:
                jobs = np.zeros(len(e2), dtype=JOB)
                toff, max_rows = 0, 1
                for t, (s_, (_, w, h, nb)) in enumerate(e2):
                    base_ = s_ * cap
                    o_hdr = (w * h * 3 + 15) // 16 * 16
                    hd = np.frombuffer(bigview, dtype=np.int32, count=PLAN_INTS, offset=base_ + o_hdr)
                    j = jobs[t]
                    j["src_off"], j["w"], j["h"], j["r0"], j["nrows"], j["out_index"] = base_, w, h, hd[2], hd[3], comp[s_]
                    j["need_h"], j["need_v"], j["left"], j["top"], j["hk"], j["vk"] = hd[4], hd[5], hd[6], hd[7], hd[8], hd[9]
                    j["hcoef_off"] = (base_ + o_hdr) // 4 + PLAN_INTS
                    j["vcoef_off"] = j["hcoef_off"] + hd[10]
                    j["tmp_off"] = toff
                    toff += int(hd[3]) * n_px * 3
                    max_rows = max(max_rows, int(hd[3]))
                djobs = torch.from_numpy(jobs.view(np.uint8).reshape(-1).copy()).to(dev)
                scratch = torch.empty(max(toff, 1), dtype=torch.uint8, device=dev)
                pending["keep"] += [djobs, scratch]

                def resize_full(djobs=djobs, scratch=scratch, n2=len(e2), max_rows=max_rows):
                    rc = L.clipmi_resize_crop_rgb8(base, djobs.data_ptr(), n2, max_rows, base, n_px, devt.data_ptr(),
                                                   scratch.data_ptr(), _lib.stream_ptr(dev))
                    _lib.check(rc, "clipmi_resize_crop_rgb8")

                pending["launch"].append(resize_full)
            if len(e3):
                # groups of files whose decoded form (coefficients, sample planes, RGB rows: ~22 bytes per pixel) fits a budget: a
                # batch of thumbnails is one group, a batch of 12-megapixel photos many - they run one after the other through
                # ONE workspace (the side stream is in order), so that HBM holds a group, not a batch, of decoded photos
                st_ = np.lib.stride_tricks.as_strided
                H3 = st_(bigview[:4 * JPEG_HDR_INTS].view(np.int32), shape=(n, JPEG_HDR_INTS), strides=(cap, 4))[e3].astype(np.int64)
                need = H3[:, 7] * 192 + (H3[:, 1] * H3[:, 2] * 3 + 15) // 16 * 16 + H3[:, 9] * n_px * 3
                groups, lo, acc = [], 0, 0
                for k in range(len(e3)):
                    if k > lo and acc + need[k] > jpeg_group_bytes:
                        groups.append((lo, k))
                        lo, acc = k, 0
                    acc += int(need[k])
                groups.append((lo, len(e3)))
                status = torch.empty(len(e3), dtype=torch.int32, device=dev)
                calls, ws_max, rgb_max, tmp_max = [], 0, 0, 0
                for lo, hi in groups:
                    recs, tables, jobs, out_sz, blocks, nt = jpeg_records(bigview, n, cap, e3[lo:hi], comp, n_px)
                    w, h, nrows = recs["width"].astype(np.int64), recs["height"].astype(np.int64), jobs["nrows"].astype(np.int64)
                    o_tab = (recs.nbytes + 15) // 16 * 16
                    o_job = (o_tab + tables.nbytes + 15) // 16 * 16
                    small = np.zeros(o_job + jobs.nbytes, np.uint8)
                    small[:recs.nbytes] = recs.view(np.uint8).reshape(-1)
                    small[o_tab:o_tab + tables.nbytes] = tables
                    small[o_job:] = jobs.view(np.uint8).reshape(-1)
                    dsmall = torch.from_numpy(small).to(dev)
                    total_blocks = int(blocks.sum())
                    ws_bytes = int(L.clipmi_jpeg_workspace_bytes(total_blocks, nt))
                    ws_max, rgb_max = max(ws_max, ws_bytes), max(rgb_max, int(out_sz.sum()))
                    tmp_max = max(tmp_max, int((nrows * n_px * 3).sum()))
                    calls.append((dsmall, o_tab, o_job, hi - lo, nt, total_blocks, int(blocks.max()), int((w * h).max()), int(nrows.max()),
                                  ws_bytes, lo))
                    pending["keep"].append(dsmall)
                ws = torch.empty(ws_max, dtype=torch.uint8, device=dev)
                rgb = torch.empty(max(rgb_max, 16), dtype=torch.uint8, device=dev)
                scratch3 = torch.empty(max(tmp_max, 1), dtype=torch.uint8, device=dev)

                def decode_jpeg():
                    for dsmall, o_tab, o_job, n3, nt, total_blocks, mb, mp, mr, ws_bytes, lo in calls:
                        sb = dsmall.data_ptr()
                        rc = L.clipmi_jpeg_decode_rgb8(base, sb, n3, sb + o_tab, nt, total_blocks, mb, mp, rgb.data_ptr(),
                                                       status.data_ptr() + 4 * lo, ws.data_ptr(), ws_bytes, _lib.stream_ptr(dev))
                        _lib.check(rc, "clipmi_jpeg_decode_rgb8")
                        rc = L.clipmi_resize_crop_rgb8(rgb.data_ptr(), sb + o_job, n3, mr, base, n_px, devt.data_ptr(), scratch3.data_ptr(),
                                                       _lib.stream_ptr(dev))
                        _lib.check(rc, "clipmi_resize_crop_rgb8")

                pending["launch"].append(decode_jpeg)
                pending["keep"] += [ws, rgb, scratch3]
                pending["status"], pending["slots"] = status, e3
            ev_copy = torch.cuda.Event()
            ev_copy.record(copy_stream)
        side = _lib.side_stream(dev)[1]
        with torch.cuda.stream(side):
            side.wait_event(ev_copy)
            for launch in pending["launch"]:
                launch()
            ev = torch.cuda.Event()
            ev.record(side)
        ev_copy.synchronize()                                 # the segment is decoded into again two batches later
        pending["keep"].append(dbig)
        if slot is not None:
            slot["ev"] = ev_copy
        return ev, pending

    def to_device(host, slot):
        with torch.cuda.stream(copy_stream):
            devt = host.to(dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(copy_stream)
        if slot is not None:
            slot["ev"] = ev
        return devt, ev

    seg_index = [0]
    pin_small = os.environ.get("CLIPMI_PIN_SHM", "1") != "0"

    def copy_out(decoded, chunk):
        """shared memory -> pinned staging (GPU) or a private tensor (CPU) -> device. numpy copies on purpose: a 65-MB torch
        copy_ fans out over every CPU the host shows (256 here) and its OpenMP team then spins through the container's CPU
        share - every other batch's decode took 80 ms instead of 15 (tools/attic/pipe_probe.py: 26.7 k images/s decode only,
        5.9 k with a torch copy behind each batch)."""
        bigview = full = None
        if len(decoded[0]) == 4:
            (view, good, bigview, full), ok, bad = decoded
        else:
            (view, good), ok, bad = decoded
        if not ok:
            return ok, bad, None, None
        if not use_gpu:
            return ok, bad, torch.from_numpy(view[good] if len(ok) != len(chunk) else view.copy()), None
        small_used = any(g_ and k not in full for k, g_ in enumerate(good)) if full else True
        if not small_used:
            # every image of the batch sits in the big segment (full size, or as a parsed JPEG file): nothing to copy out of
            # the n_px x n_px slots
            with torch.cuda.stream(copy_stream):
                devt = torch.empty((len(ok), 3, n_px, n_px), dtype=torch.uint8, device=dev)
            ev = None
        elif len(ok) == len(chunk) and pin_small and pool.pin_segment(seg_index[0]):
            # the segment itself is page-locked (hipHostRegister): copy it to the device where it lies, and let this
            # thread wait for the copy (1-2 ms) - the segment is decoded into again two batches later
            devt, ev = to_device(torch.from_numpy(view), None)
            ev.synchronize()
        else:
            slot = staging(len(ok))
            if len(ok) == len(chunk):
                np.copyto(slot["np"][:len(ok)], view)
            else:
                np.compress(good, view, axis=0, out=slot["np"][:len(ok)])
            devt, ev = to_device(slot["buf"][:len(ok)], slot)
        pending = None
        if full:
            ev, pending = device_stage(devt, bigview, full, good)
            pending["chunk"], pending["good"] = chunk, good.copy()
        if jpeg_cap and full_cap[0]:
            # the next batches' JPEG regions: 1.25 x the largest file this batch held or turned away
            used3 = max([int(v[3]) for v in (full or {}).values() if v[0] == 3] + [pool.jpeg_wanted])
            pool.jpeg_wanted = 0
            if used3:
                pool.jpeg_cap_hint = min(jpeg_cap, max(1 << 16, (used3 + used3 // 4 + 65535) // 65536 * 65536))
                full_cap[0] = max(resize_cap, pool.jpeg_cap_hint)
        return ok, bad, devt, ev, pending

    def redo_on_host(bad_slots, chunk, good, ok, bad, devt):
        """Files the device decoder reported corrupt: Pillow decides (its error handling is the reference's) - its pixels replace
        the row, or the file joins the failed ones and its row leaves the batch."""
        comp = np.cumsum(good) - 1
        drop = []
        for s_ in bad_slots:
            try:
                px = torch.from_numpy(load_uint8(chunk[s_], n_px)).to(dev)
                devt[comp[s_]].copy_(px)
            except KeyboardInterrupt:
                raise
            except Exception:
                drop.append(s_)
        if drop:
            gone = {chunk[s_] for s_ in drop}
            keep_rows = torch.tensor([r for r in range(len(ok)) if r not in {int(comp[s_]) for s_ in drop}], dtype=torch.long, device=dev)
            devt = devt.index_select(0, keep_rows)
            ok = [p_ for p_ in ok if p_ not in gone]
            bad = [p_ for p_ in chunk if p_ in gone or p_ in set(bad)]
        torch.cuda.synchronize(dev)
        return ok, bad, devt

    def stage(chunk):
        """The thread form: decode on `workers` threads, stack, pin, copy."""
        arrs = list(tpool.map(_load_safe, [(p, n_px) for p in chunk]))
        ok = [p for p, a in zip(chunk, arrs) if a is not None]
        bad = [p for p, a in zip(chunk, arrs) if a is None]
        if not ok:
            return ok, bad, None, None
        good_arrs = [a for a in arrs if a is not None]
        if not use_gpu:
            return ok, bad, torch.from_numpy(np.stack(good_arrs)), None
        slot = staging(len(ok))                            # numpy writes straight into pinned memory (no torch copy: see copy_out)
        np.stack(good_arrs, out=slot["np"][:len(ok)])
        devt, ev = to_device(slot["buf"][:len(ok)], slot)
        return ok, bad, devt, ev

    chunks = [paths[i:i + batch] for i in range(0, len(paths), batch)]

    def consume(item):
        ok, bad, devt, ev = item[:4]
        pending = item[4] if len(item) > 4 else None
        feats = None
        t0 = time.perf_counter()
        if devt is not None:
            if ev is not None:
                torch.cuda.current_stream(dev).wait_event(ev)
            feats = model.encode_image(devt, normalize=True).cpu().numpy().astype("float32")
            if pending is not None and pending["status"] is not None:
                stc = pending["status"].cpu().numpy()         # (behind the encode step: nothing waits for it in the common case)
                if stc.any():
                    ok, bad, devt = redo_on_host([int(s_) for s_ in pending["slots"][stc != 0]], pending["chunk"],
                                                 pending["good"], ok, bad, devt)
                    feats = model.encode_image(devt, normalize=True).cpu().numpy().astype("float32") if len(ok) else None
        if stats is not None:
            stats["encode_s"] = stats.get("encode_s", 0.0) + time.perf_counter() - t0
        return ok, feats, bad

    if pool is not None:
        # three stages, one thread each: decode batch j+1 (worker processes, shared-memory segment (j+1) & 1) | copy batch j
        # out of its segment and to the device | encode batch j-1 here. A segment is decoded into again only after its
        # previous batch has been copied out.
        with ThreadPoolExecutor(max_workers=1) as dec, ThreadPoolExecutor(max_workers=1) as cpy:
            copies = {}

            def decode_job(j):
                if j - 2 in copies:
                    copies[j - 2].result()                 # segment j & 1 is free again
                t0 = time.perf_counter()
                r = [pool.decode(chunks[j], n_px, copy=False, segment=j & 1, full_cap=full_cap[0], full_mode=full_mode)]
                if stats is not None:
                    stats["decode_s"] = stats.get("decode_s", 0.0) + time.perf_counter() - t0
                return r

            def copy_job(d, j):
                seg_index[0] = j & 1
                dec_ = d.result().pop()
                t0 = time.perf_counter()
                r = copy_out(dec_, chunks[j])
                if stats is not None:
                    stats["copy_s"] = stats.get("copy_s", 0.0) + time.perf_counter() - t0
                    if len(dec_[0]) == 4:
                        stats["jpeg_files"] = stats.get("jpeg_files", 0) + sum(1 for v in dec_[0][3].values() if v[0] == 3)
                return r

            def submit(j):
                d = dec.submit(decode_job, j)              # (the result travels in a list the copy stage empties: no
                copies[j] = cpy.submit(copy_job, d, j)     # view outlives its copy)

            for j in range(min(2, len(chunks))):
                submit(j)
            for ci in range(len(chunks)):
                item = copies[ci].result()
                if ci + 2 < len(chunks):
                    submit(ci + 2)
                copies.pop(ci - 2, None)
                yield consume(item)
        return
    with ThreadPoolExecutor(max_workers=workers) as tpool, ThreadPoolExecutor(max_workers=1) as stager:
        nxt = stager.submit(stage, chunks[0]) if chunks else None
        for ci in range(len(chunks)):
            item = nxt.result()
            nxt = stager.submit(stage, chunks[ci + 1]) if ci + 1 < len(chunks) else None
            yield consume(item)
