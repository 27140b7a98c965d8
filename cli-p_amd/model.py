"""Host-side mirror of the `clip` calls the reference scripts make:

    model, transform = clip.load("ViT-B/32", device=device, jit=False)   build-index.py:18, query-index.py:21
    model.eval()                                                          build-index.py:20
    transform(image).unsqueeze(0).to(device)                              build-index.py:48
    model.encode_image(image)                                             build-index.py:49
    clip.tokenize([in_text]).to(device)                                   query-index.py:107
    model.encode_text(texts)                                              query-index.py:108

Same names, argument meaning and error behaviour; the arithmetic runs in libclipmi.so
(clipmi_encode_image / clipmi_encode_text). Differences a caller can see: weights must come from
a LOCAL file (no download); batches are first-class (the reference feeds B = 1).
"""
import os

import numpy as np
import torch

from . import _lib, weights

_DTYPES = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16, torch.uint8: _lib.U8}

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


class _Visual:
    def __init__(self, res):
        self.input_resolution = res


class CLIP:
    """Both towers resident in HBM as packed blobs; encode_* enqueue the HIP kernel sequences on
    torch's current stream and return device tensors (f32 [B, E])."""

    def __init__(self, state_dict, device="cuda:0", vision_weights="bf16"):
        """vision_weights="fp8": the image tower's linear layers run on the FP8 matrix cores (e4m3 weights with
        per-channel scales, activations quantised per row on the fly; BASELINE.json configs[4]). Lower accuracy
        than bf16 (tests state the measured tolerance); the text tower and the search are unchanged."""
        self.device = torch.device(device)
        self.dims = weights.infer_dims(state_dict)
        self.vision_weights = vision_weights
        self.vision, self._vblob = weights.pack_vision(state_dict, self.device, weight_format=vision_weights)
        self.text, self._tblob = weights.pack_text(state_dict, self.device)
        self.visual = _Visual(self.dims["res"])
        self.context_length = self.dims["ctx"]
        self.embed_dim = self.dims["embed"]
        self._ws = {}                # workspace per (HIP stream, tower): concurrent encoders never share one
        self.max_batch = 1024        # images per kernel sequence at most (see image_chunks)
        self.round_chunks = True     # cut inputs at whole rounds of GEMM tiles (False: max_batch-sized chunks)

    def eval(self):
        return self

    def _workspace(self, need, stream_key, tower):
        """One workspace per (stream, tower). A block that has to grow is dropped only after the stream that used it
        has drained (a raw hipStream_t passed by the caller is unknown to torch's caching allocator, which would
        otherwise hand the freed block to another tensor while kernels still write to it)."""
        key = (stream_key, tower)
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            if ws is not None:
                torch.cuda.synchronize(self.device)
            ws = self._ws[key] = torch.empty(int(need), dtype=torch.uint8, device=self.device)
        return ws

    def _require_gpu(self, what):
        if self.device.type != "cuda":
            raise _lib.ClipmiError(f"{what} needs the HIP path (device {self.device} is not a GPU); no CPU fallback")

    def image_chunk(self, limit=None):
        """The largest image count <= limit (default max_batch) whose tokens fill whole rounds of 256 x 256 output tiles
        on the 256 CUs for the narrowest GEMM (N = width): 435 / 870 for ViT-B/32 (one / two rounds). 0 if even one round
        is more than `limit` images."""
        limit = self.max_batch if limit is None else limit
        W, Lv = self.dims["v_width"], self.dims["v_tokens"]
        cols = max(1, W // 256)
        best = 0
        for rounds in range(1, 4096):
            imgs = (rounds * 256 // cols) * 256 // Lv
            if imgs > limit:
                break
            best = imgs
        return best

    chunks_in_flight = 2          # kernel sequences of ONE encode_image kept in flight (caller's stream + one side stream; 1 = off)

    def image_lanes(self, B):
        """How encode_image runs B images (round 5): [(lo, hi, lane)] - kernel sequences alternating between the caller's stream
        (lane 0) and ONE internal stream (lane 1). An input of two or more WHOLE one-round chunks (870, 1305, 1740 .. images for
        ViT-B/32) is cut into those chunks (435): two sequences of 435 in flight encode 870 images 2.2 % faster than one of 870 (same
        box: 103.8 k -> 106.1 k images/s; one stream's kernel boundaries, ramps and HBM-bound store passes run beside the other's
        K-loops; two sequences of 870: 105.4 k; three / four lanes lose it again: 103.5 k / 102.7 k -
        tools/attic/encode_two_streams.py). Smaller inputs, and chunks_in_flight = 1, keep image_chunks()' one stream. Results do
        not depend on the cut (test_encode_image_batch_invariance_and_dtypes, test_encode_image_two_sequences_in_flight)."""
        one = self.image_chunk(limit=max(1, self.max_batch // 2)) if self.round_chunks else 0
        if self.chunks_in_flight > 1 and one and B >= 2 * one and B % one == 0:
            # (whole one-round chunks only: a small remainder as a sequence of its own costs more than the overlap gives -
            #  1000 images 95.3 k -> 93.3 k images/s, 1024: 94.6 k -> 90.7 k, against 870 / 1740 / 2610: +2.1 ... 2.3 %)
            chunks = [one] * (B // one)
            lanes = 2
        else:
            chunks, lanes = self.image_chunks(B), 1
        out, lo = [], 0
        for i, n_ in enumerate(chunks):
            out.append((lo, lo + n_, i % lanes))
            lo += n_
        return out

    def image_chunks(self, B):
        """How encode_image cuts B images into kernel sequences: one sequence up to max_batch; above it whole-round
        chunks first and the remainder last (1740 = 870 + 870: 95.5 k -> 102.9 k images/s, 1305 = 870 + 435: 88 k -> 101 k;
        they used to be 1024 + rest). Cutting inputs BELOW max_batch at round boundaries was measured too
        (tools/attic/chunk_sweep.py): +10 % at 436 images, -9 % at 700 - the kernel choice per GEMM already handles a ragged last
        round, so those stay whole. Results do not depend on the cut (test_encode_image_batch_invariance_and_dtypes)."""
        if B <= self.max_batch:
            return [B] if B > 0 else []
        step = (self.image_chunk() if self.round_chunks else 0) or self.max_batch
        out, left = [], B
        while left > self.max_batch:
            out.append(step)
            left -= step
        if left > 0:
            out.append(left)
        return out

    def encode_image(self, image, normalize=False, out=None, stream=None):
        """image: [B,3,R,R] f32/bf16 (output of `transform`, already normalised) or uint8 raw RGB
        (normalisation fused on the device). Returns f32 [B,E]; `normalize=True` also applies
        build-index.py:50 (x / x.norm(dim=-1, keepdim=True)) in the same stream.
        `out` (f32 [B,E] device tensor) and `stream` (a raw hipStream_t as int, e.g. one created with a
        CU mask) let a caller run several encoders concurrently; defaults: fresh tensor, torch's stream."""
        self._require_gpu("encode_image")
        L = _lib.lib()
        if not isinstance(image, torch.Tensor):
            image = torch.as_tensor(image)
        R = self.dims["res"]
        if image.dim() != 4 or image.shape[1] != 3 or image.shape[2] != R or image.shape[3] != R:
            raise ValueError(f"encode_image: expected [B,3,{R},{R}], got {tuple(image.shape)}")
        if image.dtype not in _DTYPES:
            image = image.float()
        image = image.to(self.device).contiguous()
        B = image.shape[0]
        if out is None:
            out = torch.empty((B, self.embed_dim), dtype=torch.float32, device=self.device)
        import ctypes as _C

        def run(lo, hi, sp):
            need = L.clipmi_encode_image_workspace_bytes(self.vision, hi - lo)
            if need == 0:
                raise _lib.ClipmiError("encode_image: " + _lib.last_error())
            ws = self._workspace(need, sp.value, "vision")            # every stream owns its workspace
            rc = L.clipmi_encode_image(self.vision, self._vblob.data_ptr(), image[lo:hi].data_ptr(),
                                       _DTYPES[image.dtype], hi - lo, out[lo:hi].data_ptr(), int(bool(normalize)),
                                       ws.data_ptr(), ws.numel(), sp)
            _lib.check(rc, "clipmi_encode_image")

        lanes = self.image_lanes(B)
        if stream is not None or all(l == 0 for _, _, l in lanes):
            # a caller's own raw stream, or nothing to overlap: one kernel sequence after the other on that stream
            sp = _lib.stream_ptr(self.device) if stream is None else _C.c_void_p(int(stream))
            for lo, hi, _ in lanes:
                run(lo, hi, sp)
            return out
        # chunks alternate between the caller's stream and ONE side stream (as IndexFlatIP._search_pipelined: this ROCm gives only
        # a process's first streams a hardware queue of their own); the caller's stream waits for the side stream at the end
        cur, side = _lib.side_stream(self.device)              # one internal stream per process and caller's stream
        side.wait_stream(cur)                                  # the pixels, `out` and the weights are ready
        for lo, hi, lane in lanes:
            with torch.cuda.stream(side if lane else cur):
                run(lo, hi, _lib.stream_ptr(self.device))
        cur.wait_stream(side)
        return out

    def encode_text(self, text, normalize=False):
        """text: int [Q, ctx] token ids (clip.tokenize output). Returns f32 [Q,E]."""
        self._require_gpu("encode_text")
        L = _lib.lib()
        if not isinstance(text, torch.Tensor):
            text = torch.as_tensor(text)
        if text.dim() != 2 or text.shape[1] != self.context_length:
            raise ValueError(f"encode_text: expected [Q,{self.context_length}], got {tuple(text.shape)}")
        # Causal attention: rows behind the EOT token (the pooled row = the first argmax of the ids, as upstream takes it)
        # cannot reach it, so the tower runs on the first max(EOT) + 1 positions only - the same bits for a fraction of
        # the rows (a typical prompt is ~10 tokens of the 77). Done when the ids are still on the host (the tokenizer's
        # output, query-index.py:107): no device round trip is spent on finding the length.
        tower, Lp = self.text, self.context_length
        if text.device.type == "cpu" and text.numel():
            Lp = min(self.context_length, int(text.argmax(dim=1).max()) + 1)
            if Lp < self.context_length:
                tower = type(self.text).from_buffer_copy(self.text)
                tower.tokens = Lp
                text = text[:, :Lp]
        ids = text.to(device=self.device, dtype=torch.int32).contiguous()
        Q = ids.shape[0]
        out = torch.empty((Q, self.embed_dim), dtype=torch.float32, device=self.device)
        for lo in range(0, Q, self.max_batch):
            hi = min(Q, lo + self.max_batch)
            need = L.clipmi_encode_text_workspace_bytes(tower, hi - lo)
            if need == 0:
                raise _lib.ClipmiError("encode_text: " + _lib.last_error())
            ws = self._workspace(need, _lib.stream_ptr(self.device).value, "text")
            rc = L.clipmi_encode_text(tower, self._tblob.data_ptr(), ids[lo:hi].data_ptr(), hi - lo,
                                      out[lo:hi].data_ptr(), int(bool(normalize)), ws.data_ptr(), ws.numel(),
                                      _lib.stream_ptr(self.device))
            _lib.check(rc, "clipmi_encode_text")
        return out


def make_transform(n_px):
    """The upstream `_transform(n_px)`: Resize(n_px, bicubic) on the shorter side, CenterCrop(n_px),
    RGB, ToTensor, Normalize(CLIP mean/std) — restated with Pillow + numpy (torchvision is not a
    dependency). Returns f32 [3, n_px, n_px]."""
    from PIL import Image

    mean = np.asarray(CLIP_MEAN, np.float32).reshape(3, 1, 1)
    std = np.asarray(CLIP_STD, np.float32).reshape(3, 1, 1)

    def transform(img):
        w, h = img.size
        if not (w <= h and w == n_px) and not (h <= w and h == n_px):
            if w <= h:
                nw, nh = n_px, int(n_px * h / w)
            else:
                nh, nw = n_px, int(n_px * w / h)
            img = img.resize((nw, nh), Image.BICUBIC)
            w, h = nw, nh
        left = int(round((w - n_px) / 2.0))
        top = int(round((h - n_px) / 2.0))
        img = img.crop((left, top, left + n_px, top + n_px)).convert("RGB")
        a = np.asarray(img, dtype=np.float32).transpose(2, 0, 1) / 255.0
        return torch.from_numpy((a - mean) / std)

    return transform


def available_models():
    return [k for k in weights.ARCHS if not k.startswith("toy")]


def load(name, device="cuda" if torch.cuda.is_available() else "cpu", jit=False, seed=None):
    """clip.load stand-in. `name` is a path to a LOCAL checkpoint with OpenAI key names
    (TorchScript archive such as ViT-B-32.pt, pickled state-dict, or safetensors). An architecture
    name ("ViT-B/32") is resolved through $CLIPMI_WEIGHTS_DIR/<name with / -> ->.pt; with
    `seed` given (or CLIPMI_RANDOM_WEIGHTS=<seed>) seeded random weights of that architecture are
    used instead — for synthetic benchmarks and tests only. Nothing is ever downloaded."""
    if device == "cuda":
        device = "cuda:0"
    if os.path.exists(name):
        sd = weights.load_state_dict(name)
    elif name in weights.ARCHS:
        env_seed = os.environ.get("CLIPMI_RANDOM_WEIGHTS")
        wdir = os.environ.get("CLIPMI_WEIGHTS_DIR")
        cand = os.path.join(wdir, name.replace("/", "-").replace("@", "-") + ".pt") if wdir else None
        if seed is None and cand and os.path.exists(cand):
            sd = weights.load_state_dict(cand)
        elif seed is not None or env_seed is not None:
            sd = weights.random_state_dict(name, seed=int(seed if seed is not None else env_seed))
        else:
            raise RuntimeError(
                f"Model {name}: no local weights. This build never downloads: pass a checkpoint path, set "
                "CLIPMI_WEIGHTS_DIR to a directory holding e.g. ViT-B-32.pt, or set CLIPMI_RANDOM_WEIGHTS=<seed> "
                "for synthetic weights.")
    else:
        raise RuntimeError(f"Model {name} not found; available models = {available_models()}")
    model = CLIP(sd, device=device)
    return model, make_transform(model.visual.input_resolution)
