"""Video egress: the step after the sampling path (``export_to_video(frames, path, fps=24)`` from diffusers.utils,
called at inference_multigpu.py:92,117, app.py:207,260, scripts/app_multigpu_engine.py:100,123).

The reference hands 241 PIL images to diffusers, which encodes H.264 through imageio/ffmpeg or OpenCV -- none of which
exists in this image (no network).  What is built here is the device-side half that matters on an MI355X box and the
container formats that need no codec:

* the uint8 frames stay on the device until they leave: ``FrameRing`` copies them to PINNED host slots on a side HIP
  stream in blocks of frames (event-tracked, so the copy of block i overlaps the colour conversion of block i+1 and the
  file write of block i-1) instead of one pageable 711 MB ``.cpu()`` + 241 PIL objects;
* ``.y4m`` (YUV4MPEG2 4:2:0, what every encoder CLI ingests): RGB -> YCbCr planes by ``pf_rgb_to_yuv420`` on the device
  (JFIF full-range BT.601, 16-bit fixed point), 1.5 B per pixel over PCIe instead of 3;
* ``.avi``: uncompressed RGB24 (bottom-up BGR DIB frames) for players that want a self-contained file;
* ``.mp4`` / other codec containers: delegated to imageio or OpenCV exactly like diffusers does when one of them is
  importable, otherwise a clear error naming the two codec-free formats.
"""
import os
import struct
import ctypes as C

import numpy as np
import torch

from . import lib as L
from .lib import check


def _as_u8_tensor(frames):
    """list of PIL / numpy HxWx3 uint8, or a uint8 tensor [T,H,W,3] (host or device) -> tensor [T,H,W,3]."""
    if isinstance(frames, torch.Tensor):
        assert frames.dtype == torch.uint8 and frames.dim() == 4 and frames.shape[-1] == 3
        return frames.contiguous()
    def one(f):
        if hasattr(f, "convert"):                              # PIL image: diffusers takes np.array(frame)
            return np.asarray(f.convert("RGB"), dtype=np.uint8)
        f = np.asarray(f)
        if np.issubdtype(f.dtype, np.floating):                # numpy float frames are in [0, 1]: (frame * 255).astype(uint8),
            return (f * 255).astype(np.uint8)                  # exactly diffusers' export_to_video
        return f.astype(np.uint8, copy=False)
    return torch.from_numpy(np.stack([one(f) for f in frames], 0))


class FrameRing:
    """device -> pinned-host transfer of byte planes in blocks, on its own HIP stream."""

    def __init__(self, slot_bytes, slots=3, device="cuda"):
        self.dev = torch.device(device)
        self.slots = [torch.empty(slot_bytes, dtype=torch.uint8).pin_memory() for _ in range(slots)]
        self.events = [None] * slots
        self.stream = torch.cuda.Stream(device=self.dev)
        self.k = 0

    def push(self, src):
        """enqueue the copy of a flat uint8 device tensor; returns (slot index, nbytes).  The producer stream is the
        current stream; the slot is reused only after `wait(slot)` was called by the consumer."""
        i = self.k % len(self.slots)
        self.k += 1
        n = src.numel()
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.dev))
        self.stream.wait_event(ready)
        with torch.cuda.stream(self.stream):
            self.slots[i][:n].copy_(src.view(-1), non_blocking=True)
            src.record_stream(self.stream)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.events[i] = ev
        return i, n

    def wait(self, i, n):
        self.events[i].synchronize()
        return self.slots[i][:n].numpy()


def frames_to_host(frames, pinned=None):
    """uint8 frames [T,H,W,3] on the device -> the same bytes in PINNED host memory (one asynchronous copy on the current
    stream + a stream sync).  `pinned`: a reusable pinned uint8 tensor of at least that many bytes (allocated when None).
    This is the end point of the sampling metric ("uint8 frames in host memory")."""
    n = frames.numel()
    if pinned is None or pinned.numel() < n:
        pinned = torch.empty(n, dtype=torch.uint8).pin_memory()
    host = pinned[:n].view(frames.shape)
    host.copy_(frames, non_blocking=True)
    torch.cuda.current_stream(frames.device).synchronize()
    return host


def _yuv_call(rgb, y_ptr, u_ptr, v_ptr, y_fs, c_fs):
    T, H, W, _ = rgb.shape
    check(L.load().pf_rgb_to_yuv420(C.c_void_p(rgb.data_ptr()), C.c_void_p(y_ptr), C.c_void_p(u_ptr), C.c_void_p(v_ptr),
                                    C.c_int(T), C.c_int(H), C.c_int(W), C.c_longlong(y_fs), C.c_longlong(c_fs),
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream)))


def rgb_to_yuv420_device(rgb):
    """uint8 [T,H,W,3] device tensor -> (Y [T,H,W], U [T,H/2,W/2], V [T,H/2,W/2]) device tensors."""
    T, H, W, _ = rgb.shape
    y = torch.empty(T, H, W, dtype=torch.uint8, device=rgb.device)
    u = torch.empty(T, H // 2, W // 2, dtype=torch.uint8, device=rgb.device)
    v = torch.empty_like(u)
    _yuv_call(rgb, y.data_ptr(), u.data_ptr(), v.data_ptr(), 0, 0)
    return y, u, v


def rgb_to_i420_frames_device(rgb):
    """uint8 [T,H,W,3] device tensor -> [T, H*W*3/2] whole I420 frames [Y | Cb | Cr], written in place by the kernel"""
    T, H, W, _ = rgb.shape
    per = H * W * 3 // 2
    out = torch.empty(T, per, dtype=torch.uint8, device=rgb.device)
    p0 = out.data_ptr()
    _yuv_call(rgb, p0, p0 + H * W, p0 + H * W + (H // 2) * (W // 2), per, per)
    return out


def y4m_header(W, H, fps):
    return f"YUV4MPEG2 W{W} H{H} F{int(fps)}:1 Ip A1:1 C420jpeg XYSCSS=420JPEG XCOLORRANGE=FULL\n".encode()


def _write_y4m(frames, path, fps, block):
    T, H, W, _ = frames.shape
    if (H | W) & 1:
        raise ValueError("y4m 4:2:0 needs even frame dimensions")
    if not frames.is_cuda:
        # PIL / numpy / host frames (what generate() returns by default, output_type="pil"): the colour conversion runs
        # on the device, so they are uploaded first (one H2D of the uint8 frames)
        if not torch.cuda.is_available():
            raise RuntimeError("the .y4m writer converts colours on the MI355X: no device is visible")
        frames = frames.cuda()
    per = H * W * 3 // 2
    ring = FrameRing(per * block, slots=3, device=frames.device)
    pending = []
    with open(path, "wb") as f:
        f.write(y4m_header(W, H, fps))

        def drain(upto):
            while len(pending) > upto:
                i, n, nf = pending.pop(0)
                host = ring.wait(i, n).reshape(nf, per)
                for k in range(nf):
                    f.write(b"FRAME\n")
                    f.write(host[k].tobytes())

        for t0 in range(0, T, block):
            nf = min(block, T - t0)
            packed = rgb_to_i420_frames_device(frames[t0:t0 + nf])          # one host slot holds whole frames
            drain(len(ring.slots) - 1)
            i, n = ring.push(packed)
            pending.append((i, n, nf))
        drain(0)


def _avi_headers(W, H, T, fps):
    row = (W * 3 + 3) & ~3
    fsz = row * H
    def chunk(tag, data):
        return tag + struct.pack("<I", len(data)) + data + (b"\0" if len(data) & 1 else b"")
    avih = struct.pack("<IIIIIIIIII4I", int(1e6 / fps), fsz * int(fps), 0, 0x10, T, 0, 1, fsz, W, H, 0, 0, 0, 0)
    strh = struct.pack("<4s4sIHHIIIIIIIIhhhh", b"vids", b"DIB ", 0, 0, 0, 0, 1, int(fps), 0, T, fsz, 0xFFFFFFFF, 0, 0, 0, W, H)
    strf = struct.pack("<IiiHHIIiiII", 40, W, H, 1, 24, 0, fsz, 0, 0, 0, 0)
    strl = b"strl" + chunk(b"strh", strh) + chunk(b"strf", strf)
    hdrl = b"hdrl" + chunk(b"avih", avih) + chunk(b"LIST", strl)
    movi_size = 4 + T * (8 + fsz)
    idx = b"".join(struct.pack("<4sIII", b"00db", 0x10, 4 + k * (8 + fsz), fsz) for k in range(T))
    body_len = 4 + (8 + len(hdrl)) + (8 + movi_size) + (8 + len(idx))
    head = b"RIFF" + struct.pack("<I", body_len) + b"AVI " + chunk(b"LIST", hdrl) + b"LIST" + struct.pack("<I", movi_size) + b"movi"
    return head, chunk(b"idx1", idx), row, fsz


def _write_avi(frames, path, fps, block):
    T, H, W, _ = frames.shape
    head, tail, row, fsz = _avi_headers(W, H, T, fps)
    if 12 + len(head) + T * (8 + fsz) + len(tail) >= 2 ** 32:
        raise ValueError("uncompressed AVI is limited to 4 GiB: use .y4m")
    ring = FrameRing(H * W * 3 * block, slots=3, device=frames.device) if frames.is_cuda else None
    pad = b"\0" * (row - W * 3)
    with open(path, "wb") as f:
        f.write(head)

        def emit(host):            # host [nf,H,W,3] RGB -> bottom-up BGR rows
            bgr = host[:, ::-1, :, ::-1]
            for k in range(bgr.shape[0]):
                f.write(b"00db" + struct.pack("<I", fsz))
                if pad:
                    for r in range(H):
                        f.write(bgr[k, r].tobytes() + pad)
                else:
                    f.write(np.ascontiguousarray(bgr[k]).tobytes())

        if ring is None:
            emit(frames.numpy())
        else:
            pending = []
            for t0 in range(0, T, block):
                nf = min(block, T - t0)
                while len(pending) > len(ring.slots) - 2:
                    i, n, m = pending.pop(0)
                    emit(ring.wait(i, n).reshape(m, H, W, 3))
                i, n = ring.push(frames[t0:t0 + nf].reshape(-1))
                pending.append((i, n, nf))
            for i, n, m in pending:
                emit(ring.wait(i, n).reshape(m, H, W, 3))
        f.write(tail)


def export_to_video(video_frames, output_video_path=None, fps=24, block_frames=16):
    """diffusers.utils.export_to_video signature (default path: a temp .y4m).  Frames: list of PIL / numpy images or the
    uint8 [T,H,W,3] tensor `generate(..., output_type="uint8")` returns (kept on the device until written)."""
    if output_video_path is None:
        import tempfile
        output_video_path = tempfile.NamedTemporaryFile(suffix=".y4m", delete=False).name
    frames = _as_u8_tensor(video_frames)
    ext = os.path.splitext(output_video_path)[1].lower()
    if ext == ".y4m":
        _write_y4m(frames, output_video_path, fps, block_frames)
    elif ext == ".avi":
        _write_avi(frames, output_video_path, fps, block_frames)
    else:
        host = frames.cpu().numpy()
        try:
            import imageio                                    # what diffusers uses when present
            with imageio.get_writer(output_video_path, fps=fps) as w:
                for fr in host:
                    w.append_data(fr)
        except ImportError:
            try:
                import cv2
                h, w_ = host.shape[1:3]
                vw = cv2.VideoWriter(output_video_path, cv2.VideoWriter_fourcc(*"mp4v"), fps, (w_, h))
                for fr in host:
                    vw.write(fr[:, :, ::-1])
                vw.release()
            except ImportError:
                raise RuntimeError(f"no video encoder (imageio / OpenCV) is installed for '{ext}': write '.y4m' "
                                   "(YUV 4:2:0, encoder input) or '.avi' (uncompressed RGB) instead") from None
    return output_video_path
