"""TEST INFRASTRUCTURE ONLY (oracle) -- numpy restatement of the colour conversion of the video egress
(pyflow_hip/video_io.py, pf_rgb_to_yuv420).  The reference never converts colours itself: it passes PIL frames to
diffusers.utils.export_to_video (inference_multigpu.py:92), whose encoder (imageio-ffmpeg / OpenCV, neither in this
image) does.  PARITY UNPINNED against the reference: there is no reference output to pin to; the formulas are the
JFIF (ITU-T T.871) full-range BT.601 matrix in the 16-bit fixed point libjpeg uses, and the test also bounds the
distance to the real-valued matrix.  Never imported by the product path."""
import numpy as np


def rgb_to_yuv420(rgb):
    """rgb uint8 [T,H,W,3] -> Y [T,H,W], Cb, Cr [T,H/2,W/2] (uint8)."""
    r = rgb[..., 0].astype(np.int64)
    g = rgb[..., 1].astype(np.int64)
    b = rgb[..., 2].astype(np.int64)
    y = (19595 * r + 38470 * g + 7471 * b + 32768) >> 16
    T, H, W = r.shape
    s = lambda a: a.reshape(T, H // 2, 2, W // 2, 2).sum(axis=(2, 4))            # noqa: E731
    sr, sg, sb = s(r), s(g), s(b)
    cb = (-11059 * sr - 21709 * sg + 32768 * sb + (128 << 18) + (1 << 17)) >> 18
    cr = (32768 * sr - 27439 * sg - 5329 * sb + (128 << 18) + (1 << 17)) >> 18
    return y.astype(np.uint8), np.clip(cb, 0, 255).astype(np.uint8), np.clip(cr, 0, 255).astype(np.uint8)


def rgb_to_yuv_real(rgb):
    """the real-valued JFIF matrix on the 2x2 block means (for the error bound of the fixed-point form)"""
    x = rgb.astype(np.float64)
    r, g, b = x[..., 0], x[..., 1], x[..., 2]
    y = 0.299 * r + 0.587 * g + 0.114 * b
    T, H, W = r.shape
    m = lambda a: a.reshape(T, H // 2, 2, W // 2, 2).mean(axis=(2, 4))           # noqa: E731
    mr, mg, mb = m(r), m(g), m(b)
    cb = 128 - 0.168736 * mr - 0.331264 * mg + 0.5 * mb
    cr = 128 + 0.5 * mr - 0.418688 * mg - 0.081312 * mb
    return y, cb, cr


def y4m_bytes(rgb, fps):
    T, H, W, _ = rgb.shape
    y, u, v = rgb_to_yuv420(rgb)
    out = [f"YUV4MPEG2 W{W} H{H} F{int(fps)}:1 Ip A1:1 C420jpeg XYSCSS=420JPEG XCOLORRANGE=FULL\n".encode()]
    for t in range(T):
        out += [b"FRAME\n", y[t].tobytes(), u[t].tobytes(), v[t].tobytes()]
    return b"".join(out)


def diffusers_export_calls(video_frames, output_video_path, fps=10, backend="imageio"):
    """What `diffusers.utils.export_to_video(video_frames, output_video_path, fps)` -- the call the reference makes at
    inference_multigpu.py:92,117, app.py:207,260 with a list of 241 PIL images and fps = 24 -- HANDS TO ITS ENCODER, restated
    from the published source of diffusers 0.30.x (src/diffusers/utils/export_utils.py; the reference pins `diffusers>=0.30.1`,
    requirements.txt:6; the package itself is not in this image):
      * a list of numpy arrays is float in [0, 1]: each frame -> (frame * 255).astype(np.uint8)        (export_utils.py)
      * a list of PIL images: each frame -> np.array(frame)                                               (H x W x 3, RGB)
      * imageio backend (the default when imageio + imageio-ffmpeg import): `imageio.get_writer(path, fps=fps)` and one
        `writer.append_data(frame)` per frame, RGB order, in sequence;
      * legacy OpenCV backend: `cv2.VideoWriter(path, fourcc('mp4v'), fps, (w, h))` and one `write(cv2.cvtColor(frame,
        cv2.COLOR_RGB2BGR))` per frame.
    Returns (open_call, [frame arrays in the order and channel order the encoder receives them]).  PARITY UNPINNED against a
    reference RUN (no encoder exists here): this pins the bytes that reach the encoder, not the encoded file."""
    frames = list(video_frames)
    if isinstance(frames[0], np.ndarray):
        frames = [(f * 255).astype(np.uint8) for f in frames]
    else:                                                  # PIL.Image.Image
        frames = [np.array(f) for f in frames]
    if backend == "imageio":
        return ("get_writer", output_video_path, dict(fps=fps)), frames
    h, w, _ = frames[0].shape
    return ("VideoWriter", output_video_path, dict(fourcc="mp4v", fps=fps, size=(w, h))), [f[:, :, ::-1] for f in frames]
