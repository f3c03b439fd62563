"""Video egress: colour conversion bit-exact against the numpy oracle, container structure, and the
device -> pinned ring -> file path against oracle-assembled bytes."""
import struct

import numpy as np
import pytest
import torch


def test_fixed_point_matrix_close_to_real():
    from oracle.video_oracle import rgb_to_yuv420, rgb_to_yuv_real
    rgb = np.random.default_rng(0).integers(0, 256, (2, 16, 24, 3), dtype=np.uint8)
    rgb[0, :2, :2] = 255
    rgb[0, 2:4, :2] = 0
    y, u, v = rgb_to_yuv420(rgb)
    yr, ur, vr = rgb_to_yuv_real(rgb)
    assert np.abs(y - yr).max() <= 0.51 and np.abs(u - ur).max() <= 0.51 and np.abs(v - vr).max() <= 0.51
    assert y[0, 0, 0] == 255 and u[0, 0, 0] == 128 and v[0, 0, 0] == 128 and y[0, 2, 0] == 0


def test_avi_container_cpu(tmp_path):
    from pyflow_hip.video_io import export_to_video
    rgb = torch.from_numpy(np.random.default_rng(1).integers(0, 256, (3, 6, 10, 3), dtype=np.uint8))
    path = export_to_video(rgb, str(tmp_path / "a.avi"), fps=24)
    raw = open(path, "rb").read()
    assert raw[:4] == b"RIFF" and raw[8:12] == b"AVI " and struct.unpack("<I", raw[4:8])[0] == len(raw) - 8
    row = (10 * 3 + 3) & ~3
    k = raw.index(b"movi") + 4
    for t in range(3):
        assert raw[k:k + 4] == b"00db" and struct.unpack("<I", raw[k + 4:k + 8])[0] == row * 6
        frame = np.frombuffer(raw[k + 8:k + 8 + row * 6], dtype=np.uint8).reshape(6, row)[:, :30].reshape(6, 10, 3)
        assert np.array_equal(frame[::-1, :, ::-1], rgb[t].numpy())          # bottom-up BGR
        k += 8 + row * 6
    assert raw[k:k + 4] == b"idx1"
    with pytest.raises(RuntimeError, match="y4m"):
        export_to_video(rgb, str(tmp_path / "a.mp4"))
    # list-of-arrays input, like the reference's list of PIL images
    p2 = export_to_video([f.numpy() for f in rgb], str(tmp_path / "b.avi"))
    assert open(p2, "rb").read() == raw


@pytest.mark.gpu
@pytest.mark.parametrize("T,H,W,block", [(5, 16, 24, 2), (33, 64, 96, 16), (1, 2, 2, 16)])
def test_y4m_device_path_bit_exact(tmp_path, T, H, W, block):
    from pyflow_hip.video_io import export_to_video, rgb_to_yuv420_device
    from oracle.video_oracle import rgb_to_yuv420, y4m_bytes
    rgb = np.random.default_rng(T).integers(0, 256, (T, H, W, 3), dtype=np.uint8)
    rgb[0, :2, :2] = 255
    d = torch.from_numpy(rgb).cuda()
    y, u, v = rgb_to_yuv420_device(d)
    yo, uo, vo = rgb_to_yuv420(rgb)
    assert np.array_equal(y.cpu().numpy(), yo) and np.array_equal(u.cpu().numpy(), uo) and np.array_equal(v.cpu().numpy(), vo)
    path = export_to_video(d, str(tmp_path / "v.y4m"), fps=24, block_frames=block)
    assert open(path, "rb").read() == y4m_bytes(rgb, 24)
    pa = export_to_video(d, str(tmp_path / "v.avi"), fps=24, block_frames=block)
    assert open(pa, "rb").read() == open(export_to_video(torch.from_numpy(rgb), str(tmp_path / "h.avi")), "rb").read()
