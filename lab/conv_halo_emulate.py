"""CPU emulation of the INDEX ARITHMETIC of lab/conv_halo_lab.hip (not of its timing): the LDS-DMA placement of the halo and
of the filter slices (swizzle on the source side, filter-row permutation), the fragment read addresses of every tap, the
v_mfma_f32_16x16x32_bf16 operand / result lane layouts and the register epilogue are transcribed formula by formula and
run on one workgroup tile with numpy; the result must equal a direct 3 x 3 x 3 convolution.  Also checks that every
16-lane fragment group touches 16 distinct 16-byte bank groups (no LDS bank conflicts) for all nine tap shifts.
Usage: python lab/conv_halo_emulate.py     (a few seconds)"""
import numpy as np

PH, PW = 16, 32
HH, HW = PH + 2, PW + 2
HALO_CHUNKS = HH * HW * 4
HALO_PIECES = (HALO_CHUNKS + 63) // 64
HALO_BYTES = 40 * 1024
WS_BYTES = 128 * 64


def main():
    rng = np.random.default_rng(0)
    Hp, Wp = 2 * PH + 2, 2 * PW + 2             # a padded frame of 2 x 2 patches; the emulated tile is the lower right one
    y0, x0 = PH, PW
    X = rng.integers(-3, 4, size=(3, Hp, Wp, 128)).astype(np.float64)       # frames t + dt, padded coordinates, channels
    W = rng.integers(-2, 3, size=(128, 27, 128)).astype(np.float64)         # [filter][tap][channel]
    out = np.zeros((PH, PW, 128))
    lds_halo = np.full((2, HALO_BYTES // 16, 8), np.nan)                     # [buffer][16-byte chunk][8 channels]
    lds_w = np.full((4, WS_BYTES // 16, 8), np.nan)
    acc = np.zeros((8, 64, 8, 4, 4))                                         # [wave][lane][f][j][r]

    def issue_halo(stage, wid, k):
        dt, q = stage >> 2, stage & 3
        pc = min(k * 8 + wid, HALO_PIECES - 1)
        for lane in range(64):
            g = min(pc * 64 + lane, HALO_CHUNKS - 1)
            hp, slot = g >> 2, g & 3
            hy, hx = divmod(hp, HW)
            c = slot ^ ((hx >> 2) & 3)
            src = X[dt, y0 + hy, x0 + hx, q * 32 + c * 8:q * 32 + c * 8 + 8]
            lds_halo[stage & 1, pc * 64 + lane] = src                         # DMA: LDS chunk = piece base + lane

    def issue_w(step, wid):
        stage, k = divmod(step, 9)
        dt, q = stage >> 2, stage & 3
        tap = dt * 9 + k
        for lane in range(64):
            r, slot = 16 * wid + (lane >> 2), lane & 3
            wn_, j, i = r >> 6, (r >> 4) & 3, r & 15
            n = 64 * wn_ + 32 * (j >> 1) + 8 * (i >> 2) + 4 * (j & 1) + (i & 3)
            c = slot ^ ((r >> 2) & 3)
            lds_w[step & 3, wid * 64 + lane] = W[n, tap, q * 32 + c * 8:q * 32 + c * 8 + 8]

    conflicts = 0
    for stage in range(12):
        for wid in range(8):
            for k in range(5):
                issue_halo(stage, wid, k)      # (the kernel issues these one stage ahead; placement is what is checked here)
        for k in range(9):
            step = stage * 9 + k
            for wid in range(8):
                issue_w(step, wid)
            dh, dw = divmod(k, 3)
            for wid in range(8):
                wm, wn = wid & 3, wid >> 2
                fw = np.zeros((4, 64, 8))
                fx = np.zeros((8, 64, 8))
                for lane in range(64):
                    fi, fc = lane & 15, lane >> 4
                    woff = (64 * wn + fi) * 64 + ((fc ^ ((fi >> 2) & 3)) << 4)
                    for j in range(4):
                        fw[j, lane] = lds_w[step & 3, (woff + j * 16 * 64) // 16]
                    for f in range(8):
                        hx = 16 * (f & 1) + fi + dw
                        xoff = hx * 64 + ((fc ^ ((hx >> 2) & 3)) << 4)
                        hy = 4 * wm + (f >> 1) + dh
                        fx[f, lane] = lds_halo[stage & 1, (hy * HW * 64 + xoff) // 16]
                # bank groups of one 16-lane group (same fc): addresses mod 256 bytes must be 16 distinct multiples of 16
                for f in range(8):
                    for fc in range(4):
                        banks = set()
                        for fi in range(16):
                            hx = 16 * (f & 1) + fi + dw
                            a = (4 * wm + (f >> 1) + dh) * HW * 64 + hx * 64 + ((fc ^ ((hx >> 2) & 3)) << 4)
                            banks.add((a % 256) // 16)
                        conflicts += len(banks) != 16
                for j in range(4):               # the filter fragments
                    for fc in range(4):
                        banks = {(((64 * wn + fi) * 64 + ((fc ^ ((fi >> 2) & 3)) << 4) + j * 1024) % 256) // 16 for fi in range(16)}
                        conflicts += len(banks) != 16
                assert not np.isnan(fw).any() and not np.isnan(fx).any()
                # v_mfma_f32_16x16x32: first operand A[m = lane & 15][k = 8 (lane >> 4) ..], second B[k][n = lane & 15];
                # D[m = 4 (lane >> 4) + r][n = lane & 15] in register r
                for f in range(8):
                    for j in range(4):
                        A = np.zeros((16, 32))
                        B = np.zeros((32, 16))
                        for lane in range(64):
                            A[lane & 15, 8 * (lane >> 4):8 * (lane >> 4) + 8] = fw[j, lane]
                            B[8 * (lane >> 4):8 * (lane >> 4) + 8, lane & 15] = fx[f, lane]
                        D = A @ B
                        for lane in range(64):
                            for r in range(4):
                                acc[wid, lane, f, j, r] += D[4 * (lane >> 4) + r, lane & 15]
    # epilogue
    for wid in range(8):
        wm, wn = wid & 3, wid >> 2
        for lane in range(64):
            fi, fc = lane & 15, lane >> 4
            for hsel in range(2):
                n = 64 * wn + 32 * hsel + 8 * fc
                for f in range(8):
                    y, x = 4 * wm + (f >> 1), 16 * (f & 1) + fi
                    for r in range(4):
                        out[y, x, n + r] = acc[wid, lane, f, 2 * hsel, r]
                        out[y, x, n + 4 + r] = acc[wid, lane, f, 2 * hsel + 1, r]
    # direct convolution of the same tile
    ref = np.zeros((PH, PW, 128))
    for dt in range(3):
        for dh in range(3):
            for dw in range(3):
                patch = X[dt, y0 + dh:y0 + dh + PH, x0 + dw:x0 + dw + PW, :]          # [PH, PW, 128]
                ref += patch @ W[:, (dt * 3 + dh) * 3 + dw, :].T
    print("max |emulated - direct| =", np.abs(out - ref).max(), " (integers: must be 0)")
    print("fragment groups with an LDS bank conflict:", conflicts)
    assert np.array_equal(out, ref) and conflicts == 0
    # the counted waits: pieces a wave issues per step in program order = [filter s + 3][halo piece if tap < 5]
    def halo_issued(k): return 1 if 0 <= k < 5 else 0
    def wait_count(k): return halo_issued(k - 2) + 1 + halo_issued(k - 1) + 1 + halo_issued(k)
    order = []                                   # program order of one wave's DMA instructions
    order += [("h", 0, k) for k in range(5)] + [("w", 0), ("w", 1), ("w", 2)]
    pos_after_step = {}
    for step in range(108):
        stage, k = divmod(step, 9)
        if step + 3 < 108:
            order.append(("w", step + 3))
        if k < 5 and stage + 1 < 12:
            order.append(("h", stage + 1, k))
        pos_after_step[step] = len(order)
    for step in range(107):
        stage, k = divmod(step, 9)
        need = order.index(("w", step + 1))      # everything up to and including this must have landed
        allowed = 0 if stage == 11 else wait_count(k)
        outstanding_ok = pos_after_step[step] - (need + 1)       # pieces issued after the needed one
        assert allowed <= outstanding_ok or stage == 11, (step, allowed, outstanding_ok)
        assert allowed == outstanding_ok or stage == 11, (step, allowed, outstanding_ok)        # exact everywhere but the drained last stage
        if k == 8 and stage + 1 < 12:            # the next stage's halo is older than the needed filter piece
            assert max(order.index(("h", stage + 1, kk)) for kk in range(5)) < need
    print("counted waits: vmcnt(X) never exceeds the pieces issued after the one that is needed")


if __name__ == "__main__":
    main()
