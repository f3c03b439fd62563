"""Can a communication kernel run BESIDE the persistent GEMM?  (round-4 review, item 2; measured on ONE GPU)

The sequence-parallel single blocks queue the MLP-branch GEMM while the K|V|Q all-to-all is in flight (pyflow_hip/flux_sp.py:
attend(overlap=...); the reference blocks in trainer_misc/communicate.py:7-26).  gemm8p launches one workgroup per CU that owns
the CU whole (128 KiB LDS, the whole register file) and walks a STATIC tile list.  This tool times, with HIP events on one
device, for R = 0, 8, 16, 32 reserved CUs (pf_gemm_set_policy(2000 + R)):
   gemm alone | exchange alone | exchange enqueued first on its own stream, GEMM right behind it on the compute stream
   (the engine's order) | GEMM first, exchange enqueued behind it on the other stream (no data dependence)
for two "exchanges": (a) a world-1 pf_all_to_all_v self-exchange through RCCL (ncclSend / ncclRecv to self: RCCL's own
kernel, the thing that must find CUs), (b) a plain device-to-device copy of the same bytes (hipMemcpyAsync: a blit kernel /
SDMA, the best case), and for two GEMMs: the MLP branch at the full sequence (M = 2 x 15 488: 25 rounds) and at a rank's rows
for P = 8 (M = 2 x 1 936: 1.25 rounds).  "hidden" = (gemm alone + exchange alone - both) / exchange alone.
usage: [COMM_OVERLAP_R=0,64,128] [NCCL_MAX_NCHANNELS=4] python tools/comm_overlap_bench.py [iters]"""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pyramid-flow_amd"))
from pyflow_hip import ops                                                                    # noqa: E402
from pyflow_hip.comm_native import NativeComm, exchange_unique_id                            # noqa: E402

D = 1920
N1 = (4 * D * 2 // 3) // 256 * 256          # the MLP branch's first column group (flux_sp.py)


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    torch.cuda.set_device(0)
    comm = NativeComm(0, 1, exchange_unique_id(0, 1))
    g = torch.Generator(device="cuda").manual_seed(0)
    nbytes = 45 << 20                                         # one rank's K|V|Q exchange at the longest sequence
    send = torch.randn(nbytes // 2, device="cuda", generator=g).bfloat16()
    recv = torch.empty_like(send)
    side = torch.cuda.Stream()
    main_s = torch.cuda.current_stream()
    ws = torch.empty(16 << 20, dtype=torch.float32, device="cuda")

    def ev():
        return torch.cuda.Event(enable_timing=True)

    for M, label in ((15488, "full sequence (P = 1 rows)"), (1936, "a rank's rows at P = 8")):
        A = torch.randn(2, M, D, device="cuda", generator=g).bfloat16()
        W = (torch.randn(N1, D, device="cuda", generator=g) * 0.02).bfloat16()
        bias = torch.randn(N1, device="cuda", generator=g)
        out = torch.empty(2, M, N1, device="cuda", dtype=torch.bfloat16)

        def gemm():
            ops.gemm(A, W, out, M, N1, D, D, D, N1, bias=bias, batch=2, strideA=M * D, strideC=M * N1, gelu_from=0,
                     tail_workspace=ws)

        def xchg_rccl():
            # NativeComm orders its stream behind the CURRENT stream at call time: issue it from the side stream so that it does
            # not wait for a GEMM queued on the compute stream (the engine's exchange waits for the K|V|Q GEMM only)
            with torch.cuda.stream(side):
                h = comm.all_to_all(recv, send, [send.numel()], [send.numel()], async_op=True)
                h.wait()

        def xchg_copy():
            with torch.cuda.stream(side):
                recv.copy_(send, non_blocking=True)

        def timed(first, second):
            """wall time (ms) from a common start to the end of both streams' work"""
            t0, t1 = ev(), ev()
            torch.cuda.synchronize()
            t0.record(main_s)
            side.wait_event(t0)
            for fn in (first, second):
                if fn is not None:
                    fn()
            main_s.wait_stream(side)
            t1.record(main_s)
            torch.cuda.synchronize()
            return t0.elapsed_time(t1)

        for R in [int(x) for x in os.environ.get("COMM_OVERLAP_R", "0,8,16,32").split(",")]:
            ops.gemm_set_policy(2000 + R)
            wgs = ops.L.load().pf_gemm_workgroups()
            for name, x in (("RCCL self send/recv", xchg_rccl), ("device copy", xchg_copy)):
                rows = {k: [] for k in ("gemm", "xchg", "xchg_then_gemm", "gemm_then_xchg")}
                for fn in (gemm, x, gemm, x):
                    fn()
                torch.cuda.synchronize()
                for _ in range(iters):
                    rows["gemm"].append(timed(gemm, None))
                    rows["xchg"].append(timed(x, None))
                    rows["xchg_then_gemm"].append(timed(x, gemm))
                    rows["gemm_then_xchg"].append(timed(gemm, x))
                m = {k: statistics.median(v) for k, v in rows.items()}
                hid1 = (m["gemm"] + m["xchg"] - m["xchg_then_gemm"]) / m["xchg"]
                hid2 = (m["gemm"] + m["xchg"] - m["gemm_then_xchg"]) / m["xchg"]
                print(f"{label}: M=2x{M} N={N1} K={D} | reserved CUs {R:2d} ({wgs} workgroups) | {name:20s} | gemm {m['gemm']:.3f} ms  "
                      f"exchange ({nbytes >> 20} MiB) {m['xchg']:.3f} ms | exchange first, gemm behind it: {m['xchg_then_gemm']:.3f} ms "
                      f"(hidden {hid1:+.2f}) | gemm first: {m['gemm_then_xchg']:.3f} ms (hidden {hid2:+.2f})", flush=True)
        ops.gemm_set_policy(2000)
    comm.close()


if __name__ == "__main__":
    main()
