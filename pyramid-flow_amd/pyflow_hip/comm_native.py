"""`NativeComm`: the SPComm interface (pyflow_hip/sp.py) on the C-ABI communicator (pf_comm_*, csrc/comm.hip) instead of
torch.distributed -- RCCL driven directly, collectives on the communicator's own HIP stream, ordering by events.

The default client of the multi-GPU paths remains torch.distributed (backend "nccl" is the same RCCL); this class is the
proof that a host WITHOUT torch.distributed can drive them through the C ABI alone, and an opt-in for the Python host
(`init_sequence_parallel_group(args, native=True)`).  Bootstrap: rank 0 draws the 128-byte unique id
(pf_comm_unique_id) and publishes it -- here through a file or an existing torch.distributed store; any byte channel
works.
"""
import ctypes as C
import os
import time

import torch

from . import lib as L
from . import ops
from .lib import check
from .sp import starts_of


def exchange_unique_id(rank, world, path=None):
    """128-byte RCCL unique id from rank 0 to everyone: through torch.distributed when a process group exists,
    otherwise through the file `path` (rank 0 writes it atomically, the others poll)."""
    lib = L.load()
    buf = (C.c_char * 128)()
    if world == 1:
        check(lib.pf_comm_unique_id(buf))
        return bytes(buf)
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        # 128 id bytes + one "rank 0 has an id" byte: when rank 0 cannot create the id (no librccl, ...) the other ranks are
        # already waiting in this broadcast -- they are told, and EVERY rank raises, instead of rank 0 raising alone and the
        # rest hanging until the process group's timeout
        t = torch.zeros(129, dtype=torch.uint8)
        err = None
        if rank == 0:
            try:
                check(lib.pf_comm_unique_id(buf))
                t[:128] = torch.frombuffer(bytearray(bytes(buf)), dtype=torch.uint8)
                t[128] = 1
            except Exception as e:          # noqa: BLE001
                err = e
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.broadcast(t, 0)
        t = t.cpu()
        if err is not None:
            raise err
        if int(t[128]) != 1:
            raise RuntimeError("exchange_unique_id: rank 0 could not create an RCCL unique id")
        return bytes(t[:128].numpy().tobytes())
    assert path is not None, "no torch.distributed group: pass a file path every rank can read"
    if rank == 0:
        check(lib.pf_comm_unique_id(buf))
        tmp = path + ".tmp"
        with open(tmp, "wb") as f:
            f.write(bytes(buf))
        os.replace(tmp, path)
        return bytes(buf)
    deadline = time.time() + 300
    while not os.path.exists(path):
        if time.time() > deadline:
            raise TimeoutError(f"unique id file {path} never appeared")
        time.sleep(0.05)
    with open(path, "rb") as f:
        return f.read()


class NativeComm:
    native = True
    backend = "pf_comm"
    recordable = True          # all_to_all / wait have launch-list entries (pf_cmdlist_all_to_all_v / pf_cmdlist_comm_wait)

    def __init__(self, rank, world, unique_id):
        """unique_id = None: a communicator WITHOUT RCCL (pf_comm_init_local) -- usable once windows are attached
        (attach_windows: the copy-engine transport), for chunks that fit a window slot"""
        lib = L.load()
        self._lib = lib
        self.rank, self.world = rank, world
        self.group = None
        h = C.c_void_p()
        if unique_id is None:
            check(lib.pf_comm_init_local(C.byref(h), C.c_int(rank), C.c_int(world)))
        else:
            check(lib.pf_comm_init(C.byref(h), C.c_int(rank), C.c_int(world), C.c_char_p(unique_id)))
        self._h = h

    def attach_windows(self, slot_bytes, gather_handles):
        """switch the v-collectives (all_to_all, shift / halo pass, all_gather_v, send / recv) to the COPY-ENGINE transport
        (csrc/comm.hip: IPC-mapped exchange windows, device-to-device copies, stream-ordered flags; no kernel, no CU) for
        chunks of at most `slot_bytes` (a multiple of 256).  gather_handles(my64: bytes) -> [bytes] * world: the host's byte
        channel (e.g. a torch.distributed all_gather_object, or files)."""
        buf = (C.c_char * 64)()
        check(self._lib.pf_comm_create_window(self._h, C.c_longlong(slot_bytes), buf))
        handles = gather_handles(bytes(buf))
        assert len(handles) == self.world and all(len(h_) == 64 for h_ in handles)
        check(self._lib.pf_comm_attach_windows(self._h, C.c_char_p(b"".join(handles))))
        self.transport = "windows"
        return self

    transport = "rccl"

    def close(self):
        if self._h:
            self._lib.pf_comm_destroy(self._h)
            self._h = None

    @staticmethod
    def _cur():
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    class _Handle:
        def __init__(self, comm):
            self.comm = comm

        def wait(self):
            """the CURRENT stream waits for the exchange on the device (no host sync), like a c10d work handle.
            While a launch list is being recorded the wait becomes a list entry on the recorder's current stream slot."""
            rec = ops.RECORDER
            if rec is not None:
                check(self.comm._lib.pf_cmdlist_comm_wait(rec.h, self.comm._h, C.c_int(rec.slot)))
                return
            check(self.comm._lib.pf_comm_wait(self.comm._h, NativeComm._cur()))

    def _no_recording(self, what):
        # only the Ulysses exchange + its wait have list entries (pf_cmdlist_all_to_all_v / pf_cmdlist_comm_wait): any
        # other collective issued while recording would run NOW, out of order with the recorded kernels
        assert ops.RECORDER is None, f"NativeComm.{what} cannot be recorded into a launch list"

    def all_to_all(self, recv, send, recv_splits, send_splits, async_op=False):
        esz = send.element_size()
        P = self.world
        arr = C.c_longlong * P
        so, ro = starts_of(send_splits), starts_of(recv_splits)
        a = (C.c_void_p(send.data_ptr()), arr(*[s * esz for s in send_splits]), arr(*[o * esz for o in so]),
             C.c_void_p(recv.data_ptr()), arr(*[r * esz for r in recv_splits]), arr(*[o * esz for o in ro]))
        rec = ops.RECORDER
        if rec is not None:      # recorded: ordered after what the list holds so far on the current slot, replayed from C
            check(self._lib.pf_cmdlist_all_to_all_v(rec.h, self._h, *a, C.c_int(rec.slot)))
        else:
            check(self._lib.pf_all_to_all_v(self._h, *a, self._cur()))
        h = self._Handle(self)
        if async_op:
            return h
        h.wait()
        return None

    def all_reduce(self, t):
        self._no_recording("all_reduce")
        assert t.dtype == torch.float32 and t.is_contiguous()
        check(self._lib.pf_all_reduce_sum_f32(self._h, C.c_void_p(t.data_ptr()), C.c_longlong(t.numel()), self._cur()))
        self._Handle(self).wait()
        return t

    def broadcast(self, t, src=0):
        self._no_recording("broadcast")
        assert t.is_contiguous()
        check(self._lib.pf_broadcast_bytes(self._h, C.c_void_p(t.data_ptr()), C.c_longlong(t.numel() * t.element_size()),
                                           C.c_int(src), self._cur()))
        self._Handle(self).wait()
        return t

    def shift(self, send_t, recv_t):
        self._no_recording("shift")
        s_ = send_t.contiguous()
        assert recv_t.is_contiguous()
        check(self._lib.pf_halo_send_recv(self._h, C.c_void_p(s_.data_ptr()), C.c_void_p(recv_t.data_ptr()),
                                          C.c_longlong(s_.numel() * s_.element_size()), self._cur()))
        self._Handle(self).wait()

    def shift_start(self, send_t, recv_t):
        """the halo pass queued on the communicator's stream; wait() of the returned handle orders the current stream
        behind it (pf_comm_wait) -- kernels launched in between run while the frames travel"""
        self._no_recording("shift")
        s_ = send_t.contiguous()
        assert recv_t.is_contiguous()
        check(self._lib.pf_halo_send_recv(self._h, C.c_void_p(s_.data_ptr()), C.c_void_p(recv_t.data_ptr()),
                                          C.c_longlong(s_.numel() * s_.element_size()), self._cur()))
        return self._Handle(self)

    def all_gather_v(self, send, recv, counts):
        """recv <- concatenation of every rank's `send` (element counts per rank in `counts`)"""
        self._no_recording("all_gather_v")
        esz = send.element_size()
        arr = C.c_longlong * self.world
        check(self._lib.pf_all_gather_v(self._h, C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()),
                                        arr(*[c * esz for c in counts]), arr(*[o * esz for o in starts_of(counts)]),
                                        self._cur()))
        self._Handle(self).wait()
        return recv

    def selftest(self, device):
        """every collective of the sampling path once with known values (SPComm.selftest over this transport)"""
        from .sp import SPComm
        return SPComm.selftest(self, device)

    def warm_p2p(self, device):
        from .sp import SPComm
        return SPComm.warm_p2p(self, device)

    def barrier(self):
        t = torch.zeros(1, dtype=torch.float32, device="cuda")
        self.all_reduce(t)
        torch.cuda.current_stream().synchronize()

    # point-to-point of the tile-parallel decode: expressed through the v-collectives (a send is an all-to-all with one
    # non-zero count); the pair must call send / recv in matching order, like c10d's send / recv
    def send(self, t, dst):
        s_ = t.contiguous()
        n = s_.numel()
        zeros = [0] * self.world
        ss = list(zeros)
        ss[dst] = n
        self.all_to_all_pair(s_, None, ss, zeros, peer=dst)

    def recv(self, t, src):
        zeros = [0] * self.world
        rs = list(zeros)
        rs[src] = t.numel()
        self.all_to_all_pair(None, t, zeros, rs, peer=src)
        return t

    def all_to_all_pair(self, send, recv, send_splits, recv_splits, peer):
        """two-rank exchange (only `peer` and this rank take part: grouped ncclSend / ncclRecv involve just the pair)"""
        self._no_recording("send / recv")
        ref = send if send is not None else recv
        esz = ref.element_size()
        arr = C.c_longlong * self.world
        zero = arr(*([0] * self.world))
        check(self._lib.pf_all_to_all_v(self._h, C.c_void_p(send.data_ptr() if send is not None else 0),
                                        arr(*[s * esz for s in send_splits]), zero,
                                        C.c_void_p(recv.data_ptr() if recv is not None else 0),
                                        arr(*[r * esz for r in recv_splits]), zero, self._cur()))
        self._Handle(self).wait()
