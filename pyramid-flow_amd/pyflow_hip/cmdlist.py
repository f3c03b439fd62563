"""Launch lists (pf_cmdlist_*, csrc/cmdlist.hip): the kernel sequence of one transformer forward recorded once per
(unit, stage) and re-issued from C -- as a replay through the same entry points, or as one hipGraph launch.

While a list is being recorded (`with recording(cl):`) the recordable wrappers of `ops` (gemm, attention, ln_modulate,
qk_norm_rope, v_transpose, copy_rows, sp_relayout) and of `NativeComm` (all_to_all, wait) append their descriptors to
it instead of launching; cross-stream joins become list entries (`cl.join`).  Nothing computes at record time."""
import contextlib
import ctypes as C

import torch

from . import lib as L
from . import ops
from .lib import check


class CommandList:
    def __init__(self):
        self._lib = L.load()
        self.h = C.c_void_p(self._lib.pf_cmdlist_create())
        self.slot = 0                # stream slot the next recorded entry goes to: 0 = compute stream, 1 = side stream
        self.runs = 0
        self._done = None            # event after the latest run (a list must outlive its work)

    def close(self):
        if self.h:
            if self._done is not None:
                self._done.synchronize()
            self._lib.pf_cmdlist_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return int(self._lib.pf_cmdlist_size(self.h))

    @property
    def is_graph(self):
        return bool(self._lib.pf_cmdlist_is_graph(self.h))

    def join(self, from_slot, to_slot):
        """everything recorded so far on stream slot `from_slot` happens before what is recorded next on `to_slot`"""
        check(self._lib.pf_cmdlist_join(self.h, C.c_int(from_slot), C.c_int(to_slot)))

    @contextlib.contextmanager
    def on_slot(self, slot):
        prev, self.slot = self.slot, slot
        try:
            yield
        finally:
            self.slot = prev

    @staticmethod
    def _sp(stream):
        return C.c_void_p(stream.cuda_stream)

    def run(self, main=None, side=None):
        main = main if main is not None else torch.cuda.current_stream()
        side = side if side is not None else main
        check(self._lib.pf_cmdlist_run(self.h, self._sp(main), self._sp(side)))
        self.runs += 1
        if self._done is None:
            self._done = torch.cuda.Event()
        self._done.record(main)

    def instantiate(self, main=None, side=None):
        """capture the replay into a hipGraph (lists without communicator entries); run() is then one graph launch"""
        main = main if main is not None else torch.cuda.current_stream()
        side = side if side is not None else main
        check(self._lib.pf_cmdlist_instantiate(self.h, self._sp(main), self._sp(side)))


@contextlib.contextmanager
def recording(cl):
    assert ops.RECORDER is None, "launch lists do not nest"
    ops.RECORDER = cl
    try:
        yield cl
    finally:
        ops.RECORDER = None
