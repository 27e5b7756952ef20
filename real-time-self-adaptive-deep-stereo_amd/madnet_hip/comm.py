"""The shared-model collective through the C-ABI (include/madnet_hip.h: mh_comm_*, mh_allreduce_sum -- RCCL over xGMI, resolved by the library at run time).

SURVEY 8(e): streams with private models need no collective; streams of several GPUs that adapt ONE model sum their gradient buffers once per step.  The
all-reduce is a plan op (Recorder.allreduce_sum -> MH_OP_ALLREDUCE), so the shared-model step replays as ONE hipGraph with the collective inside it.
torch.distributed only carries the 128-byte unique id from rank 0 to the other ranks -- that is all the host layer needs from it on this path."""
import ctypes as C

import torch

from . import _ffi


class Comm(object):
    """One communicator per process (= per GPU).  world == 1 needs no process group (a 1-rank RCCL communicator: the collective is the identity, the launch
    pattern is the real one -- what a 1-GPU box can measure of the shared-model step)."""

    def __init__(self, lib, rank=0, world=1, dist=None, group=None, device=None, warm=True):
        self.lib, self.rank, self.world = lib, int(rank), int(world)
        if not lib.comm_available():
            raise RuntimeError("RCCL is not available to libmadnet_hip.so (librccl.so not found; MADNET_HIP_RCCL = full path)")
        ident = bytearray(_ffi.COMM_ID_BYTES)
        if self.rank == 0:
            buf = (C.c_char * _ffi.COMM_ID_BYTES).from_buffer(ident)
            lib.comm_unique_id(buf)
        if self.world > 1:
            if dist is None:
                raise ValueError("world > 1: the unique id travels through torch.distributed (pass dist / group)")
            box = [bytes(ident)]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            ident = bytearray(box[0])
        h = C.c_void_p()
        src = (C.c_char * _ffi.COMM_ID_BYTES).from_buffer(ident)
        lib.comm_init(src, self.rank, self.world, C.byref(h))          # collective: every rank arrives here
        self.handle = h
        r, w, v = C.c_int32(), C.c_int32(), C.c_int32()
        lib.comm_info(self.handle, C.byref(r), C.byref(w), C.byref(v))
        assert (r.value, w.value) == (self.rank, self.world)
        self.version = v.value
        self._warm = None
        if warm:
            # RCCL sets its channels up on the first collective of a communicator: outside any stream capture, once
            # (a small and a large message: the protocols RCCL picks by size are all set up before anything is captured)
            self._warm = torch.zeros(1 << 22, device=device if device is not None else "cuda")
            st = torch.cuda.current_stream(self._warm.device)
            for n in (256, 1 << 22):
                self.allreduce(lib, [(self._warm, 0, n)], stream=st.cuda_stream)
            self.allreduce(lib, [(self._warm, 0, 1 << 20), (self._warm, 1 << 20, 4)], stream=st.cuda_stream)
            st.synchronize()
            self._warm = None

    @property
    def version_string(self):
        v = self.version
        return "%d.%d.%d" % (v // 10000, (v // 100) % 100, v % 100) if v >= 10000 else str(v)

    def allreduce(self, target, ranges, stream=None):
        """in-place fp32 sum over the ranks of [(tensor, offset, count)] (<= 8 ranges, ONE RCCL group).  target: the library (launch on `stream`) or a Recorder
        (the op is recorded on the recorder's current lane)."""
        n = len(ranges)
        assert 1 <= n <= _ffi.ALLREDUCE_MAX_BUFS
        bufs = (C.c_void_p * n)(); counts = (C.c_int64 * n)()
        for k, (t, off, cnt) in enumerate(ranges):
            assert t.dtype == torch.float32 and t.is_contiguous() and 0 <= off and off + cnt <= t.numel() and 0 < cnt < 2 ** 31
            bufs[k] = t.data_ptr() + 4 * off
            counts[k] = cnt
        target.allreduce_sum(bufs, counts, n, self.handle, C.c_void_p(stream) if stream is not None else None)

    def close(self):
        if self.handle is not None and self.handle.value:
            self.lib.comm_destroy(self.handle)
        self.handle = None
