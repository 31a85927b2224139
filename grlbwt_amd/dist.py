"""Collection-level multi-GPU driver: one process per GPU, torch.distributed transport.

The collection is sharded by record (contiguous ranges of whole strings, rank order =
collection order).  All orchestration lives in the C++ engine (grlbwt_dist_build); this
module only provides the two collectives the engine calls back for, implemented with
torch.distributed -- backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
"""
import ctypes as C
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from . import engine

_AG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64)
_A2A = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_uint64),
                   C.POINTER(C.c_uint64))


class CommStruct(C.Structure):
    _fields_ = [("rank", C.c_int), ("size", C.c_int), ("user", C.c_void_p), ("allgather", _AG), ("alltoallv", _A2A),
                ("flags", C.c_uint32)]


COMM_STREAM_ORDERED = 1
COMM_KEEP_PARTS = 2          # the image stays in parts on the ranks that induced them (include/grlbwt_hip.h)


class _DevView:
    """Zero-copy view of engine-owned device memory for torch (CUDA array interface)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def _view(ptr, nbytes, device):
    if nbytes == 0:
        return torch.empty(0, dtype=torch.uint8, device=device)
    if device.type == "cuda":
        return torch.as_tensor(_DevView(ptr, nbytes), device=device)
    buf = (C.c_uint8 * nbytes).from_address(ptr)
    return torch.from_numpy(np.frombuffer(buf, dtype=np.uint8))


class Communicator:
    """Callbacks handed to the engine (must stay alive while the engine uses them)."""

    def __init__(self, device, group=None):
        self.device = torch.device(device)
        self.group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        self.bytes_moved = 0
        self.n_allgather = 0
        self.n_alltoall = 0
        self.log = None              # set to a list to record (kind, bytes, seconds) per call
        self.seconds = 0.0           # wall time spent inside the callbacks (transport + synchronisation)
        # gloo transport with device-resident engine buffers (tests on a single-GPU box): stage through the host
        self.stage = self.device.type == "cuda" and dist.get_backend(group) == "gloo"
        # RCCL: run the engine on a torch stream and issue the collectives with that stream current, so that they
        # are ordered on it like the engine's own kernels and nobody waits on the host (GRLBWT_DIST_SYNC_COLLECTIVES=1
        # restores a full synchronisation around every exchange)
        self.stream = None
        if self.device.type == "cuda" and not self.stage and os.environ.get("GRLBWT_DIST_SYNC_COLLECTIVES") != "1":
            self.stream = torch.cuda.Stream(self.device)
        self._ag = _AG(self._allgather)
        self._a2a = _A2A(self._alltoallv)
        self.struct = CommStruct(self.rank, self.size, None, self._ag, self._a2a,
                                 COMM_STREAM_ORDERED if self.stream is not None else 0)

    def attach(self, ctx):
        """Make `ctx` run on the stream the collectives are issued on (stream-ordered mode)."""
        if self.stream is not None and getattr(ctx, "_comm_stream", None) is not self.stream:
            ctx.set_stream(self.stream.cuda_stream)
            ctx._comm_stream = self.stream

    def _sync(self):
        if self.device.type == "cuda" and self.stream is None:
            torch.cuda.synchronize(self.device)

    def _allgather(self, user, send, recv, nbytes):
        t0 = time.perf_counter()
        try:
            s = _view(send, nbytes, self.device)
            r = _view(recv, nbytes * self.size, self.device)
            if self.stage:
                hs, hr = s.cpu(), torch.empty(nbytes * self.size, dtype=torch.uint8)
                dist.all_gather_into_tensor(hr, hs, group=self.group)
                r.copy_(hr)
            elif self.stream is not None:
                with torch.cuda.stream(self.stream):
                    dist.all_gather_into_tensor(r, s, group=self.group)  # the stream waits for it, the host does not
            else:
                dist.all_gather_into_tensor(r, s, group=self.group)      # also at size 1: same code path as N > 1
            self._sync()
            self.bytes_moved += nbytes * self.size
            self.n_allgather += 1
            self.seconds += time.perf_counter() - t0
            if self.log is not None:
                self.log.append(("allgather", nbytes * self.size, time.perf_counter() - t0))
            return 0
        except Exception as e:          # never let an exception cross the C boundary
            print("grlbwt allgather callback failed:", repr(e), flush=True)
            return 1

    def _alltoallv(self, user, send, send_bytes, send_off, recv, recv_bytes, recv_off):
        """MPI_Alltoallv shape: block g = send_bytes[g] bytes at send + send_off[g] (the engine keeps blocks <= 256 MiB)."""
        t0 = time.perf_counter()
        try:
            n = self.size
            send, recv = send or 0, recv or 0
            sb = [int(send_bytes[i]) for i in range(n)]
            rb = [int(recv_bytes[i]) for i in range(n)]
            so = [int(send_off[i]) for i in range(n)]
            ro = [int(recv_off[i]) for i in range(n)]
            ins = [_view(send + so[g], sb[g], self.device) for g in range(n)]
            outs = [_view(recv + ro[g], rb[g], self.device) for g in range(n)]
            if self.stage or self.device.type == "cpu":
                # gloo has the single-buffer form only: pack on the host, exchange, unpack
                hs = torch.cat([t.cpu() for t in ins]) if sum(sb) else torch.empty(0, dtype=torch.uint8)
                hr = torch.empty(sum(rb), dtype=torch.uint8)
                dist.all_to_all_single(hr, hs, output_split_sizes=rb, input_split_sizes=sb, group=self.group)
                pos = 0
                for g in range(n):
                    if rb[g]:
                        outs[g].copy_(hr[pos:pos + rb[g]])
                    pos += rb[g]
            elif self.stream is not None:
                with torch.cuda.stream(self.stream):
                    dist.all_to_all(outs, ins, group=self.group)             # grouped send/recv on the views
            else:
                dist.all_to_all(outs, ins, group=self.group)
            self._sync()
            self.bytes_moved += sum(sb)
            self.n_alltoall += 1
            self.seconds += time.perf_counter() - t0
            if self.log is not None:
                self.log.append(("alltoallv", sum(sb), time.perf_counter() - t0))
            return 0
        except Exception as e:
            print("grlbwt alltoallv callback failed:", repr(e), flush=True)
            return 1


def pool_flags(comm):
    """Context flags for a build whose buffers are handed to the communication library: with RCCL the engine takes its
    device memory from plain hipMalloc slabs (engine.FLAG_CLASSIC_POOL), not from the on-demand virtual-memory arena."""
    return engine.FLAG_CLASSIC_POOL if (comm.device.type == "cuda" and not comm.stage) else 0


def shard_records(cells, rank, size, sep=None):
    """Contiguous record range of `cells` (numpy array of whole strings) for `rank`: the reference's
    thread_ranges split (parsing_strategies.h:208-214), by string count."""
    cells = np.asarray(cells)
    sep = cells[-1] if sep is None else sep
    ends = np.flatnonzero(cells == sep)
    n_str = len(ends)
    if n_str < size:       # every rank computes the same verdict: nobody is left waiting in a collective
        raise ValueError("%d strings cannot be sharded over %d ranks (a rank would hold no record)" % (n_str, size))
    lo_s = (n_str * rank) // size
    hi_s = (n_str * (rank + 1)) // size
    lo = 0 if lo_s == 0 else int(ends[lo_s - 1]) + 1
    hi = int(ends[hi_s - 1]) + 1 if hi_s > 0 else 0
    return cells[lo:hi]


def agree_or_raise(err, device, group=None):
    """All ranks learn whether any rank failed a local step; every rank then raises (its own error, or a notice that a
    peer failed) before the next collective."""
    bad = torch.tensor([1 if err is not None else 0], dtype=torch.int32, device=device)
    if dist.get_world_size(group) > 1:
        dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=group)
    if int(bad.item()):
        if err is not None:
            raise err
        raise engine.GrlbwtError(-22, "another rank rejected its shard")


def dist_build(ctx, comm, keep_parts=False):
    """Run the collection-level build on a context that already holds this rank's shard.  keep_parts: the image is not
    gathered; every rank keeps its part (ctx.result_part(), ctx.write_part(path))."""
    L = ctx.L
    comm.attach(ctx)
    comm.struct.flags = (comm.struct.flags & ~COMM_KEEP_PARTS) | (COMM_KEEP_PARTS if keep_parts else 0)
    L.grlbwt_dist_build.argtypes = [C.c_void_p, C.POINTER(CommStruct)]
    ctx._ck(L.grlbwt_dist_build(ctx._h, C.byref(comm.struct)))


def grl_bwt_algo_sharded(shard, cell_bytes=1, device="cpu", lib=None, flags=0, group=None, comm_out=None, keep_parts=False,
                         part_file=None):
    """BCR BWT (.rl_bwt bytes) of the whole collection whose rank-th record shard is `shard`.
    Every rank returns the same bytes -- or, with keep_parts, (offset, part bytes, image bytes): its part of the image, which
    it also writes at that offset of `part_file` if one is named (the caller removes a stale file first; the file is complete
    once every rank has returned)."""
    dev = torch.device(device)
    comm = Communicator(dev, group)
    if comm_out is not None:
        comm_out.append(comm)
    n_local = torch.tensor([len(shard) // cell_bytes if isinstance(shard, (bytes, bytearray)) else int(np.asarray(shard).size)],
                           dtype=torch.int64, device=dev)
    if comm.size > 1:
        dist.all_reduce(n_local, group=group)
    if int(n_local.item()) >= 0xFFFFFF00:
        flags |= engine.FLAG_FORCE_IDX64           # collection-wide positions/frequencies need 64 bits
    index = dev.index if dev.type == "cuda" and dev.index is not None else 0
    with engine.Context(index, flags | pool_flags(comm), lib) as ctx:
        # the shard-local checks of the upload (ill-formed shard, empty shard) must not leave the other ranks waiting in
        # the first collective of the build: the verdict is agreed on first
        err = None
        try:
            ctx.upload(shard, cell_bytes)
        except engine.GrlbwtError as e:
            err = e
        agree_or_raise(err, dev, group)
        dist_build(ctx, comm, keep_parts)
        if keep_parts:
            if part_file is not None:
                ctx.write_part(part_file)
            return ctx.result_part()[0], ctx.result_bytes(), ctx.result_size()[0]
        return ctx.result_bytes()
