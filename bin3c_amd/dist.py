"""
Row-sharded multi-GPU execution of the hot path: one process per GPU (launched by torch.distributed.run).  Every rank
holds the raw count map (the input); from the permutation on it forms and works on its ROW BLOCK of the locality-ordered
map only (b3c_mat_permute_rows; include/b3c.h "multi-GPU").  The blocks are contiguous row ranges
balanced by STORED ENTRIES (:func:`balanced_starts`: a prefix sum over the ordered row lengths), not by row
count: one genome may hold a third of the contigs and hub contigs carry 10^5..10^6 contacts.

Transport.  On real multi-GPU runs (torch.distributed backend "nccl") the session creates an RCCL communicator INSIDE
libb3c (b3c_dist_comm_create; the 128-byte unique id travels over torch.distributed once) and libb3c then issues every
collective itself on its stream -- ncclAllGather over xGMI, no Python in the loop.  With any other backend (gloo: the
CPU / one-GPU tests) libb3c calls back into :meth:`DistSession._allgather`, which moves the same buffers with
torch.distributed.

    KR, per SpMV     ONE exchange (a grouped launch): the rank's partial of p.w | rk.rk together with its slice of w (inner
                     iteration) or of v = x*(A x) (outer iteration); the vector steps on both sides of the SpMV run over all
                     rows on every rank from the gathered slices, so nothing else travels
    Louvain level 0  all-gather of the decided labels per sub-round (4 n bytes), one integer partial sum per modularity
                     evaluation; moves are applied by every rank

The reference has no counterpart (single process); this is north_star's "sharded by contig-row
range with RCCL all-reduce over xGMI for the KR column sums and community-weight vectors".  xGMI is point to point
(7 links per GPU), the vector exchange is 1-8 MB per rank and the scalar ones are latency bound, so every exchange is
ONE launch and nothing is exchanged that can be recomputed.
"""
import ctypes as C
import os

import numpy as np

from . import _native

BUF_VEC, BUF_SCAL, BUF_LABELS, BUF_ISUM, BUF_SCAL_VEC = 0, 1, 2, 3, 4


B3CDist = _native.B3CDist      # b3c_dist, include/b3c.h (layout checked against the library at load time)


ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64)
ALLGATHERV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_int)


def shard_range(n, rank, world, align=64):
    """Rows of `rank`: equal chunks (aligned so that slices stay 512-byte aligned); the last ranks may be short or empty."""
    chunk = -(-n // world)
    chunk = -(-chunk // align) * align
    r0 = min(n, rank * chunk)
    return r0, min(n, r0 + chunk), chunk


def balanced_starts(row_nnz, world, align=64):
    """Row ranges of `world` ranks over rows with `row_nnz` stored entries each (numpy or torch, in the order the ranks will
    work in), balanced by entries: range r ends where the running sum first reaches (r + 1) / world of the total, rounded to a
    multiple of `align` rows.  -> numpy int64[world + 1], ascending, starts[0] = 0, starts[world] = n.  Deterministic: every rank
    computes the same ranges from the same lengths ("balanced by nnz (prefix-sum over indptr)")."""
    import torch
    t = torch.as_tensor(row_nnz).to(torch.int64)
    n = int(t.shape[0])
    cs = torch.cumsum(t, 0)
    total = int(cs[-1].item()) if n else 0
    starts = np.zeros(world + 1, dtype=np.int64)
    starts[world] = n
    if n and total > 0:
        targets = torch.as_tensor([(total * (r + 1)) // world for r in range(world - 1)], dtype=torch.int64, device=cs.device)
        cut = (torch.searchsorted(cs, targets, right=False) + 1).cpu().numpy() if world > 1 else np.zeros(0, dtype=np.int64)
        cut = np.minimum(n, ((cut + align - 1) // align) * align)
        starts[1:world] = np.maximum.accumulate(cut)
    else:
        chunk = -(-(-(-n // world)) // align) * align          # no entries at all: equal row counts, aligned like shard_range
        starts[1:world] = np.minimum(n, np.arange(1, world) * chunk)
    return starts


def allgather_ragged(view, starts, rank, group=None):
    """Ragged in-place all-gather of a row-indexed tensor: rank r owns view[starts[r]:starts[r+1]].  Done as a sum with zeros on
    an INTEGER view (every element has exactly one non-zero contributor, so the all-reduce IS the gather, bit for bit, and raw
    64-bit payloads survive); gloo and nccl alike."""
    import torch
    import torch.distributed as dist
    v = view.view(torch.int64 if view.element_size() == 8 else torch.int32)
    a, b = int(starts[rank]), int(starts[rank + 1])
    keep = v[a:b].clone()
    v.zero_()
    v[a:b] = keep
    dist.all_reduce(v, group=group)


def allgather_inplace(full, rank, world, group=None):
    """All-gather `full` (world equal slices, this rank's already in place).  One collective."""
    import torch
    import torch.distributed as dist
    cnt = full.numel() // world
    mine = full[rank * cnt:(rank + 1) * cnt]
    backend = dist.get_backend(group)
    if backend == 'nccl':
        # in place: the send buffer is this rank's slice of the receive buffer (NCCL/RCCL define this case -- sendbuff ==
        # recvbuff + rank * count -- and torch's FSDP all-gathers the same way); saves a copy kernel per exchange
        if os.environ.get('B3C_DIST_COPY_SEND'):
            mine = mine.clone()
        dist.all_gather_into_tensor(full, mine, group=group)
    elif full.is_cuda:
        # gloo moves CUDA tensors only for all_reduce/broadcast (single-GPU CI): gather = sum with zeros -- on an INTEGER
        # view, because libb3c sends raw 64-bit words through the float64 buffers (symmetry checksums, packed row-best
        # words, degrees) and a floating-point add would quieten a payload that happens to be a signalling NaN
        v = full.view(torch.int64 if full.element_size() == 8 else torch.int32)
        keep = v[rank * cnt:(rank + 1) * cnt].clone()
        v.zero_()
        v[rank * cnt:(rank + 1) * cnt] = keep
        dist.all_reduce(v, group=group)
    else:
        dist.all_gather(list(full.view(world, cnt).unbind(0)), mine.clone(), group=group)


class DistSession(object):
    """Owns the exchange buffers of one context and installs them with b3c_ctx_set_dist."""

    def __init__(self, ctx, n, group=None):
        import torch
        import torch.distributed as dist
        assert dist.is_initialized(), 'torch.distributed must be initialised (torch.distributed.run launches bench.py)'
        self.ctx = ctx
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.n = n
        self.r0, self.r1, self.chunk = shard_range(n, self.rank, self.world)
        dev = torch.device('cuda', ctx.device)
        tot = self.chunk * self.world
        self.vec = torch.zeros(tot, dtype=torch.float64, device=dev)
        self.scal = torch.zeros(4 * self.world, dtype=torch.float64, device=dev)
        self.labels = torch.zeros(tot, dtype=torch.int32, device=dev)
        self.isum = torch.zeros(2 * self.world, dtype=torch.int64, device=dev)
        self._bufs = {BUF_VEC: self.vec, BUF_SCAL: self.scal, BUF_LABELS: self.labels, BUF_ISUM: self.isum}
        self._error = None
        self._cb = ALLGATHER_FN(self._allgather)          # keep a reference: ctypes callbacks are not owned by C
        self._cb2 = ALLREDUCE_FN(self._allreduce)
        self._cb3 = ALLGATHERV_FN(self._allgatherv)
        self.starts = None                                # ragged row ranges (set_starts); None: equal chunks
        lib = ctx.lib
        # RCCL inside libb3c when the job runs on the nccl backend (or on request); the callback transport otherwise
        want = os.environ.get('B3C_DIST_TRANSPORT', 'auto')
        use_rccl = 'rccl' in want or (want in ('auto', 'direct') and dist.get_backend(group) == 'nccl')
        self.comm = None
        if use_rccl:
            ident = np.zeros(128, dtype=np.uint8)
            if self.rank == 0:
                ctx._check(lib.b3c_dist_unique_id(ident.ctypes.data_as(C.c_void_p)))
            t = torch.from_numpy(ident)
            if dist.get_backend(group) == 'nccl':
                t = t.to(dev)
                dist.broadcast(t, 0, group=group)
                ident = t.cpu().numpy()
            else:
                dist.broadcast(t, 0, group=group)
            comm = C.c_void_p()
            ctx._check(lib.b3c_dist_comm_create(ctx.h, self.world, self.rank, ident.ctypes.data_as(C.c_void_p), C.byref(comm)))
            self.comm = comm
        self.transport = 'rccl (in libb3c)' if self.comm else 'callback ({})'.format(dist.get_backend(group))
        self.direct = False
        self._x0 = int(lib.b3c_ctx_exchanges(ctx.h))
        torch.cuda.synchronize(dev)                       # the buffers were zero-filled on torch's stream, libb3c writes them on its own
        self._install(None)
        # Ragged row ranges need a transport that moves unequal slices (a group of broadcasts on RCCL).  One small exchange on
        # deliberately uneven ranges tells whether it does on THIS stack; if not, the session keeps equal row ranges (the plain
        # in-place all-gather of rounds 1-3) instead of failing the run.
        self.ragged_ok = self.world == 1 or self._selftest_ragged(dev)
        torch.cuda.synchronize(dev)
        self._x0 = int(lib.b3c_ctx_exchanges(ctx.h))       # (the self-test's exchange is not the caller's)
        # Direct transport (B3C_DIST_TRANSPORT=direct[+rccl]): the all-gathers as peer writes into IPC-mapped windows (dist.hip);
        # all-reduces and the aggregation's all-to-all stay on the transport chosen above
        if 'direct' in want and self.world > 1:
            self._open_window(dev)
        ctx.dist_session = self                           # pipeline.run_hot_path shards the map over the ranks when set

    def _install(self, starts):
        """(re)install the sharding description; `starts`: int64[world + 1] row ranges or None for equal chunks"""
        self.starts = None if starts is None else np.ascontiguousarray(starts, dtype=np.int64)
        sp = self.starts.ctypes.data_as(C.c_void_p) if self.starts is not None else None
        d = B3CDist(self.rank, self.world, self.chunk, self.vec.data_ptr(), self.scal.data_ptr(),
                    self.labels.data_ptr(), self.isum.data_ptr(), C.cast(self._cb, C.c_void_p), None,
                    self.comm.value if self.comm else None, C.cast(self._cb2, C.c_void_p), sp, C.cast(self._cb3, C.c_void_p))
        self._struct = d
        self.ctx._check(self.ctx.lib.b3c_ctx_set_dist(self.ctx.h, C.byref(d)))
        if self.starts is None:
            self.r0, self.r1, _ = shard_range(self.n, self.rank, self.world)
        else:
            self.r0, self.r1 = int(self.starts[self.rank]), int(self.starts[self.rank + 1])

    def _selftest_ragged(self, dev):
        import torch
        import torch.distributed as dist
        n, w = self.n, self.world
        st = np.array([n * r * (r + 1) // (w * (w + 1)) for r in range(w + 1)], dtype=np.int64)      # rank r holds r + 1 shares
        ok = 1
        try:
            self._install(st)
            v = torch.full((n,), -1, dtype=torch.int32, device=dev)
            v[int(st[self.rank]):int(st[self.rank + 1])] = self.rank
            torch.cuda.synchronize(dev)
            self.ctx.dist_allgather_i32(v)
            want = torch.repeat_interleave(torch.arange(w, dtype=torch.int32, device=dev), torch.as_tensor(np.diff(st), device=dev))
            ok = int(torch.equal(v, want))
        except Exception:
            ok = 0
        finally:
            try:
                self._install(None)
            except Exception:
                ok = 0
        red_dev = dev if dist.get_backend(self.group) == 'nccl' else 'cpu'
        t = torch.tensor([ok], dtype=torch.int32, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return bool(int(t.item()))

    def _open_window(self, dev):
        """every rank creates its window, the 64-byte IPC handles travel over torch.distributed, every rank maps its peers'"""
        import torch
        import torch.distributed as dist
        lib = self.ctx.lib
        slot = 8 * self.world * self.chunk + 64 * self.world + 4096
        mine = np.zeros(64, dtype=np.uint8)
        self.ctx._check(lib.b3c_dist_window_create(self.ctx.h, slot, mine.ctypes.data_as(C.c_void_p)))
        handles = [None] * self.world
        dist.all_gather_object(handles, mine.tobytes(), group=self.group)
        buf = np.frombuffer(b''.join(handles), dtype=np.uint8).copy()
        assert buf.shape[0] == 64 * self.world
        self.ctx._check(lib.b3c_dist_window_open(self.ctx.h, buf.ctypes.data_as(C.c_void_p)))
        dist.barrier(group=self.group)                   # every window is mapped everywhere before the first push
        self.direct = True
        self.transport += ' + direct peer writes for the all-gathers'
        self._xd0 = int(lib.b3c_ctx_exchanges_direct(self.ctx.h))

    @property
    def n_exchanges_direct(self):
        """exchanges of this session that went through the window (peer writes) instead of a collective"""
        return int(self.ctx.lib.b3c_ctx_exchanges_direct(self.ctx.h)) - getattr(self, '_xd0', 0) if self.direct else 0

    def set_starts(self, starts):
        """Row ranges of the ranks from here on (all ranks must pass the same array), or None for equal chunks again.  The
        pipeline balances them by stored entries once the locality order is known (balanced_starts) and returns to equal
        chunks for the ordering passes of the next call, which run on the still un-ordered map."""
        if starts is None and self.starts is None:
            return
        if starts is not None and not getattr(self, 'ragged_ok', True):
            return                                        # this stack's transport does not move ragged slices: equal ranges stay
        if starts is not None:
            st = np.asarray(starts, dtype=np.int64)
            assert st.shape == (self.world + 1,) and st[0] == 0 and st[-1] == self.n and np.all(np.diff(st) >= 0), st
        self.ctx.sync()
        self._install(starts)

    @property
    def n_exchanges(self):
        """collectives libb3c issued since this session was installed (either transport)"""
        return int(self.ctx.lib.b3c_ctx_exchanges(self.ctx.h)) - self._x0

    def _allgather(self, user, buf_id):
        import torch
        try:
            with torch.cuda.stream(self.ctx.torch_stream):    # the stream libb3c launches on: ordered with its kernels
                for b in ((BUF_SCAL, BUF_VEC) if buf_id == BUF_SCAL_VEC else (buf_id,)):
                    allgather_inplace(self._bufs[b], self.rank, self.world, self.group)
            return 0
        except Exception as ex:                                # never let an exception cross the C frame
            self._error = ex
            return 1

    def _allgatherv(self, user, dev_ptr, esz, starts_p, world):
        """callback transport: ragged in-place all-gather of the row-indexed vector at device pointer `dev_ptr`"""
        import torch
        try:
            starts = [int(starts_p[r]) for r in range(world + 1)]
            total = starts[world]

            class _View(object):
                __cuda_array_interface__ = {'shape': (total,), 'typestr': '<i8' if esz == 8 else '<i4', 'data': (int(dev_ptr), False),
                                            'version': 2}
            with torch.cuda.stream(self.ctx.torch_stream):
                if total:
                    t = torch.as_tensor(_View(), device=torch.device('cuda', self.ctx.device))
                    allgather_ragged(t, starts, self.rank, self.group)
            return 0
        except Exception as ex:
            self._error = ex
            return 1

    def _allreduce(self, user, dev_ptr, count):
        """callback transport: sum `count` uint64 at device pointer `dev_ptr` over the ranks (int64 view: same bits mod 2^64)"""
        import torch
        import torch.distributed as dist
        try:
            class _View(object):
                __cuda_array_interface__ = {'shape': (int(count),), 'typestr': '<i8', 'data': (int(dev_ptr), False), 'version': 2}
            with torch.cuda.stream(self.ctx.torch_stream):
                t = torch.as_tensor(_View(), device=torch.device('cuda', self.ctx.device))
                dist.all_reduce(t, group=self.group)
            return 0
        except Exception as ex:
            self._error = ex
            return 1

    def close(self):
        if getattr(self.ctx, 'dist_session', None) is self:
            self.ctx.dist_session = None
        if self.ctx.h is not None:
            self._final = self.n_exchanges
            self._final_direct = self.n_exchanges_direct
            if self.direct:
                import torch.distributed as dist
                self.ctx.sync()
                dist.barrier(group=self.group)           # nobody unmaps a window a peer may still write into
                self.ctx.lib.b3c_dist_window_close(self.ctx.h)
                self.direct = False
            self.ctx.lib.b3c_ctx_set_dist(self.ctx.h, None)
            if self.comm:
                self.ctx._check(self.ctx.lib.b3c_dist_comm_destroy(self.ctx.h, self.comm))
                self.comm = None

    def check(self):
        if self._error is not None:
            raise self._error
