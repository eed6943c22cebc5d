"""Row-sharded products across the GPUs of one node: one process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

The path shards by rows: y[r] depends on row r only, so each rank owns a contiguous
row range of A as an ordinary CSR with local row_ptr and GLOBAL column ids, x is replicated, and the
one real exchange step is the all-gather of the y shards.  The transposed product uses the same
scheme on a row shard of A' (= the columns of A this rank owns), built once by an all-to-all of the
entries (`exchange_transpose_entries`), so both directions are "local SpMV + all-gather".

Nothing here computes: `local_spmv(y_local, x_full)` is injected -- the HIP C-ABI in the product
(`hip_local_spmv`), the oracle in the gloo tests.
"""
import torch
import torch.distributed as dist


def _host_collective(t, group):
    """gloo moves CUDA tensors only for some collectives; route through the host when the backend is gloo
    (used by the single-GPU rehearsal of the multi-rank path; RCCL takes device tensors directly)"""
    return t.is_cuda and dist.get_backend(group) == "gloo"


class _Done:
    """handle of a collective that already completed (host-routed gloo path, single rank)"""

    def wait(self):
        return True


class _Then:
    """handle of an asynchronous collective followed by device work (e.g. unpacking a padded gather buffer):
    wait() orders the current stream after the collective, then enqueues `after()` on it"""

    def __init__(self, work, after):
        self.work, self.after = work, after

    def wait(self):
        self.work.wait()
        self.after()
        return True


def all_gather_into_async(out, inp, group=None):
    """start the all-gather; the returned handle's wait() orders the CURRENT stream after it.  Kernels launched
    on the current stream in between overlap with the transfer (RCCL runs on its own stream)."""
    if _host_collective(inp, group):
        all_gather_into(out, inp, group)
        return _Done()
    return dist.all_gather_into_tensor(out, inp, group=group, async_op=True)


def all_reduce_sum_async(t, group=None):
    if _host_collective(t, group):
        all_reduce_sum(t, group)
        return _Done()
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True)


def all_gather_into(out, inp, group=None):
    if _host_collective(inp, group):
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(o, inp.cpu(), group=group)
        out.copy_(o)
    else:
        dist.all_gather_into_tensor(out, inp, group=group)


def all_to_all(out, inp, out_split=None, in_split=None, group=None):
    if _host_collective(inp, group):
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu(), out_split, in_split, group=group)
        out.copy_(o)
    else:
        dist.all_to_all_single(out, inp, out_split, in_split, group=group)


def even_row_partition(nrow, world):
    """contiguous row ranges of (almost) equal size: bounds[r] .. bounds[r+1]"""
    base, rem = divmod(nrow, world)
    bounds = [0]
    for r in range(world):
        bounds.append(bounds[-1] + base + (1 if r < rem else 0))
    return bounds


def nnz_balanced_partition(row_ptr, world):
    """contiguous row ranges holding (almost) equal numbers of non-zeros -- what a power-law matrix
    (BASELINE config 5) needs; row_ptr is the global CSR row pointer (any integer tensor / array)."""
    rp = torch.as_tensor(row_ptr).to(torch.int64).cpu()
    nrow = rp.numel() - 1
    nnz = int(rp[-1])
    targets = torch.tensor([nnz * r // world for r in range(1, world)], dtype=torch.int64)
    cuts = torch.searchsorted(rp, targets, right=False).clamp_(0, nrow)
    bounds = [0] + [int(c) for c in cuts] + [nrow]
    for i in range(1, len(bounds)):
        bounds[i] = max(bounds[i], bounds[i - 1])
    return bounds


class ShardedOperator:
    """y = A x with rows [bounds[rank], bounds[rank+1]) of A on this rank.

    apply(y_full, x_full): local product into this rank's slice, then all-gather of the slices.
    Unequal shards are gathered through a padded buffer (RCCL's all-gather wants equal counts).
    """

    def __init__(self, local_spmv, bounds, group=None):
        self.local_spmv = local_spmv
        self.bounds = list(bounds)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        assert len(self.bounds) == self.world + 1
        self.lo, self.hi = self.bounds[self.rank], self.bounds[self.rank + 1]
        sizes = [self.bounds[r + 1] - self.bounds[r] for r in range(self.world)]
        self.sizes = sizes
        self.equal = len(set(sizes)) == 1
        self.max_rows = max(sizes)
        self._pad = None
        self._y_local = None

    @property
    def nrow(self):
        return self.bounds[-1]

    def _buffers(self, like):
        if self._y_local is None or self._y_local.device != like.device:
            self._y_local = torch.empty(self.max_rows, dtype=like.dtype, device=like.device)
            if not self.equal:
                self._pad = torch.empty(self.max_rows * self.world, dtype=like.dtype, device=like.device)
        return self._y_local

    def apply_local(self, y_local, x_full):
        self.local_spmv(y_local, x_full)

    def local(self, y_full, x_full):
        """local product only; returns the tensor `gather` has to be called with"""
        n_local = self.hi - self.lo
        if self.world == 1:
            self.local_spmv(y_full, x_full)
            return y_full
        y_local = self._buffers(y_full)
        self.local_spmv(y_local[:n_local], x_full)
        return y_local

    def gather(self, y_full, y_local):
        """the exchange step: all-gather of the y shards (no-op on one rank)"""
        if self.world == 1:
            return y_full
        n_local = self.hi - self.lo
        if self.equal:
            all_gather_into(y_full, y_local[:n_local], self.group)
        else:
            all_gather_into(self._pad, y_local, self.group)
            self._unpack(y_full)
        return y_full

    def _unpack(self, y_full):
        for r in range(self.world):
            y_full[self.bounds[r]:self.bounds[r + 1]] = \
                self._pad[r * self.max_rows: r * self.max_rows + self.sizes[r]]

    def gather_async(self, y_full, y_local):
        """start the exchange step and return a handle whose wait() leaves y_full complete on the current stream.
        Equal shards: one all-gather straight into y_full.  Unequal shards (the nnz-balanced cut of a power-law
        matrix, BASELINE config 5): one all-gather of max-sized shards into a padded buffer, unpacked by `world`
        slice copies when the handle is waited for -- the transfer itself overlaps whatever is launched in between."""
        if self.world == 1:
            return _Done()
        if self.equal:
            return all_gather_into_async(y_full, y_local[:self.hi - self.lo], self.group)
        work = all_gather_into_async(self._pad, y_local, self.group)
        return _Then(work, lambda: self._unpack(y_full))

    def apply(self, y_full, x_full):
        return self.gather(y_full, self.local(y_full, x_full))


def all_reduce_sum(t, group=None):
    if _host_collective(t, group):
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


class TransposedShardedOperator:
    """z = A' u for a row-sharded A: every rank multiplies its own rows' transpose with its own slice of u
    (z_r = A_r' u_r, a full-length vector) and the partial results are summed by an all-reduce.
    `local_tspmv(z_full, u_local)` is the local transposed product (cached CSR of A_r' on the GPU)."""

    def __init__(self, local_tspmv, bounds, group=None):
        self.local_tspmv = local_tspmv
        self.bounds = list(bounds)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.lo, self.hi = self.bounds[self.rank], self.bounds[self.rank + 1]

    def apply_local(self, z_full, u_full):
        self.local_tspmv(z_full, u_full[self.lo:self.hi])

    def reduce(self, z_full):
        """the exchange step: sum of the partial results over the ranks (no-op on one rank)"""
        if self.world > 1:
            all_reduce_sum(z_full, self.group)
        return z_full

    def reduce_async(self, z_full):
        return all_reduce_sum_async(z_full, self.group) if self.world > 1 else _Done()

    def apply(self, z_full, u_full):
        self.apply_local(z_full, u_full)
        return self.reduce(z_full)


class TransposedGatherOperator(ShardedOperator):
    """z = A' u as "rows of A' + all-gather": this rank owns rows [col_bounds[rank], col_bounds[rank+1]) of A'
    (= those columns of A), built once from the row shards of A by `exchange_transpose_entries`; u (length
    nrow of A) is replicated, the local product gives this rank's slice of z and the one exchange step is the
    all-gather of the z slices (config 5: 100 MB per rank instead of an 800 MB all-reduce of full-length partials).
    `local_spmv(z_local, u_full)` is the product with the local shard of A'."""

    def __init__(self, local_spmv, col_bounds, group=None):
        super().__init__(local_spmv, col_bounds, group)


def build_transposed_shard(row_ptr_local, cols_global, vals, row_lo, col_bounds, group=None):
    """Entries of this rank's row shard of A (local CSR row_ptr, GLOBAL column ids, first global row row_lo) ->
    this rank's row shard of A' as COO (rows local to the shard, GLOBAL column ids = rows of A, values), each row of
    A' in ascending A-row order (the order a stable column sort of the whole matrix gives)."""
    n_local = row_ptr_local.numel() - 1
    lens = (row_ptr_local[1:] - row_ptr_local[:-1]).to(torch.int64)
    rows_global = torch.repeat_interleave(
        torch.arange(row_lo, row_lo + n_local, device=cols_global.device, dtype=cols_global.dtype), lens)
    return exchange_transpose_entries(rows_global, cols_global, vals, col_bounds, group)


def exchange_transpose_entries(rows_global, cols_global, vals, col_bounds, group=None):
    """One-time build of this rank's row shard of A'.

    Each rank holds some entries (row, col, val) of A with GLOBAL ids.  Entry (r, c, v) of A is entry
    (c, r, v) of A'; it goes to the rank that owns row c of A' (col_bounds = row partition of A').
    Returns (rows_of_At_local, cols_of_At_global, vals) for this rank, ordered by (source rank, source
    order): since the source ranks hold ascending row ranges of A, every row of A' receives its entries
    in ascending A-row order -- the order a stable column sort of the whole A would give.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    bounds_t = torch.tensor(col_bounds[1:-1], dtype=cols_global.dtype, device=cols_global.device)

    def local_part():
        d = torch.bucketize(cols_global, bounds_t, right=True)
        # `world` distinct keys: one byte each sorts in a fraction of the time of int64 keys (a stable partition)
        o = torch.sort(d.to(torch.uint8) if world <= 255 else d, stable=True).indices
        return d, o, torch.bincount(d, minlength=world).to(torch.int64)

    if world > 1 and _host_collective(cols_global, group):
        # ranks sharing ONE GPU (gloo rehearsal): one after the other -- the device sorts of several processes time-sliced
        # on one card were seen to starve each other for ever (see bench.py in_turns)
        dest = order = send_counts = None
        for r in range(world):
            if r == rank:
                dest, order, send_counts = local_part()
                torch.cuda.synchronize()
            dist.barrier(group)
    else:
        dest, order, send_counts = local_part()
    lo = col_bounds[rank]
    if world == 1:
        return (cols_global[order] - lo), rows_global[order], (None if vals is None else vals[order])
    recv_counts = torch.empty_like(send_counts)
    all_to_all(recv_counts, send_counts, group=group)
    s_split = [int(v) for v in send_counts.cpu()]
    r_split = [int(v) for v in recv_counts.cpu()]
    n_recv = sum(r_split)

    def xchg(t):
        out = torch.empty(n_recv, dtype=t.dtype, device=t.device)
        all_to_all(out, t[order].contiguous(), r_split, s_split, group=group)
        return out

    at_rows = xchg(cols_global) - lo
    at_cols = xchg(rows_global)
    at_vals = None if vals is None else xchg(vals)
    return at_rows, at_cols, at_vals


def hip_local_spmv(matrix, stream_fn=None):
    """local_spmv backed by the HIP C-ABI (fs_spmv on the current torch stream)"""
    from . import capi

    def f(y_local, x_full):
        matrix.spmv(y_local, x_full, stream_fn() if stream_fn else capi.current_stream())
    return f
