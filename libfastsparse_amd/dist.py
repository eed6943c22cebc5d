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
import os
import sys

import torch
import torch.distributed as dist


def _option_epoch():
    """the library's option epoch (capi.set_option counts); 0 when the HIP library is not in use (gloo tests with injected kernels)"""
    try:
        from . import capi
        return capi.option_epoch()
    except Exception:
        return 0


def _single(world):
    """one rank and nothing to exchange -- unless FS_DIST_FORCE_COLLECTIVES=1 asks for the multi-rank code path anyway (a
    one-rank RCCL group on the one-GPU box: the collectives, their stream ordering and the unpack all run; the counterpart
    of FS_DIST_FORCE_RCCL in the native path)"""
    return world == 1 and os.environ.get("FS_DIST_FORCE_COLLECTIVES", "0") != "1"


def _host_collective(t, group):
    """gloo moves CUDA tensors only for some collectives; route through the host when the backend is gloo
    (used by the single-GPU rehearsal of the multi-rank path; RCCL takes device tensors directly)"""
    return t.is_cuda and dist.get_backend(group) == "gloo"


class _Done:
    """handle of a collective that already completed (host-routed gloo path, single rank)"""

    def wait(self):
        return True


class _Then:
    """handle of one or several asynchronous collectives followed by device work (e.g. unpacking a padded gather
    buffer): wait() orders the current stream after the collectives, then enqueues `after()` on it"""

    def __init__(self, work, after):
        self.works = list(work) if isinstance(work, (list, tuple)) else [work]
        self.after = after

    def wait(self):
        for w in self.works:
            w.wait()
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


def _peer(r, group):
    """global rank of rank r of `group` (point-to-point calls take global ranks)"""
    return r if group is None else dist.get_global_rank(group, r)


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


def copy_segments_torch(dst, src, dst_off, src_off, counts):
    """dst[dst_off[i] : +counts[i]] = src[src_off[i] : +counts[i]] -- the portable unpack (CPU tests; the HIP provider
    injects fs_copy_segments: one launch for all segments)"""
    for d, s_, c in zip(dst_off, src_off, counts):
        if c:
            dst[d:d + c] = src[s_:s_ + c]


class EvenParts:
    """local product in parts for backends that cannot cut a product (the oracle in the CPU tests): everything is
    computed with part 0, and the rows are reported final in `nparts` even ranges only as their part comes up -- the
    exchange logic above it sees the same protocol as with a kernel that really finishes rows range by range"""

    def __init__(self, local_spmv, nrow_local):
        self.local_spmv, self.n = local_spmv, nrow_local

    def rows(self, nparts):
        return [self.n * p // nparts for p in range(nparts + 1)]

    def run(self, y_local, x_full, part, nparts):
        if part == 0:
            self.local_spmv(y_local, x_full)


class HipParts:
    """local product in parts on the HIP C-ABI (fs_spmv_part / fs_spmm_part: pass 2 of the two-pass pair in ranges of
    panels, the tiled kernels in generations of workgroups); `matrix` is a capi.Matrix, k the number of row-major columns"""

    def __init__(self, matrix, transposed=False, stream_fn=None, k=1):
        from . import capi
        self.m, self.t, self.capi, self.stream_fn, self.k = matrix, transposed, capi, stream_fn, k

    def rows(self, nparts):
        return self.m.part_rows(nparts, transposed=self.t, k=self.k)

    def run(self, y_local, x_full, part, nparts):
        st = self.stream_fn() if self.stream_fn else self.capi.current_stream()
        if self.k == 1:
            self.m.spmv_part(y_local, x_full, part, nparts, st, transposed=self.t)
        else:       # k row-major columns (fs_spmm_part: the k = 2 / 4 sweep is cut like the single-vector pair)
            self.m.spmm_part(y_local, x_full, self.k, part, nparts, st, transposed=self.t)


class ShardedOperator:
    """y = A x with rows [bounds[rank], bounds[rank+1]) of A on this rank.

    apply(y_full, x_full): local product into this rank's slice, then all-gather of the slices.
    Unequal shards are gathered through a padded buffer (RCCL's all-gather wants equal counts).
    apply_overlapped(y_full, x_full, nparts): the exchange INSIDE the product -- the local product runs in parts
    (`parts`: HipParts / EvenParts) and the all-gather of the rows part p has finished is started while part p + 1
    computes ("communication ~ compute and must be overlapped (chunk rows ...)"); what the iterating
    consumer needs, whose next product cannot start before y is complete (cg.h:15-16: y of A x is the x of A').
    `k`: doubles per row (k right-hand sides, row-major).
    """

    def __init__(self, local_spmv, bounds, group=None, parts=None, k=1, copy_segments=None, exchange="allgather"):
        self.local_spmv = local_spmv
        self.bounds = list(bounds)
        self.group = group
        self.parts = parts
        self.k = k
        # how apply_overlapped ships the rows of a part: "allgather" (one all_gather_into_tensor per part on a padded buffer,
        # unpacked by one fs_copy_segments launch) or "direct" ("each GPU pushes its shard on all 7 links": one
        # batch of point-to-point sends / receives per part, every rank's rows straight into their place in y, no padding and
        # no unpack).  Which is faster over xGMI is for a machine with more than one GPU to tell (bench.py --exchange).
        assert exchange in ("allgather", "direct")
        self.exchange = exchange
        self.copy_segments = copy_segments or copy_segments_torch
        self._plan = {}
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
        # conservative exchange: ONE whole-shard all-gather behind the finished local product instead of one per part under the
        # later parts (FS_DIST_CONSERVATIVE=1, or after verify_overlap found the overlapped exchange unsound on this machine)
        self.conservative = os.environ.get("FS_DIST_CONSERVATIVE", "") == "1"

    def verify_overlap(self, y_full, x_full, nparts=4):
        """First contact with a fabric the overlapped exchange has never run on (RCCL with more than one rank had not, when this
        was written): the same product with the exchange INSIDE it and with the plain whole-shard all-gather behind it must give
        the same vector on every rank (to rounding: the parts do not change a row's terms, only -- without fixed-order sums --
        their order).  If any rank sees a difference, every rank switches this operator to the conservative exchange.  Collective.
        Returns {"mode": "overlapped" | "conservative", "max_rel_diff": ..., and after a disagreement "first_bad_segment_per_rank":
        per rank the first (owner rank, part, row range) whose rows arrived wrong} (what bench.py puts into its record)."""
        if _single(self.world) or self.parts is None or nparts <= 1 or self.conservative:
            return {"mode": "conservative" if self.conservative else "overlapped", "checked": False}
        yo = torch.empty_like(y_full)
        self.apply_overlapped_async(yo, x_full, nparts).wait()
        self.gather(y_full, self.local(y_full, x_full))
        inject = os.environ.get("FS_DIST_INJECT_MISMATCH", "")      # "rank:part" (tests): that segment of the overlapped result is spoiled
        plan = self._part_plan(nparts, y_full) if self.exchange == "allgather" or inject else None
        if inject and plan is not None:
            r_, p_ = (int(v) for v in inject.split(":"))
            for _ in range(nparts):         # (a kernel that finishes no row in that part: the next part, cyclically, that does)
                if plan["cuts"][r_][p_ + 1] > plan["cuts"][r_][p_]:
                    break
                p_ = (p_ + 1) % nparts
            a = (self.bounds[r_] + plan["cuts"][r_][p_]) * self.k
            if plan["cuts"][r_][p_ + 1] > plan["cuts"][r_][p_]:
                yo[a] += 1.0
        scale = float(y_full.abs().max().item())
        scale = scale if scale > 0 else 1.0
        diff = float((yo - y_full).abs().max().item()) / scale
        if os.environ.get("FS_DIST_DEBUG"):
            print("verify_overlap rank", self.rank, "inject", repr(inject), "plan", None if plan is None else plan["cuts"], "diff", diff, "scale", scale, file=sys.stderr, flush=True)
        first_bad = None
        if diff > 1e-9 and plan is not None:
            # whose rows, of which part, arrived wrong on THIS rank: the first segment of the padded layout that differs
            for p in range(nparts):
                for r in range(self.world):
                    a, b = self.bounds[r] + plan["cuts"][r][p], self.bounds[r] + plan["cuts"][r][p + 1]
                    if b > a:
                        d = float((yo[a * self.k:b * self.k] - y_full[a * self.k:b * self.k]).abs().max().item()) / scale
                        if d > 1e-9:
                            first_bad = {"seen_by_rank": self.rank, "rows_owned_by_rank": r, "part": p, "of_parts": nparts,
                                         "row_range": [a, b], "max_rel_diff": d}
                            break
                if first_bad:
                    break
        bad = torch.tensor([0.0 if diff <= 1e-9 else 1.0, diff], dtype=torch.float64, device=y_full.device)
        if _host_collective(bad, self.group):
            h = bad.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.MAX, group=self.group)
            bad = h
        else:
            dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=self.group)
        self.conservative = bool(float(bad[0].item()) > 0.0)
        out = {"mode": "conservative" if self.conservative else "overlapped", "checked": True, "max_rel_diff": float(bad[1].item()),
               "parts": nparts, "exchange": self.exchange}
        if self.conservative:
            seen = [None] * self.world
            dist.all_gather_object(seen, first_bad, group=self.group)       # every rank's first bad segment (None: that rank saw none)
            out["first_bad_segment_per_rank"] = seen
        return out

    @property
    def nrow(self):
        return self.bounds[-1]

    def _buffers(self, like):
        if self._y_local is None or self._y_local.device != like.device:
            # twice the largest shard: the send window of a part may run past the end of the shard (padded counts)
            self._y_local = torch.empty(2 * self.max_rows * self.k, dtype=like.dtype, device=like.device)
            if not self.equal:
                self._pad = torch.empty(self.max_rows * self.world * self.k, dtype=like.dtype, device=like.device)
        return self._y_local

    def apply_local(self, y_local, x_full):
        self.local_spmv(y_local, x_full)

    def local(self, y_full, x_full):
        """local product only; returns the tensor `gather` has to be called with"""
        n_local = self.hi - self.lo
        if _single(self.world):
            self.local_spmv(y_full, x_full)
            return y_full
        y_local = self._buffers(y_full)
        self.local_spmv(y_local[:n_local * self.k], x_full)
        return y_local

    def gather(self, y_full, y_local):
        """the exchange step: all-gather of the y shards (no-op on one rank)"""
        if _single(self.world):
            return y_full
        n_local = self.hi - self.lo
        if self.equal:
            all_gather_into(y_full, y_local[:n_local * self.k], self.group)
        else:
            all_gather_into(self._pad, y_local[:self.max_rows * self.k], self.group)
            self._unpack(y_full)
        return y_full

    def _unpack(self, y_full):
        k, m = self.k, self.max_rows
        self.copy_segments(y_full, self._pad, [self.bounds[r] * k for r in range(self.world)],
                           [r * m * k for r in range(self.world)], [self.sizes[r] * k for r in range(self.world)])

    def gather_async(self, y_full, y_local):
        """start the exchange step and return a handle whose wait() leaves y_full complete on the current stream.
        Equal shards: one all-gather straight into y_full.  Unequal shards (the nnz-balanced cut of a power-law
        matrix, BASELINE config 5): one all-gather of max-sized shards into a padded buffer, unpacked by `world`
        slice copies when the handle is waited for -- the transfer itself overlaps whatever is launched in between."""
        if _single(self.world):
            return _Done()
        if self.equal:
            return all_gather_into_async(y_full, y_local[:(self.hi - self.lo) * self.k], self.group)
        work = all_gather_into_async(self._pad, y_local[:self.max_rows * self.k], self.group)
        return _Then(work, lambda: self._unpack(y_full))

    def apply(self, y_full, x_full):
        return self.gather(y_full, self.local(y_full, x_full))

    # ---- the exchange inside the product ---------------------------------------------------------------------
    def _part_plan(self, nparts, like):
        """once per nparts: every rank's row cuts (exchanged), the padded count of every part, the unpack table"""
        # the cuts belong to the kernel the options select: an option set since the plan was made (strict_order, reproducible,
        # spmv_kernel move the product to another kernel, which finishes its rows elsewhere) makes every rank plan again at
        # its next product -- ranks set options in step (SPMD), so they enter the exchange of the cuts together
        epoch = _option_epoch()
        if getattr(self, "_plan_epoch", None) != epoch:
            self._plan.clear()
            self._plan_epoch = epoch
        key = (nparts, like.device)
        if key in self._plan:
            return self._plan[key]
        mine = [int(v) for v in self.parts.rows(nparts)]
        cuts = [None] * self.world
        dist.all_gather_object(cuts, mine, group=self.group)
        k = self.k
        maxc = [max(cuts[r][p + 1] - cuts[r][p] for r in range(self.world)) for p in range(nparts)]
        off = [0]
        for p in range(nparts):
            off.append(off[-1] + self.world * maxc[p] * k)
        dst, src, cnt = [], [], []
        for p in range(nparts):
            for r in range(self.world):
                c = cuts[r][p + 1] - cuts[r][p]
                if c:
                    dst.append((self.bounds[r] + cuts[r][p]) * k)
                    src.append(off[p] + r * maxc[p] * k)
                    cnt.append(c * k)
        plan = {"mine": mine, "cuts": cuts, "maxc": maxc, "off": off, "dst": dst, "src": src, "cnt": cnt,
                "pad": torch.empty(max(off[-1], 1), dtype=like.dtype, device=like.device)}
        self._plan[key] = plan
        return plan

    def apply_overlapped_async(self, y_full, x_full, nparts=4):
        """local product in `nparts` parts; the all-gather of part p's rows is started as soon as part p is enqueued and
        runs under parts p + 1 ...; returns a handle whose wait() leaves y_full complete on the current stream"""
        if _single(self.world) or self.parts is None or nparts <= 1 or self.conservative:
            return self.gather_async(y_full, self.local(y_full, x_full))
        y_local = self._buffers(y_full)
        plan = self._part_plan(nparts, y_full)
        k, n_local = self.k, self.hi - self.lo
        works = []
        if self.exchange == "direct" and not _host_collective(y_full, self.group):
            # the local product writes straight into this rank's rows of y_full; after part p every peer is sent those rows and
            # this rank posts the receives of the peers' part-p rows (cuts exchanged by _part_plan) into their place in y_full
            mine_full = y_full[self.lo * k:self.hi * k]
            cuts = plan["cuts"]
            for p in range(nparts):
                self.parts.run(mine_full, x_full, p, nparts)
                ops = []
                a, b = cuts[self.rank][p], cuts[self.rank][p + 1]
                for r in range(self.world):
                    if r == self.rank:
                        continue
                    ra, rb = cuts[r][p], cuts[r][p + 1]
                    if rb > ra:
                        ops.append(dist.P2POp(dist.irecv, y_full[(self.bounds[r] + ra) * k:(self.bounds[r] + rb) * k], _peer(r, self.group),
                                              group=self.group))
                    if b > a:
                        ops.append(dist.P2POp(dist.isend, mine_full[a * k:b * k], _peer(r, self.group), group=self.group))
                if ops:
                    works.extend(dist.batch_isend_irecv(ops))
            return _Then(works, lambda: None)
        for p in range(nparts):
            self.parts.run(y_local[:n_local * k], x_full, p, nparts)
            c = plan["maxc"][p]
            if c:
                a = plan["mine"][p] * k
                works.append(all_gather_into_async(plan["pad"][plan["off"][p]:plan["off"][p + 1]], y_local[a:a + c * k], self.group))
        return _Then(works, lambda: self.copy_segments(y_full, plan["pad"], plan["dst"], plan["src"], plan["cnt"]))

    def apply_overlapped(self, y_full, x_full, nparts=4):
        self.apply_overlapped_async(y_full, x_full, nparts).wait()
        return y_full


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
        if not _single(self.world):
            all_reduce_sum(z_full, self.group)
        return z_full

    def reduce_async(self, z_full):
        return _Done() if _single(self.world) else all_reduce_sum_async(z_full, self.group)

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
    if _single(world):
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


class ShardedCG:
    """(A'A + lambda I) x = b on a row-sharded A: bsbm_cg (cg.h:25-82) and, for two right-hand sides, bsbm_cg2
    (cg.h:85-187), the consumers that ITERATE on the path (y of A x is the x of A', cg.h:15-16, 134-135).

    scheme "gather" (both directions are "rows + all-gather"): op_a = ShardedOperator on the rows of A, op_t =
        ShardedOperator on the rows of A' this rank owns (= its slice of the F unknowns).  Every rank keeps its SLICE of
        x, r, p, q; per iteration: y = A p (local rows, all-gather inside the product: apply_overlapped), q_r = A'_r y +
        lambda p_r (no exchange), the dots as all-reduces of 1 (k = 1) or 3 (k = 2) doubles, the updates on the slices,
        and the all-gather of the new p.  Moves N + F doubles per right-hand side and iteration.
    scheme "reduce": op_t = TransposedShardedOperator (every rank multiplies the transpose of its own rows; an
        all-reduce sums the full-length partials).  x, r, p, q are replicated, so the dots need no exchange and equal
        the one-GPU solver's.  Moves 2 F doubles per right-hand side and iteration, nothing of length N: the scheme for
        tall matrices (config 3's shape).
    Vector algebra is torch (device-agnostic: RCCL ranks and the gloo CPU tests run the same code); products are the
    injected local kernels."""

    def __init__(self, op_a, op_t, scheme="gather", nparts=4, group=None):
        assert scheme in ("gather", "reduce")
        self.op_a, self.op_t, self.scheme, self.nparts, self.group = op_a, op_t, scheme, nparts, group
        self.k = op_a.k
        self.world = op_a.world

    def _sum(self, t):
        """sum over the ranks of a few partial dots"""
        if not _single(self.world):
            all_reduce_sum(t, self.group)
        return t

    def _dots(self, X, Y, n):
        """k = 1: [x.y]; k = 2: [a.a', b.b', a.b'] of the row-major two-column X with Y (linalg.h:61-73)"""
        if self.k == 1:
            return torch.stack([torch.dot(X, Y)])
        X2, Y2 = X.view(n, 2), Y.view(n, 2)
        return torch.stack([torch.dot(X2[:, 0], Y2[:, 0]), torch.dot(X2[:, 1], Y2[:, 1]), torch.dot(X2[:, 0], Y2[:, 1])])

    @staticmethod
    def _solve2sym(A, rhs):
        """linalg.h:77-88 on host floats: A = [a00, a11, a01] symmetric, rhs row-major 2 x 2"""
        dinv = 1.0 / (A[0] * A[1] - A[2] * A[2])
        i0, i1, i2 = dinv * A[1], dinv * A[0], -dinv * A[2]
        return [i0 * rhs[0] + i2 * rhs[1], i2 * rhs[0] + i1 * rhs[1], i0 * rhs[2] + i2 * rhs[3], i2 * rhs[2] + i1 * rhs[3]]

    def solve(self, b_full, lam, tol, max_iter=None):
        """b_full: the whole right-hand side (F, or F x 2 row-major, flat) on every rank.  Returns (x_full, iterations);
        x_full is complete on every rank."""
        k, op_a, op_t = self.k, self.op_a, self.op_t
        F = b_full.numel() // k
        N = op_a.nrow
        gather = self.scheme == "gather"
        lo, hi = (op_t.lo, op_t.hi) if gather else (0, F)
        n = hi - lo
        dev, dt = b_full.device, b_full.dtype
        bl = b_full[lo * k:hi * k]
        y_full = torch.empty(N * k, dtype=dt, device=dev)
        p_full = torch.empty(F * k, dtype=dt, device=dev)
        x = torch.zeros(n * k, dtype=dt, device=dev)
        if k == 1:
            r = bl.clone()
            norms = None
        else:
            nb = [float(v) for v in self._sum(self._dots(bl, bl, n)[:2].clone()).sqrt().cpu()] if gather else \
                 [float(v) for v in self._dots(bl, bl, n)[:2].sqrt().cpu()]
            norms = torch.tensor(nb, dtype=dt, device=dev)
            r = (bl.view(n, 2) / norms).reshape(-1)
        p = r.clone()

        def red(t):
            return [float(v) for v in (self._sum(t) if gather else t).cpu()]

        if k == 1:
            rr_t = self._dots(r, r, n)
            rr_t = self._sum(rr_t) if gather else rr_t
            stop_t = tol * rr_t.sqrt()
            rr = None
        else:
            rr = red(self._dots(r, r, n))
        stop = tol * tol
        q = torch.empty(n * k, dtype=dt, device=dev)
        q_gather_buf = None
        it = 0
        max_iter = F if max_iter is None else max_iter
        while it < max_iter:
            # p on every rank (gather: the slices are all-gathered; reduce: it is replicated already)
            if gather and not _single(self.world):
                pl = op_t._buffers(p_full)
                pl[:n * k] = p
                op_t.gather(p_full, pl)
            else:
                p_full[:] = p
            if gather:
                op_a.apply_overlapped(y_full, p_full, self.nparts)      # y = A p, exchange inside the product
                op_t.local_spmv(q, y_full)                               # q_r = A'_r y
            else:
                t_local = op_a.local(y_full, p_full)                     # rows of this rank only: nothing of length N moves
                nl = op_a.hi - op_a.lo
                op_t.local_tspmv(q, y_full if _single(op_a.world) else t_local[:nl * k])
                op_t.reduce(q)
            q.add_(p, alpha=lam)
            if k == 1:
                # the scalars stay where the vectors are (device tensors with RCCL): alpha and beta are never fetched, the
                # one host round trip of an iteration is the convergence test
                pq = self._dots(p, q, n)
                pq = self._sum(pq) if gather else pq
                alpha = rr_t / pq
                x.addcmul_(p, alpha)
                r.addcmul_(q, -alpha)
                rr2_t = self._dots(r, r, n)
                rr2_t = self._sum(rr2_t) if gather else rr2_t
                if bool((rr2_t.sqrt() <= stop_t).item()):
                    break
                p.mul_(rr2_t / rr_t).add_(r)
                rr_t = rr2_t
                it += 1
                continue
            else:
                ptkp = red(self._dots(p, q, n))
                al = self._solve2sym(ptkp, [rr[0], rr[2], rr[2], rr[1]])
                P2, Q2, X2, R2 = p.view(n, 2), q.view(n, 2), x.view(n, 2), r.view(n, 2)
                M = torch.tensor([[al[0], al[2]], [al[1], al[3]]], dtype=dt, device=dev)   # X += P M: row i gets [a0 p0 + a1 p1, a2 p0 + a3 p1]
                X2.add_(P2 @ M)
                R2.sub_(Q2 @ M)
                rr2 = red(self._dots(r, r, n))
                if rr2[0] <= stop and rr2[1] <= stop:
                    break
                ps = self._solve2sym(rr, [rr2[0], rr2[2], rr2[2], rr2[1]])
                Mp = torch.tensor([[ps[0], ps[2]], [ps[1], ps[3]]], dtype=dt, device=dev)
                P2.copy_(R2 + P2 @ Mp)
            rr = rr2
            it += 1
        if k == 2:
            x = (x.view(n, 2) * norms).reshape(-1)
        if gather and not _single(self.world):
            x_full = torch.empty(F * k, dtype=dt, device=dev)
            xl = op_t._buffers(x_full)
            xl[:n * k] = x
            op_t.gather(x_full, xl)
            return x_full, it
        return x, it


def hip_local_spmv(matrix, stream_fn=None):
    """local_spmv backed by the HIP C-ABI (fs_spmv on the current torch stream)"""
    from . import capi

    def f(y_local, x_full):
        matrix.spmv(y_local, x_full, stream_fn() if stream_fn else capi.current_stream())
    return f
