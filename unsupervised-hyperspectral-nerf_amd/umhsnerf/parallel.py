"""Data parallelism for the hot path: rays shard across ranks (one process per GPU), the model is replicated, and the
only exchange is ONE all-reduce(sum) of the flat "fields" gradient per step (RCCL over xGMI on MI355X; gloo in the
CPU tests).  This is the hook the reference disables by forcing world_size = 1 (umhs_pipeline.py:86,108-113)."""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_rays(num_rays_global: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous [begin, end) slice of a global ray batch owned by `rank` (remainder spread over the first ranks)."""
    base, rem = divmod(num_rays_global, world_size)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def rank_seed(seed: int, rank: int) -> int:
    """nerfstudio semantics: every rank draws its own rays from seed + rank."""
    return seed + rank


EXCHANGE_DISABLED = False  # diagnostics only (bench.py's compute-only pass): skip every gradient collective
STATS = {"bytes": 0, "messages": 0}  # payload handed to all_reduce since the last reset (bench.py reports MB per step)


def _all_reduce(t: torch.Tensor, async_op: bool = False):
    STATS["bytes"] += t.numel() * t.element_size()
    STATS["messages"] += 1
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=async_op)


def allreduce_flat_grad(grad: torch.Tensor) -> float:
    """Sum the flat gradient over ranks in place; returns the factor (1/world) the optimizer applies to average it."""
    _, w = world()
    if w > 1 and not EXCHANGE_DISABLED:
        _all_reduce(grad)
    return 1.0 / w


def broadcast_params(flat: torch.Tensor, src: int = 0) -> None:
    if world()[1] > 1:
        dist.broadcast(flat, src=src)


def live_hash_rows(scalings, log2_T: int, max_fill: float = 0.5):
    """Rows of the hash table that can ever receive a gradient, for the leading (coarse) levels whose (res+1)^3 grid corners
    occupy at most ``max_fill`` of the 2^log2_T slots.  A corner (x, y, z), 0 <= x,y,z <= res = scalings[l], lives in slot
    (x ^ y*2654435761 ^ z*805459861) mod T (nerfstudio HashEncoding.hash_fn, uint32 wrap-around); every other slot of such a level
    keeps an exactly zero gradient on every rank, so it need not travel in the all-reduce.  -> (n_sparse_levels, int64 rows)."""
    import numpy as np

    T = 1 << log2_T
    rows, n_sparse = [], 0
    for l, s in enumerate([int(v) for v in scalings.tolist()]):
        if (s + 1) ** 3 > max_fill * T:
            break
        c = np.arange(s + 1, dtype=np.uint64)
        x, y, z = np.meshgrid(c, (c * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF), (c * np.uint64(805459861)) & np.uint64(0xFFFFFFFF),
                              indexing="ij")
        slot = np.unique((x ^ y ^ z) & np.uint64(T - 1)).astype(np.int64)
        rows.append(slot + l * T)
        n_sparse = l + 1
    if not rows:
        return 0, torch.empty(0, dtype=torch.int64)
    return n_sparse, torch.from_numpy(np.concatenate(rows))


class FlatGradSink:
    """Persistent flat gradient buffer for the "fields" parameter, with early reduction of finished segments.

    ``ops.FieldFn.backward`` writes the gradient straight into this buffer and hands it over as ``param.grad`` itself (no
    autograd accumulation copy).  With more than one rank every segment is all-reduced (async, RCCL stream) as soon as the
    kernel that finishes it has been launched: the small MLP/endmember tail right after the field backward, then the hash
    table per level group while the partition/reduce kernels of the next group still run -- so most of the 67 MB
    exchange hides behind the tail of the backward instead of following it.  ``UMHSAdam.step`` waits for the pending
    reductions and applies 1/world.

    Gradient accumulation (the trainer's ``gradient_accumulation_steps``, scripts/rgb+spectral.sh:5): a backward that finds
    ``param.grad`` to be this very buffer ADDS to it (``accumulating``): the hash-grid scatter runs in its ``+=`` mode and the
    MLP / endmember tail goes through a small temporary.  With more than one rank the exchange must start in the last
    micro-step only: the trainer sets ``defer_reduce`` on the others (UMHSPipeline does, from its ``gradient_accumulation_steps``).
    """

    def __init__(self, param: torch.nn.Parameter, level_groups: Optional[int] = None):
        self.param, self.buffer, self.works = param, None, []
        # Two 8-level (33.5 MB) messages by default: RCCL's ring all-reduce over xGMI loses ~30 % of its bus bandwidth at 16 MB, so
        # finer groups buy less overlap than they cost in transfer time; UMHS_REDUCE_GROUPS overrides (1, 2, 4, 8, 16).
        self.level_groups = int(level_groups if level_groups is not None else os.environ.get("UMHS_REDUCE_GROUPS", "2"))
        self.async_reduce = True  # exchanges are issued as soon as a level group's gradient is final and waited for in front of the optimizer
        self.reduced_ptr = None
        self.sparse_levels, self.sparse_rows = 0, None  # set_sparse_levels(): coarse levels travel as their live rows only
        self.fused_adam, self.adam_done = None, None  # UMHSAdam.arm_fused() / what the backward then did (step, begin, end)
        self.accumulating = False  # this backward adds to the buffer (set by begin())
        self.defer_reduce = False  # not the last micro-step of an accumulation window: no exchange yet

    def set_sparse_levels(self, scalings, log2_T: int) -> None:
        """Coarse hash levels use a small, rank-independent subset of their 2^log2_T slots (4,913 of 524,288 at level 0): send the
        live rows of those levels as one compact message instead of their whole slabs (-15 MB of 67 MB at the reference sizes)."""
        self.sparse_levels, rows = live_hash_rows(scalings, log2_T)
        self.sparse_rows = rows.to(self.param.device) if self.sparse_levels else None
        self.table_rows = 1 << log2_T

    def take_fused_adam(self, param) -> Optional[dict]:
        """The optimizer step a trainer armed for this backward (one shot), if it is for this parameter and nothing is reduced."""
        fa, self.fused_adam = self.fused_adam, None
        if fa is None or world()[1] != 1 or param is not self.param:
            return None
        return fa

    def _grad_is_buffer(self) -> bool:
        g = self.param.grad
        return g is not None and self.buffer is not None and g.data_ptr() == self.buffer.data_ptr() and g.shape == self.buffer.shape

    def owns_next_backward(self) -> bool:
        """True when the coming backward may write this buffer directly: no gradient yet (overwrite), or the gradient IS this
        buffer (a further micro-step of an accumulation window: add).  A foreign ``param.grad`` leaves the job to autograd."""
        return self.param.grad is None or self._grad_is_buffer()

    def begin(self) -> torch.Tensor:
        p = self.param
        self.accumulating = self._grad_is_buffer()
        if self.accumulating:
            if self.works or self.reduced_ptr is not None:
                raise RuntimeError("gradient accumulation on more than one rank: the previous micro-step already started its all-reduce. "
                                   "Tell the pipeline (UMHSPipelineConfig.gradient_accumulation_steps = the trainer's value) so that "
                                   "only the last micro-step exchanges gradients")
            return self.buffer
        self.fresh = self.buffer is None or self.buffer.shape != p.shape or self.buffer.device != p.device
        if self.fresh:  # zeroed once: the backward overwrites every parameter's entry each step, never the alignment padding
            self.buffer = torch.zeros_like(p.data)
        self.works.clear()
        self.reduced_ptr = None
        return self.buffer

    def reducing(self) -> bool:
        return world()[1] > 1 and self.async_reduce and not self.defer_reduce and not EXCHANGE_DISABLED

    def groups(self, n_levels: int):
        if not self.reducing():
            return [(0, n_levels)]
        g = max(1, n_levels // self.level_groups)
        out = [(l, min(g, n_levels - l)) for l in range(0, n_levels, g)]
        while len(out) > 1 and out[0][1] < self.sparse_levels:  # the compact message needs all sparse levels in the first group
            out[0:2] = [(0, out[0][1] + out[1][1])]
        return out

    def segment_done(self, view: torch.Tensor) -> None:
        if self.reducing():
            off = (view.data_ptr() - self.buffer.data_ptr()) // self.buffer.element_size()
            self.works.append((off, off + view.numel(), _all_reduce(view, async_op=True), None))

    def table_levels_done(self, table: torch.Tensor, l0: int, cnt: int) -> None:
        """Levels [l0, l0+cnt) of the table gradient ([L*T, 2] view of the buffer) are final: start their reduction."""
        if not self.reducing():
            return
        T, ns = getattr(self, "table_rows", 0), self.sparse_levels
        lo, hi = l0, l0 + cnt
        if ns and lo < ns:  # the sparse levels of this group: one compact message (needs the whole sparse range in one group)
            if lo != 0 or hi < ns:
                raise RuntimeError("sparse hash levels must be finished by the first level group")
            compact = table.index_select(0, self.sparse_rows)
            off = (table.data_ptr() - self.buffer.data_ptr()) // self.buffer.element_size()
            w = _all_reduce(compact, async_op=True)
            self.works.append((off, off + ns * T * table.shape[1], w, (table, compact)))
            lo = ns
        if lo < hi:
            self.segment_done(table[lo * T:hi * T])

    def commit(self) -> None:
        self.param.grad = self.buffer
        if self.works:
            self.reduced_ptr = self.buffer.data_ptr()

    def reduced_segments(self, grad: torch.Tensor):
        """Yields (begin, end) element ranges of ``grad`` in the order their all-reduce was started, each after the consumer stream
        has been made to wait for that reduction -- so the optimizer can update segment k while segments k+1.. are still on the
        wire.  Empty when nothing was reduced early (the caller then reduces the whole gradient itself)."""
        if not self.works:
            return
        if grad.data_ptr() != self.reduced_ptr:
            raise RuntimeError("param.grad is not the buffer that was reduced (was .grad replaced after backward?)")
        works, self.works = self.works, []
        covered = 0
        for a, b, w, sparse in works:
            w.wait()
            if sparse is not None:  # scatter the reduced live rows back; the other rows of these levels are zero on every rank
                table, compact = sparse
                table.index_copy_(0, self.sparse_rows, compact)
            covered += b - a
            yield a, b
        if covered != grad.numel():
            raise RuntimeError(f"early reduction covered {covered} of {grad.numel()} gradient elements")

    def finish(self, grad: torch.Tensor) -> bool:
        """Wait for pending reductions; True if ``grad`` is already summed over the ranks."""
        return len(list(self.reduced_segments(grad))) > 0
