"""Host logic of the multi-GPU path (one process per GPU, torch.distributed; backend "nccl" is RCCL
on ROCm, "gloo" in the CPU tests).  No device code here, so it is exercised on CPU with gloo.

The hot path shards by RAY: every per-step ray set is split rank r -> indices r::world, each rank
renders and back-propagates its shard, and the only data-path collectives per step are
  * one all_gather of the rendered masked colours (<= 139 KB at 378x504) so every rank can run the
    image-space SDS term on the assembled frame, and
  * one all_reduce(sum) of the flat 1,191,688-float gradient bucket of both MLPs (4.77 MB).
"""
import torch


def shard(idx, rank, world):
    """Strided shard: rank r owns positions r, r+world, ...  (equal sizes up to 1)."""
    return idx if world == 1 else idx[rank::world].contiguous()


def shard_sizes(n, world):
    return [len(range(r, n, world)) for r in range(world)]


def unshard_order(n, world):
    """Permutation p with cat([x[r::world] for r])[p] == x."""
    order = torch.cat([torch.arange(r, n, world) for r in range(world)])
    inv = torch.empty_like(order)
    inv[order] = torch.arange(n)
    return inv


def all_gather_ragged(local, n_total, rank, world, dist):
    """Gather strided shards of a [n_r, C] tensor into [n_total, C] in ORIGINAL order.  The local
    shard keeps its autograd history (the gathered copies of other ranks are constants), which is
    exactly what is needed: d loss / d local shard is formed on every rank."""
    if world == 1:
        return local
    sizes = shard_sizes(n_total, world)
    mx = max(sizes)
    pad = local.detach()
    if pad.shape[0] < mx:
        pad = torch.cat([pad, pad.new_zeros((mx - pad.shape[0],) + tuple(pad.shape[1:]))], 0)
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad.contiguous())
    parts = [p[:s] for p, s in zip(parts, sizes)]
    parts[rank] = local
    return torch.cat(parts, 0)[unshard_order(n_total, world).to(local.device)]


def block_bounds(n, rank, world):
    """Contiguous block [lo, hi) of rank `rank` when n items are cut into `world` blocks of ceil(n / world) (the last
    ones may be short or empty).  Contiguous rather than strided: a block of a frame's rays is a run of whole image rows,
    so the rank's ray rows, its chunks and its slice of the gathered image are dense."""
    per = -(-n // world)
    lo = min(rank * per, n)
    return lo, min(lo + per, n)


def all_gather_blocks(local, n_total, rank, world, dist):
    """Every rank holds rows block_bounds(n_total, rank, world) of an [n_total, ...] tensor; returns the whole tensor on
    every rank with ONE all_gather of equal (zero-padded) blocks -- the frame assembly of the ray-sharded render
    (STRONG scaling: one frame's rays over all GPUs, DS_NeRF/run.py:1131 batchify_rays is the loop being split).  No
    autograd: inference path."""
    if world == 1:
        return local
    per = -(-n_total // world)
    pad = local.detach().contiguous()
    if pad.shape[0] < per:
        pad = torch.cat([pad, pad.new_zeros((per - pad.shape[0],) + tuple(pad.shape[1:]))], 0)
    out = pad.new_empty((world * per,) + tuple(pad.shape[1:]))
    try:
        dist.all_gather_into_tensor(out, pad)
    except (RuntimeError, NotImplementedError, AttributeError):       # a backend without the flat form
        parts = list(out.view((world, per) + tuple(pad.shape[1:])).unbind(0))
        dist.all_gather(parts, pad)
    return out[:n_total]


class FlatGradBucket:
    """One contiguous fp32 bucket for all parameters: grads are copied in, all-reduced once, and
    handed back as views (so the optimizer reads the reduced values without another copy).
    A parameter whose gradient is None on this rank contributes zeros and receives the reduced value (it may be
    non-None on another rank); a parameter that is None on EVERY rank therefore gets a zero gradient here where
    single-process training would skip it in Adam.  The trainer always back-propagates through both MLPs, so this
    does not occur on the hot path."""

    def __init__(self, params):
        self.params = list(params)
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n, device=self.params[0].device, dtype=torch.float32)
        self.views, o = [], 0
        for p in self.params:
            self.views.append(self.flat[o:o + p.numel()].view_as(p))
            o += p.numel()

    def all_reduce(self, dist, world):
        if world == 1:
            return
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
            else:
                v.copy_(p.grad)
        dist.all_reduce(self.flat)
        for p, v in zip(self.params, self.views):
            p.grad = v
