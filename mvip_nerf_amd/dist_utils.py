"""Host logic of the multi-GPU path (one process per GPU, torch.distributed; backend "nccl" is RCCL
on ROCm, "gloo" in the CPU tests).  No device code here, so it is exercised on CPU with gloo.

The hot path shards by RAY: every per-step ray set is split rank r -> indices r::world, each rank
renders and back-propagates its shard, and the only data-path collectives per step are
  * one all_gather of the rendered masked colours (<= 139 KB at 378x504) so every rank can run the
    image-space SDS term on the assembled frame, and
  * one all_reduce(sum) of the flat 1,191,688-float gradient bucket of both MLPs (4.77 MB) -- since round 5 as TWO
    asynchronous halves (`OverlappedGradBuckets`): the coarse network's half is on the wire while the backward of the
    fine network's largest render (the masked set, created first and therefore back-propagated last) still runs.
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
    # The form of the collective is chosen UP FRONT from facts every rank shares (the backend's name and the API's presence), never
    # by catching an error: a rank that failed inside one collective and then entered another would desynchronise its peers,
    # and the original error would be lost.  Errors of the collective itself propagate.
    if _flat_gather_ok(dist):
        dist.all_gather_into_tensor(out, pad)
    else:
        dist.all_gather(list(out.view((world, per) + tuple(pad.shape[1:])).unbind(0)), pad)
    return out[:n_total]


def _flat_gather_ok(dist):
    """all_gather_into_tensor exists and the backend is RCCL ("nccl" on ROCm); gloo and anything else take the list form."""
    if not hasattr(dist, 'all_gather_into_tensor'):
        return False
    try:
        return str(dist.get_backend()).lower() == 'nccl'
    except (AttributeError, RuntimeError, ValueError):          # a stand-in `dist` of the tests, or no default group
        return False


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


class OverlappedGradBuckets(FlatGradBucket):
    """The same flat bucket cut into GROUPS of consecutive parameters (the trainer passes [coarse network, fine network]),
    each all-reduced ASYNCHRONOUSLY as soon as every gradient of the group is final, i.e. while the rest of
    loss.backward() still runs; `finish()` waits for the handles before the optimizer reads the gradients.

    "Final" is told by torch's post-accumulate-grad hooks: the engine sums all uses of a leaf before its AccumulateGrad node
    runs, so the hook fires once per parameter per backward.  The second-stage iteration's graph makes that useful: the
    coarse network only receives a gradient through `rgb0` of the colour batch (DS_NeRF/run.py:1023; every other render
    reaches it through detached depths, run.py:1812), whose nodes were created AFTER the masked render's and are therefore
    back-propagated BEFORE it -- the coarse half is complete while the fine network's largest backward has not started.

    Every rank issues the collectives in GROUP ORDER whatever its graph looks like (a rank whose colour shard is empty
    never completes the coarse group in a hook): a group is launched from a hook only when all earlier groups have been
    launched, everything else is launched by `finish()` in order.  Replaces the reference's nn.DataParallel gradient
    reduction (DS_NeRF/run.py:1491, :1527).  Values are those of FlatGradBucket.all_reduce (a sum over ranks of the same
    addends; tests/test_distributed_cpu.py compares the two bit for bit under gloo)."""

    def __init__(self, params, group_sizes=None):
        super().__init__(params)
        n = len(self.params)
        sizes = [n] if not group_sizes else list(group_sizes)
        assert sum(sizes) == n and all(k > 0 for k in sizes), 'group_sizes must partition the parameter list'
        self.group_of, self.bounds = [], []                  # parameter index -> group; group -> [lo, hi) in the flat bucket
        o = k = 0
        for g, cnt in enumerate(sizes):
            lo = o
            for _ in range(cnt):
                self.group_of.append(g)
                o += self.params[k].numel()
                k += 1
            self.bounds.append((lo, o))
        self.group_sizes = sizes
        self.dist = None
        self._armed = False
        self._hooks = [p.register_post_accumulate_grad_hook(lambda q, i=i: self._ready(i)) for i, p in enumerate(self.params)]
        self._reset()

    def _reset(self):
        self._pending = list(self.group_sizes)
        self._seen = [False] * len(self.params)
        self._launched = 0                                   # groups 0 .. _launched-1 are on the wire (or done)
        self._handles = []
        self._blocking = False
        self.launched_in_backward = 0                        # diagnostics (tests, bench): groups whose reduce overlapped

    def begin(self, dist, world):
        """Call before loss.backward(); with world == 1 nothing is hooked up."""
        self.abort()                                         # a backward that raised after a launch left handles behind: wait, never drop
        self._reset()
        self.dist, self._armed = dist, world > 1
        # device tensors over a transport other than RCCL (the several-ranks-on-one-GPU debug mode over gloo, where an
        # asynchronous device-tensor collective faulted -- see sds_shard.evaluate) are reduced in place, blocking; asked once
        # per backward, not inside every hook
        self._blocking = bool(self._armed and self.flat.is_cuda and str(dist.get_backend()).lower() != 'nccl')

    def abort(self):
        """Error path (the trainer's try / finally around loss.backward()): issue every collective the peers will issue -- they
        block in finish() otherwise -- wait for all of them and disarm.  The gradients of this iteration are not handed back."""
        if getattr(self, '_armed', False):
            self._armed = False
            while self._launched < len(self.group_sizes):
                self._launch(self._launched)
        for h in getattr(self, '_handles', []):
            if h is not None:
                h.wait()
        self._handles = []

    def _launch(self, g, asynchronous=True):
        lo, hi = self.bounds[g]
        k0 = sum(self.group_sizes[:g])
        for i in range(k0, k0 + self.group_sizes[g]):
            p, v = self.params[i], self.views[i]
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
        if self._blocking:
            torch.cuda.synchronize(self.flat.device)
            asynchronous = False
        self._handles.append(self.dist.all_reduce(self.flat[lo:hi], async_op=True) if asynchronous
                             else self.dist.all_reduce(self.flat[lo:hi]))
        self._launched = g + 1

    def _ready(self, i):
        if not self._armed or self._seen[i]:
            return
        self._seen[i] = True
        g = self.group_of[i]
        self._pending[g] -= 1
        # launch every complete group that is next in order (a later group may have completed first: it waits its turn)
        while self._launched < len(self.group_sizes) and self._pending[self._launched] == 0:
            self._launch(self._launched)
            self.launched_in_backward += 1

    def finish(self):
        """After loss.backward(): launch what the hooks could not (in order), wait, hand the reduced views back as .grad."""
        if not self._armed:
            return
        self._armed = False
        while self._launched < len(self.group_sizes):
            self._launch(self._launched)
        for h in self._handles:
            if h is not None:
                h.wait()
        self._handles = []
        for p, v in zip(self.params, self.views):
            p.grad = v

    def all_reduce(self, dist, world):
        """FlatGradBucket's blocking form (kept for callers that do not bracket their backward with begin / finish)."""
        if self._armed:
            return self.finish()
        return super().all_reduce(dist, world)
