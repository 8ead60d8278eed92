"""A process-wide pool of side streams, by ROLE.

HIP maps streams onto a small number of hardware queues (4 by default); a stream created after those are taken shares a queue
with an earlier one, and two streams that share a queue do not run concurrently.  Measured in round 6: with six extra streams
created earlier in the process, the BASELINE configs[2] iteration -- whose SDS terms replay on streams of their own
(nerf/utils.Pretrain_Model.cal_loss) -- took 153.4 ms instead of 146.0 (`python tools/config_step_profile.py 2`; bench.py, which
had built several trainers and diffusion wrappers by then, 150-154 ms): the later objects' term streams had landed on occupied
queues.  Every consumer therefore takes its streams from here: the count is bounded by the roles (<= 3 SDS terms + the
masked-image encode + the graph-capture warm-up), whatever number of trainers, Pretrain_Model and StableDiffusion objects a
process builds, and they are created in one fixed order (see `get`).
"""
import torch

_POOL = {}


def _index(device):
    d = torch.device(device)
    return d.index if d.index is not None else torch.cuda.current_device()


_ORDER = (('encode', 0), ('term', 0), ('term', 1), ('term', 2), ('capture', 0))


def get(device, role, k=0):
    """The stream of (role, k) on `device` ('term' k = 0..2, 'encode', 'capture').  The first call on a device creates the whole
    fixed set in ONE order -- encode, term 0, term 1, term 2, capture: with four hardware queues and the default stream on the
    first, the masked-image encode (which overlaps a step running on the default stream or on a term stream) and two term streams
    get queues of their own; term 2 lands on the default stream's (idle while the terms run) and the capture warm-up stream on the
    encode stream's (used during captures only).  Measured: with the three term streams created BEFORE the encode stream the
    single-term step was 0.35 ms slower (fp16 mode 15.9 -> 16.3 ms): its encode branch shared a queue."""
    idx = _index(device)
    if (idx, 'encode', 0) not in _POOL:
        for r, j in _ORDER:
            _POOL[(idx, r, j)] = torch.cuda.Stream(device=torch.device('cuda', idx))
    key = (idx, role, int(k))
    s = _POOL.get(key)
    if s is None:
        s = _POOL[key] = torch.cuda.Stream(device=torch.device('cuda', idx))
    return s


def term_streams(device, n):
    """Streams for n concurrently evaluated SDS terms (the first three exist from the first use of ANY role on)."""
    return [get(device, 'term', k) for k in range(int(n))]
