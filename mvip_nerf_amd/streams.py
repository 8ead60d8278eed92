"""A process-wide pool of side streams, by ROLE.

HIP maps streams onto a small number of hardware queues (4 by default); a stream created after those are taken shares a queue
with an earlier one, and two streams that share a queue do not run concurrently.  Measured in round 6: with six extra streams
created earlier in the process, the BASELINE configs[2] iteration -- whose SDS terms replay on streams of their own
(nerf/utils.Pretrain_Model.cal_loss) -- took 153.4 ms instead of 146.0 (`python tools/config_step_profile.py 2`; bench.py, which
had built several trainers and diffusion wrappers by then, 150-154 ms): the later objects' term streams had landed on occupied
queues.  Every consumer therefore takes its streams from here: the count is bounded by the roles (<= 3 SDS terms + the
masked-image encode + the graph-capture warm-up), whatever number of trainers, Pretrain_Model and StableDiffusion objects a
process builds, and the term streams are the first ones created.
"""
import torch

_POOL = {}


def _index(device):
    d = torch.device(device)
    return d.index if d.index is not None else torch.cuda.current_device()


def get(device, role, k=0):
    """The stream of (role, k) on `device` ('term' k = 0..2, 'encode', 'capture'); created on first use, then shared."""
    key = (_index(device), role, int(k))
    if (key[0], 'term', 0) not in _POOL:                 # the three term streams take their queues before any other role does
        for t in range(3):
            _POOL[(key[0], 'term', t)] = torch.cuda.Stream(device=torch.device('cuda', key[0]))
    s = _POOL.get(key)
    if s is None:
        s = _POOL[key] = torch.cuda.Stream(device=torch.device('cuda', key[0]))
    return s


def term_streams(device, n):
    """Streams for n concurrently evaluated SDS terms (the first three exist from the first use of ANY role on)."""
    return [get(device, 'term', k) for k in range(int(n))]
