"""View-sharded evaluation of the SDS terms of one second-stage iteration (host logic only: torch.distributed
calls on whatever backend the caller initialised -- "nccl" = RCCL on the GPU box, "gloo" in the CPU tests).

The reference evaluates the <= 7 diffusion-prior terms of an iteration one after the other on ONE device
(DS_NeRF/nerf/utils.py:280-302 dispatching to DS_NeRF/guidance/sd_utils.py:120-599): RGB, the <= 5 neighbour
views of the collaborative term, the normal map.  They are independent UNet / VAE evaluations, so here every
rank evaluates the terms it owns and only their RESULTS travel:

  phase 1  "latent" terms (the collaborative term's views except the last: forward only, each yields its share
           w (eps_hat - eps) of the accumulated latent gradient, DS_NeRF/guidance/sd_utils.py:575) -> ONE
           all_reduce(sum) of a [1, 4, 64, 64] tensor (64 KB), started asynchronously;
  phase 2  "image" terms (RGB, normal, the collaborative term's LAST view -- the only one whose graph receives a
           gradient in the reference, sd_utils.py:597-599): the owner back-propagates the term to its input image
           and broadcasts d term / d image (<= 3 broadcasts of one frame each);
  phase 3  every rank adds  sum_terms <d_image, image>  to its loss.  `image` is the frame assembled by all_gather
           whose LOCAL ray shard carries autograd history, so backward() leaves exactly this rank's share of the
           parameter gradient, and the usual single all_reduce of the flat gradient bucket sums the shares.

Ownership is round-robin over the terms in the order given (callers list the expensive image terms first).
With world == 1 the same code evaluates everything locally: that is what the tests compare against.
"""
from dataclasses import dataclass
from typing import Callable, Optional

import torch


@dataclass
class Term:
    name: str
    phase: int                                  # 1: latent share, 2: image gradient
    run: Callable                               # phase 1: run() -> latent tensor; phase 2: run(latent_sum) -> d_image
    image: Optional[torch.Tensor] = None        # phase 2: the assembled input frame (autograd on the local shard)
    latent_shape: tuple = (1, 4, 64, 64)
    needs_latent_sum: bool = False


def owner_of(index, world):
    return index % world


def evaluate(terms, rank, world, dist, device):
    """Returns the surrogate loss  sum_{image terms} <d term / d image, image>  (a scalar with autograd history
    through the local shards), having run only the terms owned by `rank`."""
    owners = [owner_of(k, world) for k in range(len(terms))]
    latent_terms = [k for k, t in enumerate(terms) if t.phase == 1]
    # ---- phase 1: latent shares of the terms this rank owns, summed over ranks --------------------------------
    latent_sum, work = None, None
    if latent_terms:
        latent_sum = torch.zeros(terms[latent_terms[0]].latent_shape, device=device, dtype=torch.float32)
        for k in latent_terms:
            if owners[k] == rank:
                with torch.no_grad():
                    latent_sum += terms[k].run().reshape(latent_sum.shape).float()
        if world > 1:
            # overlapped with this rank's independent image terms on RCCL; the gloo transport (CPU tests, and the
            # several-ranks-on-one-GPU debug mode, where an async device-tensor collective faulted) runs it in place
            if latent_sum.is_cuda and dist.get_backend() != 'nccl':
                torch.cuda.synchronize(device)
                dist.all_reduce(latent_sum)
            else:
                work = dist.all_reduce(latent_sum, async_op=True)
    # ---- phase 2: image gradients; terms that do not need the latent sum run under the all_reduce ---------------
    grads = {}
    order = [k for k, t in enumerate(terms) if t.phase == 2]
    order.sort(key=lambda k: terms[k].needs_latent_sum)
    for k in order:
        if owners[k] != rank:
            continue
        if terms[k].needs_latent_sum and work is not None:
            work.wait()
            work = None
        share = None
        if terms[k].needs_latent_sum:
            # no phase-1 term at all (the collaborative term with ONE neighbour view): the other views' shares sum to
            # zero, which is what the single-process loop starts from (DS_NeRF/guidance/sd_utils.py:442)
            share = latent_sum if latent_sum is not None else torch.zeros(terms[k].latent_shape, device=device,
                                                                           dtype=torch.float32)
        grads[k] = terms[k].run(share).detach()
    if work is not None:
        work.wait()
    loss = None
    for k, t in enumerate(terms):
        if t.phase != 2:
            continue
        d = grads.get(k)
        if world > 1:
            if d is None:
                d = torch.empty(t.image.shape, device=device, dtype=torch.float32)
            d = d.contiguous().float()
            dist.broadcast(d, src=owners[k])
        term = (d * t.image).sum()
        loss = term if loss is None else loss + term
    return loss
