"""Per-kernel time of one steady-state UNet forward (torch profiler, device time) for the library path and the
HIP transformer path; prints the top kernels with call counts, the total device-busy time and the wall time."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvip_nerf_amd.guidance import sd_nets                    # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    unet = sd_nets.UNet2DConditionModel().to(dev).eval()
    for p in unet.parameters():
        p.requires_grad_(False)
    x = torch.randn(2, 9, 64, 64, device=dev)
    ctx = torch.randn(2, 77, 768, device=dev)
    t = torch.tensor(417, device=dev)
    out = {}
    modes = {'library': (False, False, False), 'hip': (True, True, False), 'hip+conv1x1': (True, True, True)}
    for name in sys.argv[1:] or list(modes):
        sd_nets.USE_HIP_TRANSFORMER, sd_nets.USE_HIP_TIME_LINEARS, sd_nets.USE_MFMA_CONV1X1 = modes[name]
        with torch.no_grad():
            for _ in range(3):
                unet(x, t, encoder_hidden_states=ctx)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                unet(x, t, encoder_hidden_states=ctx)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / 5 * 1e3
            with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
                unet(x, t, encoder_hidden_states=ctx)
                torch.cuda.synchronize()
        ev = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)
        total = sum(e.device_time_total for e in ev) / 1e3
        n = sum(e.count for e in ev)
        print(f'== {name}: wall {wall:.2f} ms, device-busy {total:.2f} ms, {n} kernels')
        rows = []
        for e in ev[:28]:
            print(f'  {e.device_time_total / 1e3:8.3f} ms  x{e.count:4d}  {e.key[:110]}')
            rows.append([round(e.device_time_total / 1e3, 3), e.count, e.key[:110]])
        out[name] = {'wall_ms': round(wall, 3), 'device_busy_ms': round(total, 3), 'kernels': n, 'top': rows}
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(out, open('gpurun_out/unet_profile.json', 'w'), indent=1)


if __name__ == '__main__':
    main()
