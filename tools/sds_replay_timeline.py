"""Where does the REPLAYED SDS step spend its wall time?  Reads a `rocprofv3 --kernel-trace --output-format csv` of
`python tools/sds_replay_timeline.py --run N` (N graph-replayed train_step_sd steps) and, per steady-state step (delimited by the
bilinear-resize kernel that opens each step), reports: wall span, the UNION of the kernels' busy intervals (device not idle),
the sum of the kernels' durations (two captured streams overlap), the idle time split into gaps behind a kernel boundary, and
the kernel families ordered by their share of the union.

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d <dir> -o run -- python3 tools/sds_replay_timeline.py --run 6
    python tools/sds_replay_timeline.py <dir>/.../run_kernel_trace.csv gpurun_out/r6_sds_replay_timeline.json
"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(n):
    import torch
    from mvip_nerf_amd.guidance.sd_utils import StableDiffusion
    dev = torch.device('cuda', 0)
    sd = StableDiffusion(dev, False, False, use_graphs=True)
    g = torch.Generator(device=dev).manual_seed(2)
    pred = torch.rand(1, 3, 378, 504, device=dev, generator=g).requires_grad_(True)
    mask = torch.zeros(1, 1, 378, 504, device=dev)
    mask[:, :, 137:241, 196:307] = 1
    for i in range(n + 2):
        pred.grad = None
        (1e-4 * sd.train_step_sd(1000 + i, mask, 'a stone bench in a park', pred, guidance_scale=7.5)).sum().backward()
        torch.cuda.synchronize()
    print('done', n)


def family(name):
    for key in ('conv3x3_f16x3', 'gemm5_f16x3', 'gemm_f16x3', 'gemm2_f16x3', 'attn_f16x3', 'cv_to_split', 'cv_absmax', 'gn_bwd', 'gn_moments',
                'cv_split_reduce', 'cv_im2col', 'cv_col2im', 'ln_apply', 'ln_stats', 'gm_pack', 'softmax_rows', 'linear_small', 'resize_bilinear'):
        if key in name:
            return key
    return 'mvip other' if 'mvip::' in name else 'torch / copies'


def analyse(path, out_path):
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    rows.sort(key=lambda r: r['s'])
    marks = [i for i, r in enumerate(rows) if 'resize_bilinear_fwd' in r['Kernel_Name']]
    starts = marks[::2]                                    # two resizes per step (image, mask)
    steps = []
    for a, b in zip(starts[2:-1], starts[3:]):             # skip the eager warm-ups / capture
        seg = rows[a:b]
        t0, t1 = seg[0]['s'], rows[b]['s']
        iv = sorted((r['s'], r['e'], r['Kernel_Name']) for r in seg)
        union, gaps, cur_s, cur_e, last_name = 0, [], iv[0][0], iv[0][1], iv[0][2]
        for s, e, nm in iv[1:]:
            if s > cur_e:
                union += cur_e - cur_s
                gaps.append((s - cur_e, last_name))
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
            if e >= cur_e:
                last_name = nm
        union += cur_e - cur_s
        fam = collections.defaultdict(lambda: [0, 0])
        for r in seg:
            f = fam[family(r['Kernel_Name'])]
            f[0] += r['e'] - r['s']
            f[1] += 1
        steps.append({'wall_ms': (t1 - t0) / 1e6, 'launches': len(seg), 'union_busy_ms': union / 1e6,
                      'sum_of_kernel_ms': sum(r['e'] - r['s'] for r in seg) / 1e6,
                      'idle_ms': (t1 - t0 - union) / 1e6, 'gaps': len(gaps),
                      'idle_in_gaps_over_5us_ms': sum(g for g, _ in gaps if g > 5000) / 1e6,
                      'median_gap_us': sorted(g for g, _ in gaps)[len(gaps) // 2] / 1e3 if gaps else 0.0,
                      'families_ms': {k: [round(v[0] / 1e6, 3), v[1]] for k, v in sorted(fam.items(), key=lambda kv: -kv[1][0])}})
    steps.sort(key=lambda s: s['wall_ms'])
    med = steps[len(steps) // 2]
    out = {'what': 'one steady-state hipGraph-REPLAYED train_step_sd (median of %d by wall span) from a rocprofv3 kernel trace: wall span between two '
                   'steps\' first kernels, union of the busy intervals over both captured streams, idle = wall - union (the host refills the '
                   'noise buffers and launches the graph between steps: part of the idle time is before the first kernel)' % len(steps),
           'median_step': med, 'all_walls_ms': [round(s['wall_ms'], 3) for s in steps]}
    json.dump(out, open(out_path, 'w'), indent=1)
    print(json.dumps({k: v for k, v in med.items() if k != 'families_ms'}))
    for k, v in med['families_ms'].items():
        print('   %-18s %8.3f ms  x %d' % (k, v[0], v[1]))


if __name__ == '__main__':
    if sys.argv[1] == '--run':
        run(int(sys.argv[2]))
    else:
        analyse(sys.argv[1], sys.argv[2])
