"""Steady-state per-step summary of a rocprofv3 --kernel-trace of tools/sds_profile_steps.py.

Steps are delimited by the first bilinear-resize kernel of each train_step_sd (two per step: image and
mask), so library warm-up (MIOpen find mode) and the end-of-run idle never enter the window.
usage: python tools/sds_trace_summary.py <kernel_trace.csv> [out.csv]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'upsample_bilinear2d_out' in r['Kernel_Name'] and 'backward' not in r['Kernel_Name']]
starts = marks[::2]
a, b = starts[-2], starts[-1]                 # the last complete step
seg = rows[a:b]
span = (int(rows[b]['Start_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e6
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg) / 1e6
agg = collections.defaultdict(lambda: [0, 0])
for r in seg:
    k = r['Kernel_Name'][:100]
    agg[k][0] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    agg[k][1] += 1
out = open(sys.argv[2], 'w') if len(sys.argv) > 2 else sys.stdout
w = csv.writer(out)
w.writerow([f'kernel (one steady-state train_step_sd: {len(seg)} kernels, span {span:.2f} ms, GPU-busy {busy:.2f} ms)', 'calls', 'total_ms'])
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
    w.writerow([k, v[1], f'{v[0] / 1e6:.3f}'])
