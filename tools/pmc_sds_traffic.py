"""HBM traffic of one STEADY-STATE SDS step per kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE;
counter_collection.csv):
    python tools/pmc_sds_traffic.py <fetch.csv> <write.csv> <steps_run> out.json [skip=2]
FETCH_SIZE / WRITE_SIZE come in KB (MI355X_MICROARCH.md, HBM section); FETCH_SIZE is doubled (the gfx950 correction of the
same section: the counter sees 32-byte requests as half of what HBM moves).  The profiled program
(tools/sds_profile_steps.py) runs `steps_run` eager steps; a step STARTS at the first of its two bilinear-resize launches
(image, mask), so the dispatches are cut there: the first `skip` steps -- weight packing, the CLIP text tower, norm bounds,
allocator warm-up -- are reported as `one_time` rows and excluded from the per-step totals, which average the remaining
steps."""
import collections
import csv
import json
import sys

fetch_csv, write_csv, steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
skip = int(sys.argv[5]) if len(sys.argv) > 5 else 2
if steps <= skip:
    raise SystemExit(f'need more than {skip} profiled steps')
MARK = 'resize_bilinear_fwd_kernel'


def load(path):
    """-> (steady: {kernel: [KB, dispatches, ns]}, warm: same) split at the start of step `skip`."""
    per = {}
    for r in csv.DictReader(open(path)):
        d = int(r['Dispatch_Id'])
        e = per.setdefault(d, [r['Kernel_Name'][:100], 0.0, int(r['End_Timestamp']) - int(r['Start_Timestamp'])])
        e[1] += float(r['Counter_Value'])
    ids = sorted(per)
    marks = [d for d in ids if MARK in per[d][0]]
    if len(marks) != 2 * steps:
        raise SystemExit(f'{path}: expected {2 * steps} {MARK} dispatches, found {len(marks)}')
    cut = marks[2 * skip]
    steady, warm = collections.defaultdict(lambda: [0.0, 0, 0]), collections.defaultdict(lambda: [0.0, 0, 0])
    for d in ids:
        k, kb, ns = per[d]
        a = steady[k] if d >= cut else warm[k]
        a[0] += kb; a[1] += 1; a[2] += ns
    return steady, warm


(f, fw), (w, ww) = load(fetch_csv), load(write_csv)
n = steps - skip
rows = []
for k in sorted(set(f) | set(w), key=lambda k: -(2 * f.get(k, [0])[0] + w.get(k, [0])[0])):
    fk, wk = f.get(k, [0.0, 0, 0]), w.get(k, [0.0, 0, 0])
    ms = max(fk[2], wk[2]) / 1e6 / n
    mb = (2 * fk[0] + wk[0]) * 1024 / 1e6 / n
    rows.append({'kernel': k, 'dispatches_per_step': round(max(fk[1], wk[1]) / n, 1),
                 'fetch_MB_per_step': round(2 * fk[0] * 1024 / 1e6 / n, 2), 'write_MB_per_step': round(wk[0] * 1024 / 1e6 / n, 2),
                 'ms_per_step': round(ms, 3), 'TB_per_s': round(mb / 1e6 / (ms * 1e-3), 2) if ms else None, 'one_time': False})
tot = sum(r['fetch_MB_per_step'] + r['write_MB_per_step'] for r in rows) * 1e6
once = []
for k in sorted(set(fw) | set(ww), key=lambda k: -(2 * fw.get(k, [0])[0] + ww.get(k, [0])[0])):
    if k in f or k in w:
        continue                                       # also runs in steady state: a per-step kernel, listed above
    fk, wk = fw.get(k, [0.0, 0, 0]), ww.get(k, [0.0, 0, 0])
    once.append({'kernel': k, 'dispatches': max(fk[1], wk[1]), 'fetch_MB': round(2 * fk[0] * 1024 / 1e6, 2),
                 'write_MB': round(wk[0] * 1024 / 1e6, 2), 'one_time': True})
json.dump({'hbm_bytes_per_step': tot, 'steps_profiled': steps, 'steps_skipped': skip, 'steps_averaged': n,
           'one_time_bytes': sum(r['fetch_MB'] + r['write_MB'] for r in once) * 1e6,
           'how': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate runs, --kernel-trace only) over tools/sds_profile_steps.py; '
                  'KB counters, FETCH doubled per the gfx950 note of MI355X_MICROARCH.md; eager steps (one dispatch per kernel node); '
                  f'steady state = steps {skip}..{steps - 1}, cut at the bilinear-resize launches',
           'kernels': rows[:40], 'one_time_kernels': once[:20]}, open(out, 'w'), indent=1)
print('HBM bytes per steady-state step: %.2f GB' % (tot / 1e9))
for r in rows[:14]:
    print(r)
