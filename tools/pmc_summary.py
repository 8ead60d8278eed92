"""Per-kernel summary of a `rocprofv3 --pmc ... --kernel-trace --output-format csv` run:
usage: python tools/pmc_summary.py <counter_collection.csv> [out.json]
       python tools/pmc_summary.py <counter_collection.csv> --longest <kernel name substring>
(the second form prints one JSON object for the LONGEST dispatch of the named kernel: counter, value, duration, grid --
the dominant render launch of the FETCH_SIZE / WRITE_SIZE passes)
For every kernel name: dispatches, total duration, and the sum of each collected counter.  With
SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE it also prints MFMA-pipe utilisation =
MFMA busy cycles / (GRBM_GUI_ACTIVE x 128): GRBM_GUI_ACTIVE comes back summed over the 8 XCDs, each with
32 CUs x 4 SIMDs."""
import collections
import csv
import json
import sys

if len(sys.argv) > 3 and sys.argv[2] == '--longest':
    sel = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[3] in r['Kernel_Name']]
    best = max(sel, key=lambda r: int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    print(json.dumps({'counter': best['Counter_Name'], 'value_KB': float(best['Counter_Value']),
                      'ms': (int(best['End_Timestamp']) - int(best['Start_Timestamp'])) / 1e6, 'grid': best['Grid_Size']}))
    sys.exit(0)
rows = csv.DictReader(open(sys.argv[1]))
agg = collections.defaultdict(lambda: {'dispatches': set(), 'ns': 0, 'counters': collections.defaultdict(float)})
seen = set()
for r in rows:
    k = r['Kernel_Name'][:90]
    a = agg[k]
    did = r['Dispatch_Id']
    if (k, did) not in seen:
        seen.add((k, did))
        a['dispatches'].add(did)
        if 'End_Timestamp' in r and r['End_Timestamp']:
            a['ns'] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    a['counters'][r['Counter_Name']] += float(r['Counter_Value'])
out = []
for k, a in agg.items():
    c = dict(a['counters'])
    e = {'kernel': k, 'dispatches': len(a['dispatches']), 'total_ms': a['ns'] / 1e6, **c}
    if c.get('GRBM_GUI_ACTIVE') and 'SQ_VALU_MFMA_BUSY_CYCLES' in c:
        e['mfma_pipe_util'] = c['SQ_VALU_MFMA_BUSY_CYCLES'] / (c['GRBM_GUI_ACTIVE'] * 128)
    out.append(e)
out.sort(key=lambda e: -e.get('GRBM_GUI_ACTIVE', e['total_ms']))
if len(sys.argv) > 2:
    json.dump(out[:40], open(sys.argv[2], 'w'), indent=1)
for e in out[:14]:
    print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in e.items()}))
