"""Group the dispatches of a `rocprofv3 --kernel-trace --output-format csv` run by (kernel name, grid size): count, mean /
min duration, total -- which launch shapes a kernel family spends its time on.
usage: python tools/kernel_grid_groups.py <kernel_trace.csv> <name substring> [out.json]"""
import collections, csv, json, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r['Kernel_Name']]
g = collections.defaultdict(list)
for r in rows:
    g[(r['Kernel_Name'][:70], int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1), int(r.get('Grid_Size_Y', 1) or 1), int(r.get('Grid_Size_Z', 1) or 1))].append(
        (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
out = [{'kernel': k[0], 'workgroups': k[1:], 'dispatches': len(v), 'mean_us': round(sum(v) / len(v), 2), 'min_us': round(min(v), 2),
        'total_ms': round(sum(v) / 1e3, 3)} for k, v in g.items()]
out.sort(key=lambda e: -e['total_ms'])
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], 'w'), indent=1)
for e in out[:40]:
    print(json.dumps(e))
