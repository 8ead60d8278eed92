import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for th in (8, 16, 32, 64, 128):
    r = bench.cpu_baseline(n_rays=512, threads=th)
    print(th, r['value'], r['sample'], flush=True)
