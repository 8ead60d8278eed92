import sys, torch
sys.path.insert(0, '.')
from tools.sds_bench import run
print(run(False, steps=3))
