"""A/B of the weight-gradient kernel's slab length (stages per workgroup) on the bench training iteration; each
setting in its own process (the knob is read once per process)."""
import json, os, subprocess, sys
out = {}
for cap in (32, 64, 128, 256):
    env = dict(os.environ, MVIP_WGRAD_STAGES=str(cap))
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'train_speed.py')], env=env,
                       capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith('{')]
    out[cap] = json.loads(line[-1]) if line else r.stderr[-300:]
    print(cap, out[cap], flush=True)
