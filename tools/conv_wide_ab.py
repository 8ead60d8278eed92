"""A/B of the convolution's workgroup shape (MVIP_CONV_WIDE: 1 = eight-wave 16x32-pixel tiles where the grid allows,
0 = four-wave 8x32 tiles) on the VAE / UNet shapes and on the whole SDS step; one process per setting."""
import json, os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
for wide in (os.environ.get('MVIP_AB_MODES', '0,1').split(',')):
    env = dict(os.environ, MVIP_CONV_WIDE=wide)
    r = subprocess.run([sys.executable, os.path.join(here, 'conv_bench.py')], env=env, capture_output=True, text=True)
    for l in r.stdout.splitlines():
        if l.startswith('{'):
            d = json.loads(l)
            print('wide', wide, d['shape'], 'fused_ms', d['fused_ms'], 'TF', d['fused_TFLOPs_equiv'], 'rel', f"{d['rel_diff_vs_lib']:.1e}", flush=True)
    r = subprocess.run([sys.executable, os.path.join(here, 'sds_step_profile.py')], env=env, capture_output=True, text=True)
    for l in r.stdout.splitlines():
        if l.startswith('==') or 'conv3x3' in l:
            print('wide', wide, l, flush=True)
