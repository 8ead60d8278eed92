"""Build libmvipnerf.so (gfx950) in-tree with hipcc.  `python -m mvip_nerf_amd.csrc.build [-f]`.

Each .hip is compiled to an object (cached on a content hash of the source + headers + flags) and
the objects are linked into mvip_nerf_amd/lib/libmvipnerf.so.  hipcc cross-compiles for gfx950
without a GPU, so this runs in the build container; the .so then travels with the tree.
"""
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
EXTRA_FLAGS = os.environ.get('MVIP_EXTRA_FLAGS', '').split()      # experiments only
# A build with any -DMVIP_EXPERIMENT_* macro (timing experiments whose RESULTS ARE WRONG) never lands in lib/: it goes to
# lib_experiment/, which _lib.load() opens only when MVIP_LIB_PATH points there AND MVIP_ALLOW_EXPERIMENT_BUILD=1.
EXPERIMENT = any(f.startswith('-DMVIP_EXPERIMENT') for f in EXTRA_FLAGS)
LIB_DIR = os.path.join(PKG, 'lib_experiment' if EXPERIMENT else 'lib')
OBJ_DIR = os.path.join(LIB_DIR, 'obj')
LIB_PATH = os.path.join(LIB_DIR, 'libmvipnerf.so')

SOURCES = ['api.hip', 'rays.hip', 'composite.hip', 'sample_pdf.hip', 'mlp_pack.hip', 'mlp_fwd.hip', 'mlp_fwd16.hip', 'mlp_fwd_f16x3.hip', 'mlp_fwd16_f16x3.hip',
           'mlp_bwd.hip', 'mlp_bwd16.hip', 'mlp_bwd_f16x3.hip', 'normal_fit.hip', 'sds_elem.hip', 'group_norm.hip', 'conv3x3.hip', 'attention.hip', 'transformer.hip', 'hashgrid.hip', 'hashgrid_fused.hip', 'skinny_gemm.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-Wall',
         '-Wno-unused-function'] + EXTRA_FLAGS


def _hipcc():
    for c in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if c and os.path.exists(c):
            return c
    raise RuntimeError('hipcc not found (need ROCm with gfx950 support)')


def _headers():
    hs = [os.path.join(HERE, f) for f in sorted(os.listdir(HERE)) if f.endswith('.h')]
    hs.append(os.path.join(os.path.dirname(PKG), 'include', 'mvip_nerf.h'))
    return hs


def _digest(path, header_blob):
    h = hashlib.sha256()
    h.update(open(path, 'rb').read())
    h.update(header_blob)
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()[:16]


def build(force=False, verbose=True):
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()
    header_blob = b''.join(open(h, 'rb').read() for h in _headers())
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(HERE, s))]
    jobs, objs = [], []
    for s in srcs:
        src = os.path.join(HERE, s)
        obj = os.path.join(OBJ_DIR, f'{s[:-4]}.{_digest(src, header_blob)}.o')
        objs.append(obj)
        if force or not os.path.exists(obj):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        for old in os.listdir(OBJ_DIR):           # drop stale objects of this source
            if old.startswith(os.path.basename(src)[:-4] + '.') and os.path.join(OBJ_DIR, old) != obj:
                os.remove(os.path.join(OBJ_DIR, old))
        cmd = [hipcc] + FLAGS + ['-c', src, '-o', obj]
        if verbose:
            print('[mvip build]', os.path.basename(src), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed on {src}:\n{r.stdout}\n{r.stderr}')
        return obj

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    if jobs or force or not os.path.exists(LIB_PATH):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB_PATH] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
        if verbose:
            print('[mvip build] linked', LIB_PATH, flush=True)
    return LIB_PATH


if __name__ == '__main__':
    build(force='-f' in sys.argv)
