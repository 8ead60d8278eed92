"""Register / LDS / scratch usage of the gfx950 kernels in one object of the in-tree build:
    python tools/kernel_regs.py conv3x3 [name filter ...]
(extracts the .hip_fatbin of mvip_nerf_amd/lib/obj/<stem>.*.o, unbundles the gfx950 code object, reads the AMDGPU metadata
notes).  A build-container tool: needs /opt/rocm/lib/llvm/bin."""
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = '/opt/rocm/lib/llvm/bin'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def demangle(n):
    for tool in (os.path.join(LLVM, 'llvm-cxxfilt'), 'c++filt'):
        try:
            return subprocess.run([tool, n], capture_output=True, text=True).stdout.strip() or n
        except FileNotFoundError:
            continue
    return n


def main():
    stem, filters = sys.argv[1], sys.argv[2:]
    obj = sorted(glob.glob(os.path.join(ROOT, 'mvip_nerf_amd', 'lib', 'obj', stem + '.*.o')))[-1]
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, 'fat.bin'), os.path.join(d, 'k.co')
        subprocess.run([os.path.join(LLVM, 'llvm-objcopy'), '--dump-section', f'.hip_fatbin={fat}', obj], check=True)
        subprocess.run([os.path.join(LLVM, 'clang-offload-bundler'), '--type=o', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950',
                        f'--input={fat}', f'--output={co}', '--unbundle'], check=True)
        notes = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', co], capture_output=True, text=True).stdout
    for k in re.split(r'\n\s+- \.agpr_count', notes)[1:]:
        name = re.search(r'\.name:\s+(\S+)', k)
        if not name:
            continue
        dem = demangle(name.group(1))
        if filters and not any(f in dem for f in filters):
            continue
        g = lambda f: (re.search(r'\.' + f + r':\s+(\d+)', k) or [None, '?'])[1]
        ag = re.match(r':\s+(\d+)', k)
        print(dem[:90].ljust(90), 'agpr', ag.group(1) if ag else '?', 'vgpr', g('vgpr_count'), 'sgpr', g('sgpr_count'), 'spill',
              g('vgpr_spill_count'), 'scratch', g('private_segment_fixed_size'), 'lds', g('group_segment_fixed_size'))


if __name__ == '__main__':
    main()
