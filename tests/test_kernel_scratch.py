"""Zero scratch on every shipped kernel (VERDICT r4 task 5): the gfx950 code objects of the in-tree build are unbundled
(what tools/kernel_regs.py does) and every kernel's AMDGPU metadata must say private_segment_fixed_size == 0 and
vgpr_spill_count == 0 -- a spilled register or a dynamically indexed private array inside a hot loop is a memory round trip
per iteration that no profile of the source shows.  The allow-list names the one-time PACK kernels (run once per weight
tensor, outside every step).  A CPU test: it reads the objects `python -m mvip_nerf_amd.csrc.build` left in lib/obj."""
import glob
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'

# kernels allowed to use scratch: weight packers, launched once per frozen weight tensor (48 B each: the fp16 pair builder's
# small private arrays), never inside a render, a training iteration or an SDS step
ALLOW = ('mvip::cv_pack_kernel(', 'mvip::gm_pack_kernel(', '_ZN4mvip14cv_pack_kernelE', '_ZN4mvip14gm_pack_kernelE')
# kernels whose scratch accesses must all lie OUTSIDE every loop (checked on the disassembly: no backward branch spans one).
# mlp_wgrad_kernel<1, 32, true> holds 8 x 2 accumulator tiles = all 256 AGPRs + 256 VGPRs of operands; the allocator parks a dozen
# values that live across the stage loop (prologue -> flush, and the once-per-workgroup odd last stage) in scratch.  Attempts to
# recompute them instead moved three of the spills INTO the loop header; as it stands the stage loop touches no scratch.
ALLOW_OUTSIDE_LOOPS = ('void mvip::mlp_wgrad_kernel<1, 32, true>(', '_ZN4mvip16mlp_wgrad_kernelILi1ELi32ELb1EEE')
MAX_SPILLED_OUTSIDE_LOOPS = 16


def _kernels(obj):
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, 'fat.bin'), os.path.join(d, 'k.co')
        r = subprocess.run([os.path.join(LLVM, 'llvm-objcopy'), '--dump-section', f'.hip_fatbin={fat}', obj], capture_output=True, text=True)
        if r.returncode != 0:
            assert "'.hip_fatbin' not found" in r.stderr, r.stderr       # a host-only source (api.hip): no device code at all
            return []
        subprocess.run([os.path.join(LLVM, 'clang-offload-bundler'), '--type=o', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950',
                        f'--input={fat}', f'--output={co}', '--unbundle'], check=True)
        notes = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', co], capture_output=True, text=True, check=True).stdout
    out = []
    for k in re.split(r'\n\s+- \.agpr_count', notes)[1:]:
        name = re.search(r'\.name:\s+(\S+)', k)
        if not name:
            continue
        g = lambda f: int((re.search(r'\.' + f + r':\s+(\d+)', k) or [None, '-1'])[1])
        out.append((name.group(1), g('private_segment_fixed_size'), g('vgpr_spill_count'), g('sgpr_spill_count')))
    return out


def _scratch_inside_loops(obj, mangled):
    """Addresses of scratch_* instructions of kernel `mangled` that lie inside a loop = are spanned by a backward branch."""
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, 'fat.bin'), os.path.join(d, 'k.co')
        subprocess.run([os.path.join(LLVM, 'llvm-objcopy'), '--dump-section', f'.hip_fatbin={fat}', obj], check=True)
        subprocess.run([os.path.join(LLVM, 'clang-offload-bundler'), '--type=o', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950',
                        f'--input={fat}', f'--output={co}', '--unbundle'], check=True)
        dis = subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', f'--disassemble-symbols={mangled}', co],
                             capture_output=True, text=True, check=True).stdout
    # control-flow graph of the kernel from the disassembly; "inside a loop" = in a block of a strongly connected component
    # that has a cycle (a backward branch alone proves nothing: block placement puts shared tails before their users)
    insts = []                                           # (address, opcode, branch target or None)
    for line in dis.split('\n'):
        m = re.match(r'\s+(\S+)\s+(.*?)\s*//\s*([0-9A-Fa-f]+):', line)
        if not m:
            continue
        op, args, addr = m.group(1), m.group(2), int(m.group(3), 16)
        target = None
        if op.startswith('s_cbranch') or op == 's_branch':
            simm = int(args.split()[0])
            simm = simm - 65536 if simm >= 32768 else simm
            target = addr + 4 + 4 * simm
        assert not op.startswith(('s_setpc', 's_swappc')), 'indirect branch: the graph below would be incomplete'
        insts.append((addr, op, target))
    assert len(insts) > 1000, 'disassembly of the kernel not found'
    addrs = [a for a, _, _ in insts]
    leaders = {addrs[0]} | {t for _, _, t in insts if t is not None}
    for k, (_, op, t) in enumerate(insts[:-1]):
        if t is not None or op == 's_endpgm':
            leaders.add(addrs[k + 1])
    block_of, blocks, cur = {}, [], None                 # address -> block index; block = [first, last instruction index]
    for k, a in enumerate(addrs):
        if a in leaders:
            blocks.append([k, k])
            cur = len(blocks) - 1
        else:
            blocks[cur][1] = k
        block_of[a] = cur
    succ = []
    for first, last in blocks:
        _, op, t = insts[last]
        out = []
        if t is not None:
            out.append(block_of[t])
        if op != 's_branch' and op != 's_endpgm' and last + 1 < len(insts):
            out.append(block_of[addrs[last + 1]])
        succ.append(out)
    # Tarjan, iterative
    n = len(blocks)
    index, low, on, comp, stack, counter = [None] * n, [0] * n, [False] * n, [None] * n, [], [0]
    for root in range(n):
        if index[root] is not None:
            continue
        work = [(root, 0)]
        while work:
            v, i = work.pop()
            if i == 0:
                index[v] = low[v] = counter[0]
                counter[0] += 1
                stack.append(v)
                on[v] = True
            recurse = False
            for j in range(i, len(succ[v])):
                w = succ[v][j]
                if index[w] is None:
                    work.append((v, j + 1))
                    work.append((w, 0))
                    recurse = True
                    break
                if on[w]:
                    low[v] = min(low[v], index[w])
            if recurse:
                continue
            if low[v] == index[v]:
                members = []
                while True:
                    w = stack.pop()
                    on[w] = False
                    members.append(w)
                    if w == v:
                        break
                cyclic = len(members) > 1 or v in succ[v]
                for w in members:
                    comp[w] = cyclic
            if work:
                u = work[-1][0]
                low[u] = min(low[u], low[v])
    scratch = [a for a, op, _ in insts if op.startswith('scratch_')]
    assert any(comp), 'no loop found in a kernel that has a stage loop'
    return [a for a in scratch if comp[block_of[a]]], len(scratch)


def _demangle(names):
    import shutil
    tool = next((t for t in (os.path.join(LLVM, 'llvm-cxxfilt'), shutil.which('c++filt')) if t and os.path.exists(t)), None)
    if tool is None:
        return {n: n for n in names}
    r = subprocess.run([tool], input='\n'.join(names) + '\n', capture_output=True, text=True)
    got = r.stdout.split('\n')
    return {n: (got[i] if i < len(got) and got[i] else n) for i, n in enumerate(names)}


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, 'clang-offload-bundler')), reason='needs the ROCm LLVM tools')
def test_no_shipped_kernel_uses_scratch():
    from mvip_nerf_amd.csrc.build import build, SOURCES
    build(verbose=False)                                 # a no-op when the objects are current
    objs = sorted(glob.glob(os.path.join(ROOT, 'mvip_nerf_amd', 'lib', 'obj', '*.o')))
    stems = {os.path.basename(o).split('.')[0] for o in objs}
    assert stems == {s[:-4] for s in SOURCES}, 'one current object per source (stale objects are removed by the build)'
    rows = []
    for o in objs:
        rows += [(os.path.basename(o).split('.')[0],) + k for k in _kernels(o)]
    assert len(rows) > 150                               # every kernel of the library was looked at
    dem = _demangle([r[1] for r in rows])
    bad = []
    obj_of = {os.path.basename(o).split('.')[0]: o for o in objs}
    for stem, name, scratch, vspill, sspill in rows:
        d = dem[name]
        assert scratch >= 0 and vspill >= 0, f'metadata not found for {d}'
        if not (scratch or vspill) or d.startswith(ALLOW):      # (spilled SGPRs live in VGPR lanes: v_writelane, no memory)
            continue
        if d.startswith(ALLOW_OUTSIDE_LOOPS) and vspill <= MAX_SPILLED_OUTSIDE_LOOPS:
            inside, n = _scratch_inside_loops(obj_of[stem], name)
            assert n > 0
            if not inside:
                continue
            bad.append(f'{stem}: {d[:140]}: {len(inside)} of {n} scratch instructions INSIDE a loop')
            continue
        bad.append(f'{stem}: {d[:140]}: scratch {scratch} B, {vspill} VGPRs / {sspill} SGPRs spilled')
    assert not bad, 'kernels with scratch:\n' + '\n'.join(bad)


def _disassembly(obj, mangled):
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, 'fat.bin'), os.path.join(d, 'k.co')
        subprocess.run([os.path.join(LLVM, 'llvm-objcopy'), '--dump-section', f'.hip_fatbin={fat}', obj], check=True)
        subprocess.run([os.path.join(LLVM, 'clang-offload-bundler'), '--type=o', '--targets=hipv4-amdgcn-amd-amdhsa--gfx950',
                        f'--input={fat}', f'--output={co}', '--unbundle'], check=True)
        return subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', f'--disassemble-symbols={mangled}', co],
                              capture_output=True, text=True, check=True).stdout


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, 'clang-offload-bundler')), reason='needs the ROCm LLVM tools')
def test_split_precision_stash_stores_are_streaming_stores():
    """Round 5: the activation / delta stash of the split-precision training kernels leaves through non-temporal stores
    (csrc/mlp_device.h::stash_store<true>; configs[2] 154 -> 144 ms, profiles/r5_nt_stash_ab.json); the exact-fp32 kernels
    keep ordinary stores.  Checked on the ISA of the in-tree build."""
    from mvip_nerf_amd.csrc.build import build
    build(verbose=False)
    obj = lambda stem: glob.glob(os.path.join(ROOT, 'mvip_nerf_amd', 'lib', 'obj', stem + '.*.o'))[0]
    count = lambda dis: (len(re.findall(r'global_store_dword\s[^\n]*\bnt\b', dis)), len(re.findall(r'global_store_dword\s', dis)))
    # the stash-writing two-wave forward <rays, STASH, 16, 4>: 632 of its stores are stash rows, one is the raw output
    nt, total = count(_disassembly(obj('mlp_fwd16_f16x3'), '_ZN4mvip4f16h28mlp_forward_f16x3_w16_kernelILb1ELb1ELi16ELi4EEEvPKfS3_S3_liPfS4_l'))
    assert nt >= 600 and total - nt <= 4, (nt, total)
    names = [k[0] for k in _kernels(obj('mlp_bwd_f16x3'))]
    delta = [n for n in names if 'mlp_delta_f16x3_kernel' in n]
    assert delta
    nt, total = count(_disassembly(obj('mlp_bwd_f16x3'), delta[0]))
    assert nt >= 100 and nt >= 0.9 * total, (nt, total)
    # exact fp32: ordinary stores
    names = [k[0] for k in _kernels(obj('mlp_bwd16'))]
    d16 = [n for n in names if 'mlp_delta16_kernel' in n]
    assert d16
    nt, total = count(_disassembly(obj('mlp_bwd16'), d16[0]))
    assert nt == 0 and total > 100, (nt, total)
