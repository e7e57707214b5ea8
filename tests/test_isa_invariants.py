"""Properties of the compiled gfx950 code that the kernels' hand-written synchronisation relies on, checked on the ISA hipcc
emits for this tree (hipcc cross-compiles without a GPU: a few seconds).  Not a GPU test.

ADVICE r05 (medium): `rowidct_p_pers_kernel` (pygpa_amd/csrc/gpa_unwrap_rowpers.hip) guards the LDS landing zone of the next
row pair's LDS-DMA with a hand-counted `s_waitcnt vmcnt(2 NQ)` at the top of its loop: "everything but the 2 NQ youngest
vector-memory operations has completed".  That is right only while the youngest operations of an iteration are exactly its
2 NQ = 8 16-byte stores.  A toolchain that splits those stores or spills a register inside the loop (scratch accesses count in
vmcnt) would make the wait too loose and the transform could read a landing zone the DMA is still writing -- silently.  This test
fails the build instead."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get('HIPCC') or shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-fno-gpu-rdc', '-ffp-contract=fast', '-fno-slp-vectorize',
         '-Wno-unused-result', '-Wno-unused-value', '--cuda-device-only', '-S']
VM = re.compile(r'^\s*((?:global|buffer|scratch|flat)_[a-z0-9_]+)')


def _isa(tu, tmp_path):
    if not os.path.exists(HIPCC):
        pytest.skip('hipcc not found')
    out = str(tmp_path / (tu + '.s'))
    r = subprocess.run([HIPCC] + FLAGS + [os.path.join(ROOT, 'pygpa_amd', 'csrc', tu + '.hip'), '-o', out],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return open(out).read()


def _function(text, needle):
    """the lines of the first function whose mangled name contains `needle`, and its kernel descriptor block"""
    m = re.search(r'^(_Z\w*%s\w*):' % needle, text, re.M)
    assert m, 'kernel %s not found in the ISA' % needle
    name = m.group(1)
    body = text[m.end():text.index('.Lfunc_end', m.end())]
    desc = text[text.index('.amdhsa_kernel ' + name):]
    desc = desc[:desc.index('.end_amdhsa_kernel')]
    return name, body.splitlines(), desc


def test_persistent_row_kernel_vmcnt_invariant(tmp_path):
    NQ = 4
    name, lines, desc = _function(_isa('gpa_unwrap_rowpers', tmp_path), 'rowidct_p_pers_kernel')
    assert re.search(r'\.amdhsa_private_segment_fixed_size\s+0\b', desc), 'the kernel uses scratch: spills count in vmcnt'
    assert not any('scratch_' in ln for ln in lines)
    # the loop header: the label in front of the hand-written wait, followed by the LDS-DMA of the next pair
    hdr = [i for i, ln in enumerate(lines) if re.match(r'^\.LBB\d+_\d+:.*Loop Header', ln)
           and any('global_load_lds_dwordx4' in x for x in lines[i:i + 60])]
    assert len(hdr) == 1, 'expected ONE loop around the LDS-DMA, found %d' % len(hdr)
    h = hdr[0]
    first = next(ln for ln in lines[h + 1:] if ln.strip() and not ln.strip().startswith(';'))
    assert re.match(r'\s*s_waitcnt vmcnt\(%d\)' % (2 * NQ), first), first
    label = lines[h].split(':')[0]
    # the rotated loop: header -> DMA, loads of the previous direction -> branch to the body block, which ends in the stores
    # and falls through into the header again
    body_lbl = [i for i, ln in enumerate(lines) if re.match(r'^\.LBB\d+_\d+:.*in Loop: Header=%s\b' % label.lstrip('.L'), ln) and i < h]
    assert body_lbl, 'loop body block not found'
    b = body_lbl[0]
    body_ops = [VM.match(ln).group(1) for ln in lines[b:h] if VM.match(ln)]
    assert body_ops[-2 * NQ:] == ['global_store_dwordx4'] * (2 * NQ), body_ops[-12:]
    assert body_ops.count('global_store_dwordx4') == 2 * NQ and not [o for o in body_ops if o.startswith('scratch')], body_ops
    back = next(i for i in range(h, len(lines)) if re.match(r'\s*s_branch\s+%s\b' % lines[b].split(':')[0], lines[i]))
    head_ops = [VM.match(ln).group(1) for ln in lines[h:back] if VM.match(ln)]
    assert set(head_ops) == {'global_load_lds_dwordx4', 'global_load_dwordx4'}, head_ops
    assert head_ops.count('global_load_dwordx4') == 2 * NQ
    # in program order the DMA precedes the loads, and nothing but the loads and the stores is younger than it
    assert head_ops.index('global_load_dwordx4') > max(i for i, o in enumerate(head_ops) if o == 'global_load_lds_dwordx4')


@pytest.mark.parametrize('kernel', ['rowidct_p_halfpers_kernelILi13E', 'rowidct_p_halfpers_kernelILi14E',
                                    'rowdct_halfpers_kernelILi13E', 'rowdct_halfpers_kernelILi14E'])
def test_persistent_half_length_kernels_vmcnt_invariant(kernel, tmp_path):
    """round 6: the persistent half-length row kernels (gpa_unwrap_rowhalfpers.hip) guard their LDS landing zone the same way --
    `s_waitcnt vmcnt(8)` at the top of the row loop, i.e. "everything older than the previous row's (at least) eight stores is
    done".  What must hold: no scratch anywhere in the kernel (a spill inside the loop would be a vector-memory operation the
    count does not know), the hand-written wait is the first instruction of the loop, the LDS-DMA of the next row is requested
    before every other vector-memory operation of the iteration (hipcc's own waits for those then never fall short of the DMA),
    and an iteration ends in at least eight stores with nothing but stores after the last load."""
    global FLAGS
    saved = FLAGS
    FLAGS = [f for f in FLAGS if not f.startswith('-ffp-contract')] + ['-ffp-contract=off']   # as pygpa_amd/build.py compiles it
    try:
        text = _isa('gpa_unwrap_rowhalfpers', tmp_path)
    finally:
        FLAGS = saved
    name, lines, desc = _function(text, kernel)
    assert re.search(r'\.amdhsa_private_segment_fixed_size\s+0\b', desc), 'the kernel uses scratch: spills count in vmcnt'
    assert not any('scratch_' in ln for ln in lines)
    hdr = [i for i, ln in enumerate(lines) if re.match(r'^\.LBB\d+_\d+:.*Loop Header', ln)
           and any('global_load_lds_dwordx4' in x for x in lines[i:i + 120])]
    assert len(hdr) == 1, 'expected ONE loop around the LDS-DMA, found %d' % len(hdr)
    h = hdr[0]
    # (the loop opens with a register copy -- the opaque thread index; the wait must come before any barrier or memory operation)
    first = next(ln for ln in lines[h + 1:] if VM.match(ln) or re.match(r'\s*(s_barrier|s_waitcnt vmcnt)', ln))
    assert re.match(r'\s*s_waitcnt vmcnt\(8\)', first), first
    # program order of the vector-memory operations from the loop header to the end of the function (the loop is the last thing
    # the kernel does): DMA first, then loads, then only stores
    ops = [VM.match(ln).group(1) for ln in lines[h:] if VM.match(ln)]
    dma = [i for i, o in enumerate(ops) if o == 'global_load_lds_dwordx4']
    loads = [i for i, o in enumerate(ops) if o.startswith('global_load') and o != 'global_load_lds_dwordx4']
    stores = [i for i, o in enumerate(ops) if o.startswith('global_store')]
    assert len(dma) == 8 and dma == list(range(8)), ops[:12]
    assert stores and (not loads or min(stores) > max(loads)), ops
    # (hipcc rotates the loop: the block that ends an iteration may sit in front of the header; count over the whole kernel)
    allops = [VM.match(ln).group(1) for ln in lines if VM.match(ln)]
    assert sum(o in ('global_store_dwordx4', 'global_store_dword') for o in allops) >= 8, allops
