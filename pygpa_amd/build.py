"""Build libgpa_hip.so (gfx950) in-tree with hipcc.

    python -m pygpa_amd.build [--force] [--jobs N]

hipcc cross-compiles for gfx950 without a GPU.  Objects go to
pygpa_amd/csrc/_build/, the shared library to pygpa_amd/libgpa_hip.so (git-ignored,
but it travels with the source tree to the GPU box).
"""
import argparse
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
BUILD = os.path.join(CSRC, '_build')
LIB = os.path.join(HERE, 'libgpa_hip.so')
SOURCES = ['gpa_sweep.hip', 'gpa_passb_shared.hip', 'gpa_sweep_ext.hip', 'gpa_reconstruct.hip', 'gpa_unwrap.hip', 'gpa_unwrap_rows.hip', 'gpa_unwrap_rowhalf.hip', 'gpa_unwrap_pqdct.hip', 'gpa_unwrap_rowpers.hip', 'gpa_unwrap_rowhalfpers.hip', 'gpa_unwrap_cols.hip', 'gpa_unwrap_colstream.hip', 'gpa_unwrap_stencil.hip', 'gpa_unwrap_generic.hip',
           'gpa_unwrap_tables.hip', 'gpa_dft2.hip', 'gpa_gaussfft.hip', 'gpa_warp.hip', 'gpa_tiles.hip', 'gpa_peaks.hip', 'gpa_api.hip', 'gpa_api_tables.hip', 'gpa_api_sweep.hip', 'gpa_api_unwrap.hip', 'gpa_api_driver.hip',
           'gpa_api_tiles.hip', 'gpa_api_warp.hip', 'gpa_api_spectral.hip']
ARCH = 'gfx950'
# -fvisibility=hidden: the dynamic symbol table is the C ABI of include/gpa_hip.h (its declarations sit inside a visibility
# pragma) and nothing else -- the helpers the entry-point files share cannot be interposed by another library of the process
FLAGS = ['-O3', '-std=c++17', '-fPIC', '--offload-arch=' + ARCH, '-fno-gpu-rdc', '-fvisibility=hidden',
         '-Wno-unused-result', '-Wno-unused-value', '-ffp-contract=fast', '-fno-slp-vectorize']


# per-file flags, after FLAGS (the later one wins).  The half-length row kernels of the unwrap are compiled WITHOUT floating-point
# contraction: under -ffp-contract=fast hipcc fuses products into sums differently in the one-row-per-workgroup kernels and in
# the persistent ones, and one of the persistent kernels' fusions doubles the error the f32 solve leaves in its smoothest modes
# (median over 12 problems at 64 x 16384: 1.6e-3 of max|phi| against 7.2e-4; off: 7.6e-4 in both -- profiles/r06_rowhalf_pers.txt).
# Unfused, the two families compute the same bits, which tests/test_gpu_unwrap_long.py holds them to.  (A file-scope
# `#pragma clang fp contract(off)` does not do it: the back end keeps fusing under the command line's `fast`.)
EXTRA_FLAGS = {'gpa_unwrap_rowhalf.hip': ['-ffp-contract=off'], 'gpa_unwrap_rowhalfpers.hip': ['-ffp-contract=off']}


def _hipcc():
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found (set HIPCC or install ROCm)')


def _deps(src):
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    hdrs.append(os.path.join(os.path.dirname(HERE), 'include', 'gpa_hip.h'))
    return [src] + hdrs


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, jobs=None, verbose=True):
    hipcc = _hipcc()
    os.makedirs(BUILD, exist_ok=True)
    jobs = jobs or min(len(SOURCES), os.cpu_count() or 1)
    objs, todo = [], []
    for name in SOURCES:
        src = os.path.join(CSRC, name)
        obj = os.path.join(BUILD, name.replace('.hip', '.o'))
        objs.append(obj)
        if force or _stale(obj, _deps(src)):
            todo.append((src, obj))

    def compile_one(item):
        src, obj = item
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(os.path.basename(src), []) + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed for %s:\n%s\n%s' % (src, r.stdout, r.stderr))
        return r.stderr

    # objects of translation units that no longer exist (or of one-off variant builds) do not belong to this tree
    for f in os.listdir(BUILD):
        if f.endswith('.o') and os.path.join(BUILD, f) not in objs:
            os.remove(os.path.join(BUILD, f))
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        for warn in ex.map(compile_one, todo):
            if warn and verbose:
                print(warn)
    if todo or force or _stale(LIB, objs):
        cmd = [hipcc, '-shared', '-fPIC', '--offload-arch=' + ARCH, '-fno-gpu-rdc', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n%s\n%s' % (r.stdout, r.stderr))
    return LIB


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--force', action='store_true')
    ap.add_argument('--jobs', type=int, default=None)
    a = ap.parse_args()
    print(build(force=a.force, jobs=a.jobs))
