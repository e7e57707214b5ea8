"""bench.py's host logic on the CPU: weak-scaling image shapes, the byte / flop models, and that `--gpus N`
without a launcher starts N rank processes itself (here, without a GPU, they must fail loudly -- no CPU fallback)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_weak_shapes_keep_pixels_per_gpu():
    import bench
    assert bench.weak_shape(1, 4096) == (4096, 4096)
    assert bench.weak_shape(2, 4096) == (8192, 4096)
    assert bench.weak_shape(4, 4096) == (8192, 8192)            # BASELINE configs[3]
    assert bench.weak_shape(8, 4096) == (16384, 16384)          # BASELINE configs[4]'s image (2 x 4096^2 per GPU)
    for w in range(1, 8):
        a, b = bench.weak_shape(w, 512)
        assert a * b == w * 512 * 512 and a >= b


def test_kernel_models_add_up():
    import bench
    m = bench.kernel_models(4096, 4096, 4096, 4096, 3, 16, 12, 4, (9, 8))
    px = 4096 * 4096
    assert m['passA_kernel']['bytes'] == px * (4 + 2 * 4 * 12)
    assert m['passB_kernel']['bytes'] == px * (2 * 4 * 12 + 2 * 4 * 3)
    # the shared-forward pass B: same algorithmic bytes and flops, fewer executed (12 forward + 48 inverse transforms per row)
    sh = m['passB_shared_kernel']
    assert sh['bytes'] == m['passB_kernel']['bytes'] and sh['flops'] == m['passB_kernel']['flops']
    assert 0.6 < sh['executed_flops'] / sh['flops'] < 0.7
    assert m['colsolve_kernel']['bytes'] == 2 * 4 * px and m['pq_kernel']['bytes'] == 3 * 4 * px
    # nominal flops of pass B: 48 lock-ins x 4096 rows x two 4096-point FFTs
    assert 1.0 <= m["passB_kernel"]["flops"] / (48 * 4096 * 2 * 5 * 4096 * 12) < 1.2   # + carrier / filter multiplies
    s = bench.survey_bytes(4096, 4096, 3, 16, 4, (10, 10))
    assert s['total'] // px == 3192                              # 3192 B per pixel at C3


def test_gpus_flag_spawns_and_fails_loudly_without_gpu():
    """no WORLD_SIZE in the environment + --gpus 2: bench.py must start the ranks itself (torch.distributed.run) and
    pass their failure on -- on this GPU-less host every rank dies at torch.cuda.set_device / the missing device"""
    import torch
    if torch.cuda.is_available():
        return
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0', '--size', '256',
                        '--backend', 'gloo', '--share-device'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.decode().splitlines() if ln.startswith('{')]     # no JSON line from a run that did not happen
    err = r.stderr.decode()
    assert 'torch.distributed' in err or 'ChildFailedError' in err or 'cuda' in err.lower()


def test_counters_are_refused_on_other_kernel_sources(tmp_path, monkeypatch):
    """profiles/counters.json carries the hash of the kernel sources it was measured on; bench.py uses it only on a tree
    with the same hash (the bench must not trust counters of another build)"""
    import json
    import bench
    h = bench.csrc_sha256()
    assert len(h) == 16 and h == bench.csrc_sha256()
    (tmp_path / 'profiles').mkdir()
    f = tmp_path / 'profiles' / 'counters.json'
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    monkeypatch.setattr(bench, 'csrc_sha256', lambda: h)
    f.write_text(json.dumps({'_meta': {'commit': 'abc', 'csrc_sha256': h}, 'passB_shared_kernel': {'hbm_bytes': 1}}))
    assert bench.load_counters()['passB_shared_kernel']['hbm_bytes'] == 1
    f.write_text(json.dumps({'_meta': {'commit': 'abc', 'csrc_sha256': 'somethingelse'}, 'passB_shared_kernel': {'hbm_bytes': 1}}))
    c = bench.load_counters()
    assert list(c) == ['_stale'] and 'not used' in c['_stale']
