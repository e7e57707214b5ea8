"""world_size-2 gloo test of the tile-sharded path (pygpa_amd/distributed.py) on CPU.
The device stages are injected with oracle-backed callables, so what is tested is the host
logic: windowing, round-robin sharding, the two collectives and the stitching."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _oracle_compute():
    from oracle import gpa_oracle as orc

    def gradients(win, kvecs, klists, sigma, border):
        gs = [orc.sweep(win, sigma, klists[p], kvecs[p]) for p in range(len(kvecs))]
        lock = np.stack([g['lockin'] for g in gs])
        mask = orc.interior_mask(win.shape, border)
        weights = np.abs(lock) * (mask + 1e-6)
        dudx, dudy = orc.reconstruct_gradients(kvecs, np.angle(lock), weights)
        return dudx, dudy, np.linalg.norm(weights, axis=0)

    def unwrap(dx, dy, weight, kmax):
        return orc.unwrap_prediff(dx, dy, weight, kmax=kmax)
    return gradients, unwrap


def _case():
    from pygpa_amd.synthetic import hex_kvecs, gaussian_bump_displacement, hex_moire, explicit_klists
    shape = (128, 192)
    kvecs = hex_kvecs(0.17, 7.0)
    img = hex_moire(shape, kvecs, 0.3 * gaussian_bump_displacement(shape), noise=0.05, seed=9)
    kw = np.linalg.norm(kvecs, axis=1).mean() / 2.5
    return img, kvecs, explicit_klists(kvecs, kw, 2, 2)


TILINGS = {'grid': dict(grid=(2, 2)), 'window': dict(grid=None, window=(64, 128))}


def _worker(rank, world, port, out_path, tiling='grid'):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from pygpa_amd import distributed as D
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
    img, kvecs, klists = _case()
    u = D.extract_displacement_field_tiled(img, kvecs, klists=klists, halo=20, compute=_oracle_compute(),
                                           **TILINGS[tiling])
    np.save(out_path % rank, u)
    dist.barrier()
    dist.destroy_process_group()


def test_tile_plan_windows():
    from pygpa_amd import distributed as D
    tiles, (t0, t1), wshape = D.tile_plan((128, 192), (2, 3), 10)
    assert (t0, t1) == (64, 64) and len(tiles) == 6 and wshape == (84, 84)
    seen = np.zeros((128, 192), dtype=int)
    for (i, j), (w0, w1), (o0, o1), (z0, z1) in tiles:
        assert w0.stop - w0.start == 84 and w1.stop - w1.start == 84 and (z0, z1) == (64, 64)
        assert 0 <= w0.start and w0.stop <= 128 and 0 <= w1.start and w1.stop <= 192
        assert w0.start + o0 == i * 64 and w1.start + o1 == j * 64
        seen[i * 64:(i + 1) * 64, j * 64:(j + 1) * 64] += 1
    assert np.all(seen == 1)
    with pytest.raises(ValueError):
        D.tile_plan((100, 100), (3, 3), 4)


@pytest.mark.parametrize('shape,window,halo', [((300, 520), (128, 256), 12), ((256, 256), (256, 128), 20),
                                               ((1000, 130), (64, 128), 8)])
def test_tile_plan_explicit_windows(shape, window, halo):
    """windows of a prescribed (power-of-two) shape: every pixel in exactly one tile, every tile at
    least `halo` away from its window's edges unless that edge is the image's"""
    from pygpa_amd import distributed as D
    tiles, (t0, t1), wshape = D.tile_plan(shape, None, halo, window)
    assert wshape == window
    seen = np.zeros(shape, dtype=int)
    for (i, j), (w0, w1), (o0, o1), (z0, z1) in tiles:
        assert (w0.stop - w0.start, w1.stop - w1.start) == window
        assert 0 <= w0.start and w0.stop <= shape[0] and 0 <= w1.start and w1.stop <= shape[1]
        assert w0.start + o0 == i * t0 and w1.start + o1 == j * t1 and 0 < z0 <= t0 and 0 < z1 <= t1
        for o, z, w, n in ((o0, z0, w0, shape[0]), (o1, z1, w1, shape[1])):
            assert o >= halo or w.start == 0
            assert (w.stop - w.start) - (o + z) >= halo or w.stop == n
        seen[i * t0:i * t0 + z0, j * t1:j * t1 + z1] += 1
    assert np.all(seen == 1)
    with pytest.raises(ValueError):
        D.tile_plan(shape, None, 64, (64, 64))


@pytest.mark.parametrize('tiling', ['grid', 'window'])
def test_tiled_world2_matches_world1(tmp_path, tiling):
    import torch.multiprocessing as mp
    from oracle import gpa_oracle as orc
    from pygpa_amd import distributed as D
    img, kvecs, klists = _case()
    # single process, no process group: the reference result of the SAME tiling
    u1 = D.extract_displacement_field_tiled(img, kvecs, klists=klists, halo=20, compute=_oracle_compute(),
                                            **TILINGS[tiling])
    port = _free_port()
    out = str(tmp_path / 'u_rank%d.npy')
    mp.spawn(_worker, args=(2, port, out, tiling), nprocs=2, join=True)
    for r in range(2):
        ur = np.load(out % r)
        assert np.array_equal(ur, u1), 'rank %d result differs from the single-process run' % r
    # tile interiors agree with the whole-image oracle (up to the undetermined mean of each component)
    uw = orc.extract_displacement_field(img, kvecs, klists=klists)
    d = (u1 - u1.mean(axis=(1, 2), keepdims=True)) - (uw - uw.mean(axis=(1, 2), keepdims=True))
    assert np.abs(d[:, 24:-24, 24:-24]).max() < 0.05


# ---- (peak x k-vector) sharding -------------------------------------------------------------------
def _oracle_ksharded_compute():
    from oracle import gpa_oracle as orc

    def sweep(img0, sigma, klist, kref):
        g = orc.sweep(img0, sigma, klist, kref)
        return g['lockin'], g['kidx']

    def reconstruct(lockins, kvecs, border):
        mask = orc.interior_mask(lockins.shape[1:], border)
        weights = np.abs(lockins) * (mask + 1e-6)
        dudx, dudy = orc.reconstruct_gradients(kvecs, np.angle(lockins), weights)
        return dudx, dudy, np.linalg.norm(weights, axis=0)

    def unwrap(dx, dy, weight, kmax):
        return orc.unwrap_prediff(dx, dy, weight, kmax=kmax)
    return sweep, reconstruct, unwrap


def _kworker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from pygpa_amd import distributed as D
    from pygpa_amd.synthetic import explicit_klists
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
    img, kvecs, _ = _case()
    klists = explicit_klists(kvecs, np.linalg.norm(kvecs, axis=1).mean() / 2.5, 3, 3)
    u, lock, kidx = D.extract_displacement_field_ksharded(img, kvecs, sigma=6, klists=klists, compute=_oracle_ksharded_compute())
    np.savez(out_path % rank, u=u, lock=lock, kidx=kidx)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_ksharded_matches_whole_sweep(tmp_path, world):
    """9 candidates per peak dealt over 2 / 3 ranks (gloo): the gathered selection reproduces the
    sequential sweep of the oracle exactly -- winners, lock-ins and the displacement field"""
    import torch.multiprocessing as mp
    from oracle import gpa_oracle as orc
    from pygpa_amd.synthetic import explicit_klists
    img, kvecs, _ = _case()
    klists = explicit_klists(kvecs, np.linalg.norm(kvecs, axis=1).mean() / 2.5, 3, 3)
    u_ref, parts = orc.extract_displacement_field(img, kvecs, sigma=6, klists=klists, return_parts=True)
    port = _free_port()
    out = str(tmp_path / 'k_rank%d.npz')
    mp.spawn(_kworker, args=(world, port, out), nprocs=world, join=True)
    for r in range(world):
        g = np.load(out % r)
        assert np.array_equal(g['kidx'], np.stack([x['kidx'] for x in parts['gs']]))
        assert np.array_equal(g['lock'], np.stack([x['lockin'] for x in parts['gs']]))
        assert np.array_equal(g['u'], u_ref)


# ---- stacks of frames over the ranks ----------------------------------------------------------------------------------
def _oracle_stack_compute():
    from oracle import gpa_oracle as orc

    def compute(block, kvecs, klists, sigma, border, kmax):
        return np.stack([orc.extract_displacement_field(f, kvecs, sigma=sigma, klists=klists) for f in block]) \
            if len(block) else np.empty((0, 2) + block.shape[1:])
    return compute


def _stack_case():
    img, kvecs, klists = _case()
    frames = np.stack([np.roll(img, 3 * i, axis=1)[:64, :96] * (1.0 + 0.1 * i) for i in range(5)])
    return frames, kvecs, klists


def _sworker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from pygpa_amd import distributed as D
    dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
    frames, kvecs, klists = _stack_case()
    u = D.extract_displacement_field_stack_sharded(frames, kvecs, klists=klists, dtype=np.float64, compute=_oracle_stack_compute())
    own, span = D.extract_displacement_field_stack_sharded(frames, kvecs, klists=klists, dtype=np.float64, gather=False,
                                                           compute=_oracle_stack_compute())
    np.savez(out_path % rank, u=u, own=own, span=np.array(span))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_stack_sharded_over_ranks(tmp_path, world):
    """5 frames dealt in contiguous blocks over 2 / 3 ranks (gloo): no collective on the data path, the gathered
    result is the single-process result frame for frame, blocks differ in size by at most one"""
    import torch.multiprocessing as mp
    from pygpa_amd import distributed as D
    frames, kvecs, klists = _stack_case()
    assert D.stack_shares(5, 3) == [(0, 2), (2, 4), (4, 5)] and D.stack_shares(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]
    u1 = D.extract_displacement_field_stack_sharded(frames, kvecs, klists=klists, dtype=np.float64, compute=_oracle_stack_compute())
    assert u1.shape == (5, 2, 64, 96)
    port = _free_port()
    out = str(tmp_path / 'stack_rank%d.npz')
    mp.spawn(_sworker, args=(world, port, out), nprocs=world, join=True)
    spans = []
    for r in range(world):
        g = np.load(out % r)
        assert np.array_equal(g['u'], u1), 'rank %d' % r
        a, b = (int(v) for v in g['span'])
        assert np.array_equal(g['own'], u1[a:b])
        spans.append((a, b))
    assert spans == D.stack_shares(5, world)


# ---- the image-pipelined schedule (TiledPipeline.run_stream) on CPU tensors ------------------------------------------
class OracleTileBackend:
    """test double of pygpa_amd.distributed.HipTileBackend: what TiledPipeline asks of the GPU, done by the oracle on CPU
    tensors -- so the schedule (rotating unwrap owners), the device-resident mean, the gathers into preallocated receive
    buffers, the table-driven stitch, the point-to-point hand-over run under gloo without a GPU"""

    def __init__(self):
        import torch
        self.torch = torch
        self.device = torch.device('cpu')
        self.gradients, self.unwrap = _oracle_compute()
        self.iters = {}
        self.mean = None

    def tile_sums(self, wins, rects, ntiles, max_rows, out):
        tot = 0.0
        for t in range(ntiles):
            o0, o1, z0, z1 = (int(v) for v in rects[t])
            assert z0 <= max_rows
            tot += float(wins[t, o0:o0 + z0, o1:o1 + z1].to(self.torch.float64).sum())
        out[0] = tot

    def set_mean(self, sum_t, scale):
        self.mean = float(sum_t[0]) * scale

    def tile_gradients(self, win, wpitch, kvecs, klists, sigma, border, rect, local, slot):
        o0, o1, z0, z1 = rect
        w = win.numpy() - self.mean
        dudx, dudy, wn = self.gradients(np.ascontiguousarray(w), kvecs, klists, sigma, border)
        dx = np.zeros((2,) + w.shape)
        dy = np.zeros((2,) + w.shape)
        dx[:, :, :-1] = dudx
        dy[:, :-1, :] = dudy
        t = self.torch.from_numpy
        for c in range(2):
            blk = local[c, slot]
            blk.zero_()
            blk[0, :z0, :z1] = t(dx[c, o0:o0 + z0, o1:o1 + z1])
            blk[1, :z0, :z1] = t(dy[c, o0:o0 + z0, o1:o1 + z1])
            blk[2, :z0, :z1] = t(wn[o0:o0 + z0, o1:o1 + z1])

    def tiles_to_torch(self, host):
        pass

    def torch_to_tiles(self):
        pass

    def stitch(self, c, tiles, slot_stride, table, ntiles, t0, t1, gdx, gdy, gw, concurrent=False):
        """the table-driven stitch of gpa_stitch_tiles_dev on a flat view of the gathered buffer"""
        flat = tiles.contiguous().view(-1) if tiles.is_contiguous() else None
        assert flat is not None or tiles.dim() == 4
        base = tiles.storage_offset()
        store = tiles.untyped_storage()
        whole = self.torch.empty(0, dtype=tiles.dtype).set_(store)      # the buffer the slots are strided in
        n0, n1 = gw.shape
        plane = t0 * t1
        for t in range(ntiles):
            slot, r0, c0, z0, z1 = (int(v) for v in table[t])
            blk = whole[base + slot * slot_stride: base + slot * slot_stride + 3 * plane].view(3, t0, t1)
            zx, zy = min(z1, n1 - 1 - c0), min(z0, n0 - 1 - r0)
            gdx[r0:r0 + z0, c0:c0 + zx] = blk[0, :z0, :zx]
            gdy[r0:r0 + zy, c0:c0 + z1] = blk[1, :zy, :z1]
            gw[r0:r0 + z0, c0:c0 + z1] = blk[2, :z0, :z1]

    def undistort_tiles(self, image, u, out, uinv, rects, scale=1.0):
        """the oracle's undistort_image (scipy.ndimage.map_coordinates, the reference's calls) on the whole field, of which
        only the given windows are handed out -- what gpa_undistort_image_dev does with its rects"""
        from oracle import gpa_oracle as orc
        un, ui = image.numpy().astype(np.float64), scale * u.numpy().astype(np.float64)
        full_inv = orc.invert_u_overlap(-ui)
        full = orc.undistort_image(un, ui)
        t = self.torch.from_numpy
        n0, n1 = un.shape
        for r0, c0, h, w in (rects if rects is not None else [(0, 0, n0, n1)]):
            out[r0:r0 + h, c0:c0 + w] = t(full[r0:r0 + h, c0:c0 + w])
            uinv[:, r0:r0 + h, c0:c0 + w] = t(full_inv[:, r0:r0 + h, c0:c0 + w])

    def stitch_to_tiles(self, c):
        self.stitch_waits = getattr(self, 'stitch_waits', 0) + 1

    def unwrap_start(self, c, gdx, gdy, gw, out, kmax, concurrent=False):
        out.copy_(self.torch.from_numpy(self.unwrap(gdx.numpy(), gdy.numpy(), gw.numpy(), kmax)))
        self.iters[c] = kmax

    def unwrap_wait(self, c):
        return self.iters[c]

    def unwrap_to_torch(self, c):
        pass

    def sync_device(self):
        pass

    def close(self):
        pass


def _stream_images(n):
    from pygpa_amd.synthetic import hex_kvecs, gaussian_bump_displacement, hex_moire
    shape = (128, 192)
    kvecs = hex_kvecs(0.17, 7.0)
    return [hex_moire(shape, kvecs, (0.2 + 0.1 * i) * gaussian_bump_displacement(shape), noise=0.05, seed=20 + i) for i in range(n)]


def _stream_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
    from pygpa_amd import distributed as D
    _, kvecs, klists = _case()
    images = _stream_images(4)
    pipe = D.TiledPipeline(images[0].shape, kvecs, np.stack(klists), 6, 20, kmax=10, dtype=np.float64, grid=(2, 2),
                           backend=OracleTileBackend())
    ref = []
    for img in images:                      # the unpipelined step, image by image
        pipe.load(img)
        ref.append(pipe.step().clone().numpy())
    got = {}
    iters = pipe.run_stream(images, on_result=lambda i, u: got.__setitem__(i, u.clone().numpy()))
    assert len(iters) == len(images)
    for i in range(len(images)):
        a, b = pipe.owners(i)
        assert (a, b) == ((2 * i) % world, (2 * i + 1) % world)
        assert (i in got) == (rank == a), 'the field of image %d must arrive on rank %d only' % (i, a)
        if rank == a:
            assert np.array_equal(got[i], ref[i]), 'image %d: pipelined field differs from step()' % i
    assert set(pipe.stage_s) == {'load', 'mean', 'tiles', 'gather', 'unwrap_wait', 'handover'}
    np.save(out_path % rank, np.array(sorted(got)))
    pipe.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [1, 2, 3])
def test_pipelined_stream_equals_step(tmp_path, world):
    """TiledPipeline.run_stream (unwrap of image i on the rotating owners (2 i + c) % N while every rank sweeps image
    i + 1; gather of 3 of 5 fields to each owner; component 1 handed to the owner of component 0) gives, per image, the
    field of step() bit for bit, on 1, 2 and 3 ranks"""
    import torch.multiprocessing as mp
    out = str(tmp_path / 'got_rank%d.npy')
    if world == 1:
        _stream_worker(0, 1, 0, out)
    else:
        mp.spawn(_stream_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    seen = np.concatenate([np.load(out % r) for r in range(world)])
    assert sorted(seen.tolist()) == [0, 1, 2, 3]      # every image's field arrived exactly once


def _undistort_worker(rank, world, port, out_path, dist_sync=0, force=False):
    sys.path.insert(0, ROOT)
    if dist_sync:
        os.environ['GPA_DIST_SYNC'] = str(dist_sync)
    if force:
        os.environ['GPA_DIST_FORCE_COLLECTIVES'] = '1'
    if world > 1 or force:
        import torch.distributed as dist
        dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%d' % port, rank=rank, world_size=world)
    from oracle import gpa_oracle as orc
    from pygpa_amd import distributed as D
    _, kvecs, klists = _case()
    img = _stream_images(2)[1]
    pipe = D.TiledPipeline(img.shape, kvecs, np.stack(klists), 6, 20, kmax=10, dtype=np.float64, grid=(2, 3),
                           backend=OracleTileBackend())
    pipe.load(img)
    u = pipe.step().clone().numpy()
    rec, uinv = pipe.undistort(img)
    ref_rec, ref_inv = orc.undistort_image(img, u), orc.invert_u_overlap(-u)
    assert np.array_equal(rec.numpy(), ref_rec), 'undistorted image differs from the whole-image call'
    assert np.array_equal(uinv.numpy(), ref_inv)
    np.save(out_path % rank, rec.numpy())
    pipe.close()
    if world > 1 or force:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [1, 2, 3])
def test_tile_sharded_undistort_equals_whole_image(tmp_path, world):
    """configs[4]'s Lawler-Fujita stage inside the tile pipeline (TiledPipeline.undistort): every rank inverts / resamples the
    interiors of its own tiles (6 tiles over 1, 2, 3 ranks) from the broadcast field, one all_reduce assembles them -- the
    result on EVERY rank equals the whole-image undistort_image of the oracle bit for bit"""
    import torch.multiprocessing as mp
    out = str(tmp_path / 'rec_rank%d.npy')
    if world == 1:
        _undistort_worker(0, 1, 0, out)
    else:
        mp.spawn(_undistort_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    recs = [np.load(out % r) for r in range(world)]
    for r in recs[1:]:
        assert np.array_equal(r, recs[0])


def test_dist_sync_fences_leave_the_results_alone(tmp_path):
    """GPA_DIST_SYNC=2 (drain the device and meet at a barrier around EVERY collective of TiledPipeline,
    with a log line each -- the bisecting aid for the first run on a real multi-GPU node): two gloo ranks through step() and
    the tile-sharded undistortion give the fields of the unfenced run"""
    import torch.multiprocessing as mp
    out = str(tmp_path / 'rec_sync_rank%d.npy')
    mp.spawn(_undistort_worker, args=(2, _free_port(), out, 2), nprocs=2, join=True)
    ref = str(tmp_path / 'rec_ref_rank%d.npy')
    _undistort_worker(0, 1, 0, ref)
    assert np.array_equal(np.load(out % 0), np.load(ref % 0)) and np.array_equal(np.load(out % 1), np.load(ref % 0))
    # GPA_DIST_FORCE_COLLECTIVES=1: a group of ONE rank through every collective of the N > 1 path (what the GPU test runs on
    # RCCL with the one rank a single-GPU box allows), fenced as well
    one = str(tmp_path / 'rec_forced_rank%d.npy')
    mp.spawn(_undistort_worker, args=(1, _free_port(), one, 2, True), nprocs=1, join=True)
    assert np.array_equal(np.load(one % 0), np.load(ref % 0))
