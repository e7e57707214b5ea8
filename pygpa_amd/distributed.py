"""Tile-sharded displacement-field extraction over several GPUs (BASELINE
configs 4-5).  One process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on
the GPUs, "gloo" in the CPU tests) is used for exactly two collectives.

The reference has no tiling; the semantics are this build's:
  1. the image is cut into g0 x g1 tiles; every tile is processed inside a window of
     tile + 2*halo pixels (halo >= 2 sigma + 1; windows at the image border are shifted inward so
     all windows have one shape and therefore one plan);
  2. tiles are dealt round-robin to the ranks; a rank runs the sweep + per-pixel least squares
     on its windows (`gpa_extract_gradients`, no collective) and keeps the INTERIOR of the
     gradient fields dudx, dudy and of the unwrap weight -- local quantities, unlike an
     unwrapped u whose mean is undetermined per tile (phase_unwrap.py:110-114);
  3. all_gather #1 stitches the gradient tiles; the weighted unwrap is a global solve, so
     component c of u is unwrapped once by rank c % world on the stitched fields;
  4. all_gather #2 distributes the two components.
Inside a tile the lock-in wraps around the WINDOW instead of the image, so results differ from
the whole-image reference near window borders; the halo keeps that out of the interiors
(tests compare against the oracle run on the same tiling, and tile interiors against the
whole-image oracle).
"""
import numpy as np


def window_start(i, tile, halo, n):
    """first pixel of the window of tile i along an axis of length n (windows are tile + 2*halo
    long and shifted inward at the borders)"""
    return int(min(max(i * tile - halo, 0), n - (tile + 2 * halo)))


def tile_plan(shape, grid, halo):
    """list of (tile index, window slices, interior offset inside the window)"""
    n0, n1 = shape
    g0, g1 = grid
    if n0 % g0 or n1 % g1:
        raise ValueError('image shape %s is not divisible by the tile grid %s' % (shape, grid))
    t0, t1 = n0 // g0, n1 // g1
    if t0 + 2 * halo > n0 or t1 + 2 * halo > n1:
        raise ValueError('tile + 2*halo exceeds the image')
    out = []
    for i in range(g0):
        for j in range(g1):
            s0, s1 = window_start(i, t0, halo, n0), window_start(j, t1, halo, n1)
            out.append(((i, j), (slice(s0, s0 + t0 + 2 * halo), slice(s1, s1 + t1 + 2 * halo)),
                        (i * t0 - s0, j * t1 - s1)))
    return out, (t0, t1)


def _dist():
    import torch
    import torch.distributed as dist
    return torch, dist


def _initialized():
    """True when a torch.distributed process group exists.  torch is not imported here: a
    process that never imported torch.distributed cannot have a group (and the first import
    of torch on a cold machine takes minutes)."""
    import sys
    d = sys.modules.get('torch.distributed')
    return d is not None and d.is_available() and d.is_initialized()


def _all_gather_np(arr, group=None):
    """all_gather of equally shaped NumPy arrays -> list over ranks (RCCL needs device tensors)"""
    if not _initialized():
        return [arr]
    torch, dist = _dist()
    if dist.get_world_size(group) == 1:
        return [arr]
    t = torch.from_numpy(np.ascontiguousarray(arr))
    on_gpu = dist.get_backend(group) == 'nccl'
    if on_gpu:
        t = t.cuda()
    outs = [torch.empty_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(outs, t, group=group)
    return [o.cpu().numpy() if on_gpu else o.numpy() for o in outs]


def default_compute(window_shape, image_shape, nbatch, dtype, device):
    """tile-gradient and global-unwrap callables backed by libgpa_hip.so"""
    from . import _lib
    plans = {}

    def gradients(win, kvecs, klists, sigma, border):
        if 'w' not in plans:
            plans['w'] = _lib.Plan(window_shape, nbatch, dtype, device)
        return plans['w'].extract_gradients(win, kvecs, klists, sigma, border)

    def unwrap(dx, dy, weight, kmax):
        if 'g' not in plans:
            plans['g'] = _lib.Plan(image_shape, 1, dtype, device)
        return plans['g'].unwrap_prediff(dx, dy, weight, kmax=kmax)[0]
    return gradients, unwrap


def extract_displacement_field_tiled(image, kvecs, grid, sigma=None, kwscale=2.5, ksteps=3, klists=None,
                                     halo=None, kmax=10, dtype=np.float64, device=0, group=None, compute=None):
    """Tile-sharded `extract_displacement_field`.  Every rank passes the same full `image`
    (or at least its own windows' pixels) and receives the full (2, N, M) field."""
    world, rank = 1, 0
    if _initialized():
        _, dist = _dist()
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    image = np.asarray(image)
    kvecs = np.asarray(kvecs, dtype=np.float64).reshape(-1, 2)
    norms = np.linalg.norm(kvecs, axis=1)
    kw = norms.mean() / kwscale
    if sigma is None:
        sigma = int(np.ceil(1 / norms.min()))
    if klists is None:
        from .geometric_phase_analysis import _sweep_list
        klists = [_sweep_list(pk[0], pk[1], kw, kw / ksteps) for pk in kvecs]
    K = max(len(k) for k in klists)
    klists = np.stack([np.concatenate([np.asarray(k), np.repeat(np.asarray(k)[-1:], K - len(k), axis=0)]) for k in klists])
    if halo is None:
        halo = 3 * int(sigma)
    if halo < 2 * sigma + 1:
        raise ValueError('halo must be at least 2*sigma + 1')
    tiles, (t0, t1) = tile_plan(image.shape, grid, halo)
    wshape = (t0 + 2 * halo, t1 + 2 * halo)
    if compute is None:
        compute = default_compute(wshape, image.shape, len(kvecs) * K, dtype, device)
    gradients, unwrap = compute

    # --- local stage: my tiles (round robin), interiors of dudx (2), dudy (2), wnorm (1)
    per_rank = (len(tiles) + world - 1) // world
    local = np.zeros((per_rank, 5, t0, t1), dtype=dtype)
    mean = image.mean()      # the driver subtracts the IMAGE mean (geometric_phase_analysis.py:919)
    for slot, idx in enumerate(range(rank, len(tiles), world)):
        _, (w0, w1), (o0, o1) = tiles[idx]
        dudx, dudy, wn = gradients(np.ascontiguousarray(image[w0, w1] - mean), kvecs, klists, sigma, 2 * int(sigma))
        # pad the difference fields to the window shape so interiors can be cut uniformly
        dx = np.zeros((2,) + wshape, dtype=dtype)
        dy = np.zeros((2,) + wshape, dtype=dtype)
        dx[:, :, :-1] = dudx
        dy[:, :-1, :] = dudy
        local[slot, 0:2] = dx[:, o0:o0 + t0, o1:o1 + t1]
        local[slot, 2:4] = dy[:, o0:o0 + t0, o1:o1 + t1]
        local[slot, 4] = wn[o0:o0 + t0, o1:o1 + t1]

    # --- collective 1: stitch the gradient tiles
    gathered = _all_gather_np(local, group)
    n0, n1 = image.shape
    full = np.zeros((5, n0, n1), dtype=dtype)
    for idx, ((i, j), _, _) in enumerate(tiles):
        full[:, i * t0:(i + 1) * t0, j * t1:(j + 1) * t1] = gathered[idx % world][idx // world]

    # --- global unwrap, one component per rank; collective 2 distributes them
    mine = np.zeros((2, n0, n1), dtype=dtype)
    for c in range(2):
        if c % world == rank:
            mine[c] = unwrap(full[c, :, :-1], full[2 + c, :-1, :], full[4], kmax)
    parts = _all_gather_np(mine, group)
    return np.stack([parts[c % world][c] for c in range(2)])
