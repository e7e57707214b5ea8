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
With the default compute (libgpa_hip.so) everything between the upload of the image and the download
of u stays in HBM: windows are cut on the device, tile interiors are written by 2-D device copies into
compact per-rank buffers (N > 1, all-gathered as device tensors) or straight into the stitched fields
(N = 1), and the global unwrap reads them in place (`gpa_tile_gradients_dev`,
`gpa_unwrap_prediff_dev`).  The `compute=` hook (host arrays) exists for the oracle and the gloo tests.

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


def _axis_tiles(n, g, halo, W):
    """(origin, size, window start) of the tiles along one axis.  g tiles of n / g pixels in windows of
    n / g + 2 halo (W None), or windows of exactly W pixels (e.g. a power of two, which the lock-in
    FFTs run natively instead of zero-padded to >= 2 W) around ceil(n / g) pixel tiles, g derived from
    W when None; the last tile may be shorter."""
    if W is None:
        if n % g:
            raise ValueError('axis length %d is not divisible into %d tiles' % (n, g))
        t = n // g
        if t + 2 * halo > n:
            raise ValueError('tile + 2*halo exceeds the image')
        return [(i * t, t, window_start(i, t, halo, n)) for i in range(g)], t, t + 2 * halo
    if W > n:
        raise ValueError('window %d exceeds the axis length %d' % (W, n))
    if W == n:
        return [(0, n, 0)], n, W
    if W <= 2 * halo:
        raise ValueError('window %d leaves no interior with halo %d' % (W, halo))
    if g is None:
        g = -(-n // (W - 2 * halo))
    t = -(-n // g)
    if t + 2 * halo > W:
        raise ValueError('%d tiles of %d pixels + 2*halo do not fit windows of %d' % (g, t, W))
    out = []
    for i in range(g):
        o, size = i * t, min(t, n - i * t)
        if size <= 0:
            break
        out.append((o, size, int(min(max(o - (W - size) // 2, 0), n - W))))
    return out, t, W


def tile_plan(shape, grid, halo, window=None):
    """Tiles of the image and the windows they are computed in.

    Returns (tiles, (t0, t1), (w0, w1)): tiles is a list of ((i, j), window slices, offset of the tile
    inside its window, tile size); (t0, t1) the largest tile, (w0, w1) the common window shape.
    Tile (i, j) covers image[i*t0 : i*t0 + size0, j*t1 : j*t1 + size1]."""
    g0, g1 = grid if grid is not None else (None, None)
    W0, W1 = window if window is not None else (None, None)
    a0, t0, w0 = _axis_tiles(shape[0], g0, halo, W0)
    a1, t1, w1 = _axis_tiles(shape[1], g1, halo, W1)
    out = []
    for i, (o0, z0, s0) in enumerate(a0):
        for j, (o1, z1, s1) in enumerate(a1):
            out.append(((i, j), (slice(s0, s0 + w0), slice(s1, s1 + w1)), (o0 - s0, o1 - s1), (z0, z1)))
    return out, (t0, t1), (w0, w1)


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


def _tiled_device(image, kvecs, klists, sigma, halo, kmax, dtype, device, group, tiles, tshape, wshape, world, rank,
                  use_torch):
    """Device-resident tile pipeline (module docstring): numpy image in, numpy (2, N, M) out."""
    from . import _lib
    dtype = np.dtype(dtype)
    rsz = dtype.itemsize
    n0, n1 = image.shape
    t0, t1 = tshape
    npx = n0 * n1
    P, K = klists.shape[:2]
    border = 2 * int(sigma)
    plan_w = _lib.get_plan(wshape, P * K, dtype, device)
    img = np.ascontiguousarray(image, dtype=dtype)
    if not use_torch:
        d_img = _lib.DeviceBuffer(img.nbytes, device)
        d_img.upload(img)
        mean = plan_w.mean_dev(d_img.ptr, npx)
        gdx = _lib.DeviceBuffer(2 * n0 * (n1 - 1) * rsz, device)
        gdy = _lib.DeviceBuffer(2 * (n0 - 1) * n1 * rsz, device)
        gw = _lib.DeviceBuffer(npx * rsz, device)
        for (i, j), (w0, w1), (o0, o1), (z0, z1) in tiles:
            gi, gj = i * t0, j * t1
            plan_w.tile_gradients_dev(d_img.ptr, n1, w0.start, w1.start, mean, kvecs, klists, sigma, border,
                                      (o0, o1, z0, z1),
                                      (gdx.ptr + (gi * (n1 - 1) + gj) * rsz, n1 - 1, n0 * (n1 - 1)),
                                      (gdy.ptr + (gi * n1 + gj) * rsz, n1, (n0 - 1) * n1),
                                      (gw.ptr + (gi * n1 + gj) * rsz, n1))
        plan_w.sync()
        d_img.free()
        plan_g = _lib.get_plan(image.shape, 1, dtype, device)
        d_u = _lib.DeviceBuffer(2 * npx * rsz, device)
        for c in range(2):
            plan_g.unwrap_prediff_dev(gdx.ptr + c * n0 * (n1 - 1) * rsz, gdy.ptr + c * (n0 - 1) * n1 * rsz, gw.ptr,
                                      d_u.ptr + c * npx * rsz, kmax=kmax)
        u = d_u.download((2, n0, n1), dtype)
        for b in (gdx, gdy, gw, d_u):
            b.free()
        return u

    pipe = TiledPipeline(image.shape, kvecs, klists, sigma, halo, kmax, dtype, device, group, tiles=tiles,
                         tshape=tshape, wshape=wshape)
    pipe.load(img)
    u = pipe.step().cpu().numpy()
    pipe.close()
    return u


class HipTileBackend:
    """What TiledPipeline asks of the GPU, through the C ABI of libgpa_hip.so: the sum over the tile interiors (one
    launch, result on the device), the tile stage of one window (sweep + per-pixel least squares, interiors written into
    the rank's tile buffer by one copy launch), the stitching of gathered tiles (one launch per component), the global
    weighted unwrap of one component (on the global plan's own stream, waited for later), and the ORDER between the
    plans' streams and torch's current stream (events: no host synchronisation on the nccl / single-rank path).  A test
    double with the same methods (tests/test_distributed.py: CPU tensors + the oracle) lets the schedule, the collectives
    and the stitching run under gloo without a GPU."""

    def __init__(self, torch, wshape, shape, nbatch, dtype, device):
        from . import _lib
        self.torch = torch
        self.device = torch.device('cuda', int(device))
        self.rsz = np.dtype(dtype).itemsize
        self.wshape = wshape
        self.plan_w = _lib.get_plan(wshape, nbatch, dtype, device)
        self._mk = lambda: _lib.Plan(shape, 1, dtype, device)
        self._get = lambda: _lib.get_plan(shape, 1, dtype, device)
        self.plan_shared = None         # the cached global plan: every solve that does not overlap another one here
        self.plan_second = None         # a second global plan (stream + workspace), created only when a rank solves both
        self.plan_c = [None, None]      # components of one image at once; plan_c[c] = the plan of c's solve in flight

    def _plan(self, c, concurrent=False):
        """global plan of component c.  concurrent: this rank solves BOTH components of one image at once (run_stream on
        one rank) and component 1 gets a plan -- stream and workspace -- of its own; otherwise the two share the cached
        global plan (step() solves them one after the other and gains nothing from a second one: ADVICE r03)"""
        if c == 1 and concurrent:
            if self.plan_second is None:
                self.plan_second = self._mk()
            self.plan_c[1] = self.plan_second
        else:
            if self.plan_shared is None:
                self.plan_shared = self._get()
            self.plan_c[c] = self.plan_shared
        return self.plan_c[c]

    def _torch_stream(self):
        return self.torch.cuda.current_stream(self.device).cuda_stream

    def tile_sums(self, wins, rects, ntiles, max_rows, out):
        self.plan_w.tile_sums_dev(wins.data_ptr(), wins.stride(0), wins.stride(1), rects.data_ptr(), ntiles, max_rows,
                                  out.data_ptr())

    def set_mean(self, sum_t, scale):
        self.plan_w.tile_set_mean_dev(sum_t.data_ptr(), scale)

    def tile_gradients(self, win, wpitch, kvecs, klists, sigma, border, rect, local, slot):
        """local: (2, per_rank, 3, t0, t1); component c's block of this tile = (dudx_c, dudy_c, weight)"""
        t0, t1 = local.shape[3], local.shape[4]
        plane, comp = t0 * t1, local.stride(0)
        base = local[0, slot].data_ptr()
        self.plan_w.tile_gradients_meandev_dev(win.data_ptr(), wpitch, 0, 0, kvecs, klists, sigma, border, rect,
                                               (base, t1, comp), (base + plane * self.rsz, t1, comp),
                                               (base + 2 * plane * self.rsz, t1, comp))

    def tiles_to_torch(self, host):
        """what the tile plan has enqueued becomes visible to torch: its current stream waits (collectives on device
        tensors are ordered on that stream), or -- host: data about to be staged through the host -- the plan is drained"""
        if host:
            self.plan_w.sync()
        else:
            self.plan_w.stream_wait(self._torch_stream())

    def torch_to_tiles(self):
        """the tile plan's stream waits for torch's current stream (the all-reduced sum)"""
        self.plan_w.wait_stream(self._torch_stream())

    def stitch(self, c, tiles, slot_stride, table, ntiles, t0, t1, gdx, gdy, gw, concurrent=False):
        """tiles[slot] = (dudx_c, dudy_c, w) -> the full-size fields of component c, on that component's plan, whose
        stream first waits for torch's (the gather that filled `tiles`)"""
        pl = self._plan(c, concurrent)
        pl.wait_stream(self._torch_stream())
        n0, n1 = gw.shape
        pl.stitch_tiles_dev(tiles.data_ptr(), slot_stride, t0 * t1, t1, table.data_ptr(), ntiles, t0, t1,
                            [(gdx.data_ptr(), n1 - 1, n0, n1 - 1), (gdy.data_ptr(), n1, n0 - 1, n1), (gw.data_ptr(), n1, n0, n1)])

    def stitch_to_tiles(self, c):
        """the tile plan's stream waits for component c's stitch (one rank: the stitch reads the tile buffer in place, and
        the NEXT image's tile stage overwrites it -- ADVICE r04).  Called between the stitch and the solve's enqueue, so the
        tile stage waits for the stitch only, not for the solve behind it on the same stream."""
        self.plan_c[c].stream_wait(self.plan_w.stream())

    def unwrap_start(self, c, gdx, gdy, gw, out, kmax, concurrent=False):
        self._plan(c, concurrent).unwrap_prediff_enqueue_dev(gdx.data_ptr(), gdy.data_ptr(), gw.data_ptr(), out.data_ptr(), kmax=kmax)

    def unwrap_wait(self, c):
        return self.plan_c[c].unwrap_finish()

    def unwrap_to_torch(self, c):
        """torch's current stream waits for component c's plan (its field is about to be sent / handed out)"""
        self.plan_c[c].stream_wait(self._torch_stream())

    def undistort_tiles(self, image, u, out, uinv, rects, scale=1.0):
        """Lawler-Fujita on the global plan (gpa_undistort_image_dev): u_inv = invert_u_overlap(-u) and the deformed image
        resampled at r + u_inv, for the output windows `rects` only (None: everywhere); ordered after torch's stream (the
        broadcast that delivered u), and torch's stream after it"""
        pl = self._plan(0)
        pl.wait_stream(self._torch_stream())
        pl.undistort_image_dev(image.data_ptr(), u.data_ptr(), out.data_ptr(), uinv_ptr=uinv.data_ptr(), rects=rects, scale=scale)
        pl.stream_wait(self._torch_stream())

    def sync_device(self):
        self.torch.cuda.synchronize(self.device)

    def close(self):
        if self.plan_second is not None:
            self.plan_second.close()
        self.plan_second = self.plan_shared = None
        self.plan_c = [None, None]


class TiledPipeline:
    """Device-resident tile pipeline of one rank (module docstring), reusable over many images of one shape.

    `load(image)` uploads this rank's windows (1/N of the image plus halos); `step()` runs
      sum over the tile interiors (one launch) -> all_reduce (the mean of the WHOLE image that every window is offset
      by, geometric_phase_analysis.py:919; one scalar that never leaves the device) -> tile stage (sweep + least squares
      per window, C ABI) -> all_gather #1 of the gradient tiles -> stitch (one launch per component) -> global unwrap of
      component c on rank c % N -> broadcast of each component
    and returns the stitched (2, N, M) field as a device tensor that every rank holds.  Buffers are torch
    tensors (device memory + collectives are what torch is here for); with the "nccl" backend the
    collectives run on them over RCCL/xGMI, with "gloo" (CPU tests, two ranks sharing a GPU) they are
    staged through the host.  The tile buffer keeps, per component c, the block (dudx_c, dudy_c, weight) of every tile
    contiguous: what the owner of component c is sent is one slice, and the stitch reads it as it arrives."""

    def __init__(self, shape, kvecs, klists, sigma, halo, kmax=10, dtype=np.float32, device=0, group=None,
                 grid=None, window=None, tiles=None, tshape=None, wshape=None, backend=None, dist_sync=None):
        torch, dist = _dist()
        self.torch, self.dist, self.group = torch, dist, group
        import os
        self.dist_sync = int(os.environ.get('GPA_DIST_SYNC', '0') or 0) if dist_sync is None else int(dist_sync)
        # GPA_DIST_FORCE_COLLECTIVES=1 (test aid): a process group of ONE rank takes every collective of the N > 1 path anyway
        # (all_reduce, all_gather_into_tensor, gather, broadcast with itself) instead of the single-rank shortcuts -- on a
        # one-GPU box the RCCL calls and the stream-event ordering around them then really execute
        self.force_collectives = _initialized() and os.environ.get('GPA_DIST_FORCE_COLLECTIVES', '0') not in ('', '0')
        self.world, self.rank = 1, 0
        self.backend = None
        if _initialized():
            self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
            self.backend = dist.get_backend(group)
        self.multi = self.world > 1 or self.force_collectives      # the collectives run (False: one rank reads its own buffers)
        self.shape = (int(shape[0]), int(shape[1]))
        self.kvecs = np.asarray(kvecs, dtype=np.float64).reshape(-1, 2)
        self.klists = np.asarray(klists, dtype=np.float64)
        self.sigma, self.kmax = sigma, int(kmax)
        self.border = 2 * int(sigma)
        self.dtype = np.dtype(dtype)
        self.device = int(device)
        if tiles is None:
            tiles, tshape, wshape = tile_plan(self.shape, grid, halo, window)
        self.tiles, self.tshape, self.wshape = tiles, tuple(tshape), tuple(wshape)
        self.mine = list(range(self.rank, len(tiles), self.world))
        self.per_rank = (len(tiles) + self.world - 1) // self.world
        P, K = self.klists.shape[:2]
        self.unwrappers = [c for c in range(2) if c % self.world == self.rank]
        self.be = backend if backend is not None else HipTileBackend(torch, self.wshape, self.shape, P * K, self.dtype,
                                                                     self.device)
        dev = self.be.device
        self.dev = dev
        t_dt = torch.float32 if self.dtype == np.float32 else torch.float64
        n0, n1 = self.shape
        t0, t1 = self.tshape
        pr = self.per_rank
        self.wins = torch.zeros((max(len(self.mine), 1),) + self.wshape, dtype=t_dt, device=dev)
        self.local = torch.zeros((2, pr, 3, t0, t1), dtype=t_dt, device=dev)
        self.gathered = torch.zeros((self.world, 2, pr, 3, t0, t1), dtype=t_dt, device=dev) if self.multi else None
        self.gdx = torch.zeros((2, n0, n1 - 1), dtype=t_dt, device=dev)
        self.gdy = torch.zeros((2, n0 - 1, n1), dtype=t_dt, device=dev)
        self.gw = torch.zeros((n0, n1), dtype=t_dt, device=dev)
        self.u = torch.zeros((2, n0, n1), dtype=t_dt, device=dev)
        self.sums = torch.zeros(1, dtype=torch.float64, device=dev)
        # device tables: interior rectangles of this rank's windows; (slot, r0, c0, z0, z1) of every tile for the stitch of
        # a buffer gathered as (world, 2, per_rank, ...) [step] and as (world, per_rank, ...) [run_stream]
        rects = [[self.tiles[idx][2][0], self.tiles[idx][2][1], self.tiles[idx][3][0], self.tiles[idx][3][1]] for idx in self.mine]
        self.rects = torch.tensor(rects if rects else [[0, 0, 1, 1]], dtype=torch.int32, device=dev)
        self.max_rows = max([r[2] for r in rects] + [1])
        tab = lambda slot_of: torch.tensor([[slot_of(idx), ij[0] * t0, ij[1] * t1, z[0], z[1]]
                                            for idx, (ij, _, _, z) in enumerate(self.tiles)], dtype=torch.int32, device=dev)
        self.table_step = tab(lambda idx: (idx % self.world) * 2 * pr + idx // self.world)
        self.table_stream = tab(lambda idx: (idx % self.world) * pr + idx // self.world)
        self.mean = None      # (kept for callers that read it: the scalar itself stays on the device)
        self.iters = [0, 0]

    # ---- collectives (device tensors on RCCL, host-staged on gloo) -------------------------------
    def _host_staged(self):
        return self.backend is not None and self.backend != 'nccl'

    def _fence(self, what, after=False):
        """GPA_DIST_SYNC=1 (or TiledPipeline(..., dist_sync=True)): drain this rank's device and meet the other ranks at a
        barrier before AND after every collective, so that the first run on a real multi-GPU node can be bisected -- a hang
        or a wrong field then points at ONE collective instead of at whatever the streams had overlapped with it.
        GPA_DIST_SYNC=2 also logs every collective with its rank and wall time to stderr.  Off by default: the schedule
        overlaps collectives with kernels through stream events and synchronises the host once per image."""
        if not self.dist_sync or not self.multi:
            return
        self.be.sync_device()
        self.dist.barrier(group=self.group)
        if self.dist_sync > 1:
            import sys
            import time
            print('[gpa dist_sync] rank %d %s %s %.6f' % (self.rank, 'done' if after else 'enter', what, time.time()),
                  file=sys.stderr, flush=True)

    def _all_reduce_sum(self, t):
        if not self.multi:
            return t
        self._fence('all_reduce %s' % (tuple(t.shape),))
        if self._host_staged():
            h = t.cpu()
            self.dist.all_reduce(h, group=self.group)
            t.copy_(h)
        else:
            self.dist.all_reduce(t, group=self.group)
        self._fence('all_reduce', after=True)
        return t

    def _all_gather(self, out, t):
        self._fence('all_gather %s' % (tuple(t.shape),))
        if self._host_staged():
            h = out.cpu()
            self.dist.all_gather_into_tensor(h, t.cpu().contiguous().unsqueeze(0), group=self.group)   # gloo wants world x input
            out.copy_(h)
        else:
            self.dist.all_gather_into_tensor(out, t, group=self.group)
        self._fence('all_gather', after=True)

    def _broadcast(self, t, src):
        if not self.multi:
            return
        src_global = src if self.group is None else self.dist.get_global_rank(self.group, src)
        self._fence('broadcast from %d %s' % (src, tuple(t.shape)))
        if self._host_staged():
            h = t.cpu()
            self.dist.broadcast(h, src_global, group=self.group)
            if self.rank != src:
                t.copy_(h)
        else:
            self.dist.broadcast(t, src_global, group=self.group)
        self._fence('broadcast', after=True)

    # ---- data ------------------------------------------------------------------------------------
    def load(self, image=None, window_fn=None):
        """upload this rank's windows: from the full `image` (every rank passes the same array) or from
        `window_fn(slice0, slice1)` that produces the pixels of one window (so that no rank needs the
        whole image in host memory)."""
        torch = self.torch
        for slot, idx in enumerate(self.mine):
            _, (w0, w1), _, _ = self.tiles[idx]
            win = window_fn(w0, w1) if window_fn is not None else image[w0, w1]
            self.wins[slot].copy_(torch.from_numpy(np.ascontiguousarray(win, dtype=self.dtype)))
        self.be.sync_device()

    def image_mean(self):
        """mean of the whole image, left ON THE DEVICE as the tile plan's mean: every pixel lies in exactly one tile
        interior, so each rank sums its interiors in one launch (deterministic order), the ranks all-reduce the one
        double, and the tile stage reads sum / pixels from device memory -- no host synchronisation"""
        if self.mine:
            self.be.tile_sums(self.wins, self.rects, len(self.mine), self.max_rows, self.sums)
        else:
            self.sums.zero_()
        if self.multi:
            self.be.tiles_to_torch(self._host_staged())
            self._all_reduce_sum(self.sums)
            self.be.torch_to_tiles()
        self.be.set_mean(self.sums, 1.0 / (self.shape[0] * self.shape[1]))

    def _tile_stage(self):
        for slot, idx in enumerate(self.mine):
            _, _, (o0, o1), (z0, z1) = self.tiles[idx]
            self.be.tile_gradients(self.wins[slot], self.wshape[1], self.kvecs, self.klists, self.sigma, self.border,
                                   (o0, o1, z0, z1), self.local, slot)

    def step(self, timings=None):
        """one image through the unpipelined schedule.  timings: a dict that receives the DEVICE-inclusive milliseconds of
        every stage on this rank (the device is drained after each stage: a diagnostic pass, not a timed one)"""
        import time
        n0, n1 = self.shape
        t0, t1 = self.tshape
        plane = t0 * t1
        tick = [time.perf_counter()]

        def mark(name):
            if timings is not None:
                self.be.sync_device()
                now = time.perf_counter()
                timings[name] = timings.get(name, 0.0) + (now - tick[0]) * 1e3
                tick[0] = now

        if timings is not None:
            self.be.sync_device()
            tick[0] = time.perf_counter()
        self.image_mean()
        mark('mean')
        self._tile_stage()
        mark('tile_stage')
        # --- collective 1 (RCCL all_gather over xGMI): the tile blocks of every rank (one rank: read in place)
        self.be.tiles_to_torch(self._host_staged())
        if self.multi:
            self._all_gather(self.gathered, self.local)
            src = self.gathered
        else:
            src = self.local
        mark('all_gather')
        # --- stitch + global unwrap, component c on rank c % world; collective 2 hands each component to everybody
        for c in self.unwrappers:
            comp = src[0, c] if self.multi else src[c]
            self.be.stitch(c, comp, 3 * plane, self.table_step if self.multi else self.table_stream, len(self.tiles), t0, t1,
                           self.gdx[c], self.gdy[c], self.gw)
            self.be.unwrap_start(c, self.gdx[c], self.gdy[c], self.gw, self.u[c], self.kmax)
            self.iters[c] = self.be.unwrap_wait(c)
        mark('stitch+unwrap')
        for c in range(2):
            self._broadcast(self.u[c], c % self.world)
        mark('broadcast')
        return self.u

    # ---- Lawler-Fujita undistortion of the stitched field, sharded over the tiles ------------------------------------
    def undistort(self, image, u=None, scale=1.0):
        """undistort_image (geometric_phase_analysis.py:935-974) with the field this pipeline produced: every rank holds u
        (step() broadcasts it; pass another (2, N, M) device tensor otherwise) and is given the whole deformed `image`
        (host array or device tensor -- every rank the same, as for load()).  The fixed-point inversion (35 + 1 rounds of two
        cubic-spline interpolations per pixel) and the final resampling are independent per output pixel: rank r computes
        them for the interiors of the tiles it owns (gpa_undistort_image_dev with those windows; the spline prefilter of the
        whole field, ~3 % of the work, runs on every rank), and one all_reduce of the zero-filled outputs hands every rank
        the whole undistorted image and u_inv.  Returns (undistorted (N, M), u_inv (2, N, M)) as device tensors.  No host
        round trip of u.
        The two tensors are buffers of the pipeline: the NEXT undistort() call overwrites them (clone what must outlive it),
        as run_stream's results are.  The sum also turns a -0.0 of the owner into +0.0.  (the all_reduce moves 3
        full planes per rank where a gather of the owned tiles would move 1 / N of that; it runs outside every timed region.)
        `scale`: undistort_image(image, scale * u) -- scale = -1 for the field exactly as this pipeline extracted it, which is
        MINUS the displacement (tests/test_geometric_phase_analysis.py:63, :76)."""
        torch = self.torch
        u = self.u if u is None else u
        t_dt = torch.float32 if self.dtype == np.float32 else torch.float64
        if not torch.is_tensor(image):
            image = torch.from_numpy(np.ascontiguousarray(image, dtype=self.dtype))
        image = image.to(device=self.dev, dtype=t_dt).contiguous()
        if getattr(self, '_lf', None) is None:
            self._lf = (torch.zeros(self.shape, dtype=t_dt, device=self.dev), torch.zeros((2,) + self.shape, dtype=t_dt, device=self.dev))
        out, uinv = self._lf
        if not self.multi:
            self.be.undistort_tiles(image, u, out, uinv, None, scale)
            return out, uinv
        t0, t1 = self.tshape
        rects = [(self.tiles[idx][0][0] * t0, self.tiles[idx][0][1] * t1, self.tiles[idx][3][0], self.tiles[idx][3][1]) for idx in self.mine]
        out.zero_()
        uinv.zero_()
        if rects:
            self.be.undistort_tiles(image, u, out, uinv, rects, scale)
        for t in (out, uinv):
            if self._host_staged():
                self.be.sync_device()
            self._all_reduce_sum(t)      # every pixel has ONE non-zero contributor: the sum is exact
        return out, uinv

    # ---- image-pipelined schedule ---------------------------------------------------------------------------
    # step() leaves N - 2 of N GPUs idle during the global unwrap (54 % of single-GPU time).  For a STREAM of images
    # the unwrap of image i therefore runs on a rotating pair of ranks -- component c of image i on rank
    # (2 i + c) % N, on the global plan's own stream -- while ALL ranks sweep the windows of image i + 1; only the
    # two owners receive the gradient tiles they need (a gather each, 3 of the 5 fields), and the second owner hands
    # its component to the first, which then holds the whole field of image i.  No collective involves every rank's
    # full-size data any more: per image and rank 1/N of the image goes out once.
    def owners(self, i):
        """ranks that unwrap component 0 / component 1 of image i; the first one ends up with the whole field"""
        return (2 * i) % self.world, (2 * i + 1) % self.world

    def _gather_to(self, c, dst):
        """component c's tile blocks of every rank on rank dst, in its preallocated receive buffer (world, per_rank, 3,
        t0, t1) -- the ranks' slices are received in place, nothing is stacked or copied afterwards; None elsewhere.
        One rank: the tile buffer itself."""
        dist = self.dist
        part = self.local[c]                      # contiguous: one slice per owner
        if not self.multi:
            return part
        dst_global = dst if self.group is None else dist.get_global_rank(self.group, dst)
        rx = self._rx[c] if self.rank == dst else None
        self._fence('gather of component %d to %d' % (c, dst))
        if self._host_staged():
            hp = part.cpu()
            outs = [self.torch.empty_like(hp) for _ in range(self.world)] if rx is not None else None
            dist.gather(hp, outs, dst=dst_global, group=self.group)
            if rx is not None:
                for r, o in enumerate(outs):
                    rx[r].copy_(o)
            self._fence('gather', after=True)
            return rx
        dist.gather(part, list(rx.unbind(0)) if rx is not None else None, dst=dst_global, group=self.group)
        self._fence('gather', after=True)
        return rx

    def _send_component(self, t, src, dst):
        """component 1 of an image from its owner to the owner of component 0 (point to point)"""
        if src == dst:
            return
        dist = self.dist
        g = (lambda r: r) if self.group is None else (lambda r: dist.get_global_rank(self.group, r))
        self._fence('send %d -> %d' % (src, dst))
        if self._host_staged():
            if self.rank == src:
                dist.send(t.cpu(), g(dst), group=self.group)
            elif self.rank == dst:
                h = t.cpu()
                dist.recv(h, g(src), group=self.group)
                t.copy_(h)
            self._fence('send', after=True)
            return
        # RCCL: a grouped isend / irecv runs on the communicator of the whole group (a bare send / recv would build a
        # two-rank communicator per pair on first use, inside somebody's timed region)
        ops = []
        if self.rank == src:
            ops.append(dist.P2POp(dist.isend, t, g(dst), group=self.group))
        elif self.rank == dst:
            ops.append(dist.P2POp(dist.irecv, t, g(src), group=self.group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        self._fence('send', after=True)

    def run_stream(self, sources, on_result=None):
        """A stream of images of the pipeline's shape through the image-pipelined schedule.

        sources: iterable of images -- full arrays (every rank passes the same), callables window_fn(slice0,
        slice1), or None for "the windows that are loaded already".  For image i the field ends up on rank owners(i)[0]; `on_result(i, u)` is called there with the
        (2, N, M) device tensor (valid until the image after next finishes).  Returns on every rank the list of
        iteration counts [(it0, it1) or None per image] and leaves per-stage wall times of this rank in
        `self.stage_s` (load, mean, tile stage, gather + stitch, waiting for unwraps, hand-over: host time spent
        ENQUEUEING on the nccl / single-rank path, where nothing between the load and the wait for the unwraps
        synchronises with the device).
        Per image the arithmetic is that of step(): the same windows, the same tile kernels, the same global solves --
        the fields are equal bit for bit (tests/test_distributed.py, tests/test_gpu_configs.py)."""
        import time
        torch = self.torch
        n0, n1 = self.shape
        t0, t1 = self.tshape
        plane = t0 * t1
        t_dt = torch.float32 if self.dtype == np.float32 else torch.float64
        if getattr(self, '_pbuf', None) is None:
            # two sets of stitched fields (the unwrap of image i reads one while image i + 1 is stitched into the other);
            # the weight is kept per component because the two owners of an image are different ranks
            self._pbuf = [(torch.zeros((2, n0, n1 - 1), dtype=t_dt, device=self.dev),
                           torch.zeros((2, n0 - 1, n1), dtype=t_dt, device=self.dev),
                           torch.zeros((2, n0, n1), dtype=t_dt, device=self.dev)) for _ in range(2)]
            self._pu = [torch.zeros((2, n0, n1), dtype=t_dt, device=self.dev) for _ in range(2)]
            # receive buffers of the gathers (allocated once; a component's buffer is free again when its unwrap,
            # which follows the stitch on the same stream, has been waited for -- before the next image's gather)
            self._rx = [torch.zeros((self.world, self.per_rank, 3, t0, t1), dtype=t_dt, device=self.dev)
                        for _ in range(2)] if self.multi else None
        st = {k: 0.0 for k in ('load', 'mean', 'tiles', 'gather', 'unwrap_wait', 'handover')}
        iters_all = []
        pending = None      # (index, buffer set) of the image whose unwraps are in flight
        both = self.world == 1     # this rank solves both components of an image at once: two plans

        def finish(pi, pb):
            a, b = self.owners(pi)
            t = time.perf_counter()
            its = [None, None]
            for c, owner in ((0, a), (1, b)):
                if self.rank == owner:
                    its[c] = self.be.unwrap_wait(c)
            st['unwrap_wait'] += time.perf_counter() - t
            t = time.perf_counter()
            self._send_component(self._pu[pb][1], b, a)
            st['handover'] += time.perf_counter() - t
            if self.rank == a and on_result is not None:
                self.be.sync_device()
                on_result(pi, self._pu[pb])
            iters_all.append(tuple(its))

        for i, src in enumerate(sources):
            buf = i & 1
            t = time.perf_counter()
            if src is None:
                pass                      # the windows loaded last stay (benchmarks: input resident in HBM)
            elif callable(src):
                self.load(window_fn=src)
            else:
                self.load(src)
            st['load'] += time.perf_counter() - t
            t = time.perf_counter()
            self.image_mean()
            st['mean'] += time.perf_counter() - t
            t = time.perf_counter()
            self._tile_stage()
            # the previous image's unwraps have been running beside these launches; their owners wait for them now
            if pending is not None:
                st['tiles'] += time.perf_counter() - t
                finish(*pending)
                pending = None
                t = time.perf_counter()
            self.be.tiles_to_torch(self._host_staged())
            st['tiles'] += time.perf_counter() - t
            t = time.perf_counter()
            a, b = self.owners(i)
            for c, owner in ((0, a), (1, b)):
                got = self._gather_to(c, owner)
                if self.rank == owner:
                    gdx, gdy, gw = self._pbuf[buf]
                    self.be.stitch(c, got, 3 * plane, self.table_stream, len(self.tiles), t0, t1, gdx[c], gdy[c], gw[c],
                                   concurrent=both)
                    if not self.multi:
                        self.be.stitch_to_tiles(c)      # `got` IS the tile buffer: the next tile stage must not overwrite it yet
                    self.be.unwrap_start(c, gdx[c], gdy[c], gw[c], self._pu[buf][c], self.kmax, concurrent=both)
            st['gather'] += time.perf_counter() - t
            pending = (i, buf)
        if pending is not None:
            finish(*pending)
        self.stage_s = st
        return iters_all

    def close(self):
        if getattr(self, 'be', None) is not None:
            self.be.close()
        for name in ('wins', 'local', 'gathered', 'gdx', 'gdy', 'gw', 'u', '_pbuf', '_pu', '_rx', '_lf', 'rects', 'table_step',
                     'table_stream'):
            setattr(self, name, None)


def extract_displacement_field_tiled(image, kvecs, grid=None, sigma=None, kwscale=2.5, ksteps=3, klists=None,
                                     halo=None, kmax=10, dtype=np.float64, device=0, group=None, compute=None,
                                     window=None, _force_torch=False):
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
    tiles, (t0, t1), wshape = tile_plan(image.shape, grid, halo, window)
    if compute is None:
        return _tiled_device(image, kvecs, klists, sigma, halo, kmax, dtype, device, group, tiles, (t0, t1), wshape,
                             world, rank, world > 1 or _force_torch)
    gradients, unwrap = compute

    # --- local stage: my tiles (round robin), interiors of dudx (2), dudy (2), wnorm (1)
    per_rank = (len(tiles) + world - 1) // world
    local = np.zeros((per_rank, 5, t0, t1), dtype=dtype)
    mean = image.mean()      # the driver subtracts the IMAGE mean (geometric_phase_analysis.py:919)
    for slot, idx in enumerate(range(rank, len(tiles), world)):
        _, (w0, w1), (o0, o1), (z0, z1) = tiles[idx]
        dudx, dudy, wn = gradients(np.ascontiguousarray(image[w0, w1] - mean), kvecs, klists, sigma, 2 * int(sigma))
        # pad the difference fields to the window shape so interiors can be cut uniformly
        dx = np.zeros((2,) + wshape, dtype=dtype)
        dy = np.zeros((2,) + wshape, dtype=dtype)
        dx[:, :, :-1] = dudx
        dy[:, :-1, :] = dudy
        local[slot, 0:2, :z0, :z1] = dx[:, o0:o0 + z0, o1:o1 + z1]
        local[slot, 2:4, :z0, :z1] = dy[:, o0:o0 + z0, o1:o1 + z1]
        local[slot, 4, :z0, :z1] = wn[o0:o0 + z0, o1:o1 + z1]

    # --- collective 1: stitch the gradient tiles
    gathered = _all_gather_np(local, group)
    n0, n1 = image.shape
    full = np.zeros((5, n0, n1), dtype=dtype)
    for idx, ((i, j), _, _, (z0, z1)) in enumerate(tiles):
        full[:, i * t0:i * t0 + z0, j * t1:j * t1 + z1] = gathered[idx % world][idx // world][:, :z0, :z1]

    # --- global unwrap, one component per rank; collective 2 distributes them
    mine = np.zeros((2, n0, n1), dtype=dtype)
    for c in range(2):
        if c % world == rank:
            mine[c] = unwrap(full[c, :, :-1], full[2 + c, :-1, :], full[4], kmax)
    parts = _all_gather_np(mine, group)
    return np.stack([parts[c % world][c] for c in range(2)])


# ---------------------------------------------------------------------------------------------------
# (peak x k-vector) sharding of ONE image: exact strong scaling (option 1)
# ---------------------------------------------------------------------------------------------------
def default_ksharded_compute(shape, nbatch, npeaks, dtype, device):
    """sweep / reconstruct / unwrap callables backed by libgpa_hip.so (host arrays in and out)"""
    from . import _lib
    plans = {}

    def plan():
        if 'p' not in plans:
            plans['p'] = _lib.Plan(shape, max(nbatch, npeaks), dtype, device)
        return plans['p']

    def sweep(img0, sigma, klist, kref):
        lockin, kidx, _ = plan().sweep(img0, kref, klist, sigma)
        return lockin, kidx

    def reconstruct(lockins, kvecs, border):
        return plan().reconstruct_grad(lockins, kvecs, border)

    def unwrap(dx, dy, weight, kmax):
        return plan().unwrap_prediff(dx, dy, weight, kmax=kmax)[0]
    return sweep, reconstruct, unwrap


def extract_displacement_field_ksharded(image, kvecs, sigma=None, kwscale=2.5, ksteps=3, klists=None, kmax=10,
                                        dtype=np.float64, device=0, group=None, compute=None, _simulate_world=None):
    """`extract_displacement_field` of one image with the (peak x candidate) lock-ins dealt over the ranks.

    Every lock-in is independent, so rank r runs candidates r, r + N, r + 2N, ... of every peak on the whole
    (replicated) image and keeps its partial winner per pixel; one all_gather of the partial winners
    (complex lock-in + global candidate index, P x N x M each) and a local selection -- larger amplitude,
    then smaller candidate index, which is what the sequential strict '>' of the reference keeps
    (geometric_phase_analysis.py:683) -- give every rank the lock-ins of the full sweep.  The two unwraps then
    run on ranks 0 and 1 and a second all_gather distributes u.  Same numbers as one GPU (each candidate
    is computed by the same kernel), except that amplitudes tying to rounding may resolve differently."""
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
    klists = [np.asarray(k, dtype=np.float64).reshape(-1, 2) for k in klists]
    P = len(kvecs)
    K = max(len(k) for k in klists)
    if compute is None:
        compute = default_ksharded_compute(image.shape, -(-K // world), P, dtype, device)
    sweep, reconstruct, unwrap = compute
    cdtype = np.complex64 if np.dtype(dtype) == np.float32 else np.complex128
    img0 = image - image.mean()

    # --- local stage: my candidates of every peak
    def partial_winners(r, w):
        lock = np.zeros((P,) + image.shape, dtype=cdtype)
        gidx = np.full((P,) + image.shape, -1, dtype=np.int32)
        for p in range(P):
            mine = np.arange(r, len(klists[p]), w)
            if len(mine) == 0:
                continue
            lk, kidx = sweep(img0, sigma, klists[p][mine], kvecs[p])
            lock[p] = lk
            gidx[p] = np.where(kidx >= 0, mine[np.maximum(kidx, 0)], -1)
        return lock, gidx

    if _simulate_world:   # test aid: the ranks' shares one after the other in this process, no collective
        pw = [partial_winners(r, int(_simulate_world)) for r in range(int(_simulate_world))]
        locks, gidxs = [x[0] for x in pw], [x[1] for x in pw]
    else:
        lock, gidx = partial_winners(rank, world)
        # --- collective 1: partial winners of every rank
        locks = _all_gather_np(lock, group)
        gidxs = _all_gather_np(gidx, group)
    # --- selection
    best, bidx = locks[0].copy(), gidxs[0].copy()
    for lk, gi in zip(locks[1:], gidxs[1:]):
        a_new, a_old = np.abs(lk), np.abs(best)
        take = (gi >= 0) & ((bidx < 0) | (a_new > a_old) | ((a_new == a_old) & (gi < bidx)))
        best = np.where(take, lk, best)
        bidx = np.where(take, gi, bidx)

    # --- reconstruct (replicated, cheap) and the two unwraps on ranks 0 and 1; collective 2
    dudx, dudy, wnorm = reconstruct(best, kvecs, 2 * int(sigma))
    mine_u = np.zeros((2,) + image.shape, dtype=np.dtype(dtype))
    for c in range(2):
        if c % world == rank:
            mine_u[c] = unwrap(dudx[c], dudy[c], wnorm, kmax)
    parts = _all_gather_np(mine_u, group)
    u = np.stack([parts[c % world][c] for c in range(2)])
    return u, best, bidx


# ---- stacks of frames: the units shard, nothing is exchanged on the data path ----------------------------------------
def stack_shares(nframes, world):
    """contiguous blocks of frames per rank, sizes differing by at most one: [(start, stop)] * world"""
    base, extra = divmod(int(nframes), int(world))
    out, a = [], 0
    for r in range(world):
        b = a + base + (1 if r < extra else 0)
        out.append((a, b))
        a = b
    return out


def extract_displacement_field_stack_sharded(frames, kvecs, sigma=None, kwscale=2.5, ksteps=3, klists=None, kmax=10,
                                             dtype=np.float32, device=0, group=None, gather=True, compute=None):
    """A stack of frames (B, N, M) over the ranks of the process group: rank r runs
    `Plan.extract_displacement_field_stack` (one device call per chunk of frames, pygpa_amd/_lib.py) on its
    contiguous block -- the frames are independent, so there is NO data-path collective and the rate scales with
    the number of GPUs (weak scaling).  gather=True: every rank receives u of all frames (one all_gather of the
    results, padded to equal block sizes); gather=False: (u of the own block, (start, stop)).
    `compute(frames_block, kvecs, klists, sigma, border, kmax) -> u_block` replaces the device call (CPU tests)."""
    world, rank = 1, 0
    if _initialized():
        _, dist = _dist()
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    frames = np.asarray(frames)
    if frames.ndim != 3:
        raise ValueError('frames must be a stack (B, N, M)')
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
    shares = stack_shares(frames.shape[0], world)
    a, b = shares[rank]
    if compute is None:
        from . import _lib
        plan = _lib.get_plan(frames.shape[1:], len(kvecs) * K, dtype, device)

        def compute(block, kvecs, klists, sigma, border, kmax):
            return plan.extract_displacement_field_stack(block, kvecs, klists, sigma, border, kmax=kmax)[0]
    mine = compute(frames[a:b], kvecs, klists, sigma, int(2 * sigma), kmax) if b > a else \
        np.empty((0, 2) + frames.shape[1:], dtype=dtype)
    if not gather:
        return mine, (a, b)
    if world == 1:
        return mine
    width = max(s[1] - s[0] for s in shares)
    padded = np.zeros((width, 2) + frames.shape[1:], dtype=mine.dtype)
    padded[:b - a] = mine
    parts = _all_gather_np(padded, group)
    return np.concatenate([parts[r][:shares[r][1] - shares[r][0]] for r in range(world)], axis=0)
