"""One rank of the ranks-on-one-GPU tests (tests/test_gpu_configs.py): torch.distributed over gloo with the
ranks sharing cuda:0 (device-resident TiledPipeline, host-staged collectives), or over nccl (= RCCL) with a
single rank (the device-tensor collectives of the N > 1 path; RCCL does not take two ranks on one device).
usage: RANK= WORLD_SIZE= MASTER_ADDR= MASTER_PORT= python _tiled_rank_worker.py OUT_PREFIX DTYPE [BACKEND]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    out_prefix, dtype = sys.argv[1], np.dtype(sys.argv[2])
    backend = sys.argv[3] if len(sys.argv) > 3 else 'gloo'
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    if backend == 'nccl':
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', 0))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    from pygpa_amd import distributed as D
    from test_distributed import _case
    img, kvecs, klists = _case()
    # the public entry point (N > 1 -> TiledPipeline) ...
    u = D.extract_displacement_field_tiled(img, kvecs, klists=klists, halo=20, window=(64, 128), dtype=dtype,
                                           _force_torch=True)
    # ... and the reusable object, two steps on the same buffers
    pipe = D.TiledPipeline(img.shape, kvecs, np.stack(klists), 6, 20, kmax=10, dtype=dtype, device=0, grid=(2, 2))
    pipe.load(img)
    pipe.step()
    u2 = pipe.step().cpu().numpy()
    # Lawler-Fujita with the field just extracted, sharded over the tiles (every rank its own windows, one all_reduce)
    lf, uinv = pipe.undistort(img)
    np.savez(out_prefix + '_rank%d.npz' % rank, u=u, u2=u2, tiles=len(pipe.mine), iters=np.array(pipe.iters),
             lf=lf.cpu().numpy(), uinv=uinv.cpu().numpy())
    pipe.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
