"""One rank of the image-pipelined tile schedule on a shared GPU (tests/test_gpu_configs.py): torch.distributed over
gloo, every rank on cuda:0, the device work through libgpa_hip.so.  Each rank runs step() image by image, then the same
images through run_stream, and checks on the owner of every image that the two fields are equal bit for bit.
With BACKEND = nccl (one rank: RCCL does not take two ranks on one device) and GPA_DIST_FORCE_COLLECTIVES=1 the single rank
runs every collective of the N > 1 schedule on device tensors over RCCL.
usage: RANK= WORLD_SIZE= MASTER_ADDR= MASTER_PORT= python _stream_rank_worker.py OUT_PREFIX DTYPE [BACKEND]"""
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
    grouped = world > 1 or backend == 'nccl'
    if backend == 'nccl':
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', 0))
    elif world > 1:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    from pygpa_amd import distributed as D
    from test_distributed import _case, _stream_images
    _, kvecs, klists = _case()
    images = _stream_images(5)
    pipe = D.TiledPipeline(images[0].shape, kvecs, np.stack(klists), 6, 20, kmax=10, dtype=dtype, device=0, grid=(2, 2))
    ref = []
    for img in images:
        pipe.load(img)
        ref.append(pipe.step().cpu().numpy().copy())
    got = {}
    iters = pipe.run_stream(images, on_result=lambda i, u: got.__setitem__(i, u.cpu().numpy().copy()))
    ok = True
    for i in range(len(images)):
        a, _ = pipe.owners(i)
        if rank == a:
            ok = ok and i in got and np.array_equal(got[i], ref[i])
        else:
            ok = ok and i not in got
    np.savez(out_prefix + '_rank%d.npz' % rank, ok=ok, seen=np.array(sorted(got)), ref0=ref[0],
             iters=np.array([[-1 if v is None else v for v in it] for it in iters]))
    pipe.close()
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
