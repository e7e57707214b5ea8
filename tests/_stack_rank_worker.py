"""One rank of the frames-over-ranks test (tests/test_gpu_configs.py): torch.distributed over gloo, the ranks share
cuda:0, every rank runs the device stack call on its block of frames.
usage: RANK= WORLD_SIZE= MASTER_ADDR= MASTER_PORT= python _stack_rank_worker.py OUT_PREFIX DTYPE"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    out_prefix, dtype = sys.argv[1], np.dtype(sys.argv[2])
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from pygpa_amd import distributed as D
    from test_distributed import _stack_case
    frames, kvecs, klists = _stack_case()
    u = D.extract_displacement_field_stack_sharded(frames, kvecs, klists=klists, dtype=dtype, device=0)
    np.save(out_prefix + '_rank%d.npy' % rank, u)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
