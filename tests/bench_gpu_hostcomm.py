"""Helper of tests/test_gpu_multirank.py (not a test, not part of the product): runs bench.py's main() on the DEVICE with
N ranks sharing GPU 0 -- the HIP engine and the library's own multi-rank update code (epx_update_trial with per-rank
statistics slots, site offsets, flag reductions) with every collective handed to gloo (dist.HostComm), because RCCL refuses
two ranks on one device.  What it proves: `bench.py --gpus N` gets through warm-up, timed region, the collective parity
iteration, the exit of ranks != 0 and rank 0's CPU leg, and prints its one line."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ['LOCAL_RANK'] = '0'                       # every rank on the box's one device

import bench                                         # noqa: E402
from epstan_amd import dist                          # noqa: E402


def _comm(rank, world):
    import torch.distributed as tdist
    tdist.init_process_group('gloo')
    return dist.HostComm(dist.TorchComm())


if __name__ == '__main__':
    bench._COMM_FACTORY = _comm
    try:
        bench.main()
    finally:
        import torch.distributed as tdist
        if tdist.is_available() and tdist.is_initialized():
            tdist.destroy_process_group()
