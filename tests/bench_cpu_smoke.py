"""Helper of tests/test_multirank_gloo.py (not a test, not part of the product): runs bench.py's main() with the
oracle engine behind Master and a gloo transport, so that `bench.py --gpus N` executes end to end on CPU at the
driver's world size -- rendezvous, timed region, max over ranks, rank-0 JSON assembly, exit of the other ranks."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench                                         # noqa: E402
from epstan_amd import dist                          # noqa: E402
from oracle.engine_oracle import OracleEngine        # noqa: E402


def _engine(model, X, y, k_lim, **kw):
    return OracleEngine(model, X, y, k_lim, nthreads=1, **kw)


def _comm(rank, world):
    import torch.distributed as tdist
    tdist.init_process_group('gloo')
    return dist.TorchComm()


if __name__ == '__main__':
    bench._ENGINE_FACTORY = _engine
    bench._COMM_FACTORY = _comm
    try:
        bench.main()
    finally:
        # this helper made the process group, so it ends it: a rank that leaves the interpreter with gloo's threads
        # still running dies now and then with "terminate called without an active exception" (seen in 1 run of 12)
        import torch.distributed as tdist
        if tdist.is_available() and tdist.is_initialized():
            tdist.destroy_process_group()
