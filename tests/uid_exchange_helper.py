"""Helper of tests/test_multirank_gloo.py: under torchrun, every rank runs dist.EpxComm's id exchange twice."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from epstan_amd import dist                          # noqa: E402

if __name__ == '__main__':
    rank = int(os.environ['RANK'])
    assert os.environ.get('TORCHELASTIC_USE_AGENT_STORE') == 'True'
    c = dist.EpxComm()
    first = c._exchange_id(bytes([7]) * 128 if rank == 0 else b'')
    second = c._exchange_id(bytes(range(128)) if rank == 0 else b'')
    assert first == bytes([7]) * 128
    print('uid %d %s' % (rank, second.hex()), flush=True)
