"""Multi-GPU plumbing: contiguous site sharding + the one all-reduce per EP
iteration.

`EpxComm` is the product path: RCCL inside libepx.so (epx_comm_init, include/epx.h), the
reduction of /root/reference/epstan/method.py:1073-1074 running on the engine's own HIP
stream.  `TorchComm` moves host buffers through an already initialised torch.distributed
group (gloo in the CPU tests); `LocalComm` is the single-process identity.
"""

import os
import socket
import struct
import time

import numpy as np

from . import _lib


def site_range(K, rank, world):
    """Contiguous block of sites owned by `rank`."""
    return (K * rank) // world, (K * (rank + 1)) // world


class LocalComm(object):
    """Single process: every collective is the identity."""
    rank = 0
    world = 1
    native = True            # the engine's fused update (epx_update_trial) covers it

    def allreduce_sum(self, x):
        return x

    def allreduce_min_int(self, v):
        return int(v)

    def allreduce_max(self, arr):
        return np.asarray(arr, dtype=np.float64)

    def allgather_sites(self, local, K):
        return local

    def barrier(self):
        pass


def _gather_layout(local, K, world):
    lead = local.shape[:-1]
    per = int(np.prod(lead)) if lead else 1
    counts = [site_range(K, r, world) for r in range(world)]
    kmax = max(hi - lo for lo, hi in counts)
    buf = np.zeros(per * kmax)
    flat = np.asarray(local).reshape(-1, order='F')
    buf[:flat.shape[0]] = flat
    return lead, per, counts, kmax, buf


def _scatter_gathered(parts, lead, per, counts, K):
    full = np.empty(lead + (K,), order='F')
    for (lo, hi), part in zip(counts, parts):
        full[..., lo:hi] = part[:per * (hi - lo)].reshape(lead + (hi - lo,), order='F')
    return full


class EpxComm(object):
    """RCCL communicator held by libepx.so: one process per GPU, sites sharded over the ranks.

    Rank and world size default to torchrun's environment (RANK / WORLD_SIZE).  The 128-byte RCCL id goes from rank 0
    to the others through a channel that already exists when there is one -- the key-value store torchrun's agent
    serves on MASTER_ADDR:MASTER_PORT (no second port, nothing to bind) -- and otherwise over a plain TCP connection to
    MASTER_ADDR:EPX_COMM_PORT (default MASTER_PORT + 23; `bench.py` picks and verifies a free one), where rank 0
    counts a peer only once it has acknowledged the id.
    `Master` binds the communicator to its engine (`bind`, collective over all ranks)."""
    native = True
    _binds = 0                # binds so far in this process: every rank binds in the same order -> a common key

    def __init__(self, rank=None, world=None, addr=None, port=None):
        env = os.environ
        self.rank = int(env.get('RANK', '0')) if rank is None else int(rank)
        self.world = int(env.get('WORLD_SIZE', '1')) if world is None else int(world)
        self.addr = addr or env.get('MASTER_ADDR', '127.0.0.1')
        if port is None:
            port = int(env['EPX_COMM_PORT']) if 'EPX_COMM_PORT' in env else int(env.get('MASTER_PORT', '29500')) + 23
        self.port = int(port)
        self.engine = None
        if not 0 <= self.rank < self.world:
            raise ValueError('rank {} outside 0..{}'.format(self.rank, self.world - 1))

    # ---- bootstrap
    def _store_exchange(self, uid, seq):
        """Through torchrun's store (TORCHELASTIC_USE_AGENT_STORE): returns the id, or None when there is no such store.
        Which channel carries the id is decided from the environment alone, the same way on every rank: with the store
        announced, an error talking to it RAISES (no per-rank fallback to the socket, which rank 0 would then not be
        serving).  The store's client is PyTorch's; it runs in a short-lived CHILD process, so that a rank never loads
        PyTorch's bundled HIP / RCCL runtime beside the ROCm one libepx.so is linked against (the timed run and the parity
        tests execute on one runtime)."""
        env = os.environ
        if env.get('TORCHELASTIC_USE_AGENT_STORE', '') != 'True' or 'MASTER_PORT' not in env:
            return None
        import subprocess
        import sys
        key = 'epx_comm_id/%s/%s/%d' % (env.get('TORCHELASTIC_RUN_ID', '-'), env.get('TORCHELASTIC_RESTART_COUNT', '0'), seq)
        child = ('import sys\n'
                 'from datetime import timedelta\n'
                 'from torch.distributed import TCPStore\n'
                 'addr, port, world, key, uid = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]\n'
                 'store = TCPStore(addr, port, world, False, timedelta(seconds=300))\n'
                 'if uid:\n'
                 '    store.set(key, bytes.fromhex(uid))\n'
                 '    print("ok")\n'
                 'else:\n'
                 '    print(bytes(store.get(key)).hex())\n')
        cenv = dict(env)
        cenv['CUDA_VISIBLE_DEVICES'] = ''          # the child talks to a TCP store: it needs no device
        cenv['HIP_VISIBLE_DEVICES'] = ''
        # (the child imports torch: 2-3 s once the image is paged in, up to a minute or two on a fresh box -- hence the limit)
        try:
            out = subprocess.run([sys.executable, '-c', child, self.addr, env['MASTER_PORT'], str(self.world), key,
                                  uid.hex() if self.rank == 0 else ''], env=cenv, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                 timeout=600)
        except subprocess.TimeoutExpired as ex:
            raise RuntimeError('epx: the RCCL id did not travel through torchrun\'s store (%s:%s) within 600 s (rank %d of %d): %s'
                               % (self.addr, env['MASTER_PORT'], self.rank, self.world,
                                  (ex.stderr or b'').decode('utf-8', 'replace')[-400:]))
        if out.returncode != 0:
            raise RuntimeError('epx: the RCCL id could not travel through torchrun\'s store (%s:%s): %s'
                               % (self.addr, env['MASTER_PORT'], out.stderr.decode('utf-8', 'replace')[-400:]))
        if self.rank == 0:
            return uid
        got = bytes.fromhex(out.stdout.decode().strip().splitlines()[-1])
        if len(got) != _lib.COMM_ID_BYTES:
            raise RuntimeError('epx: the RCCL id from the store has %d bytes' % len(got))
        return got

    def _exchange_id(self, uid):
        """Rank 0 serves `uid` to every other rank; the others fetch it (retrying while rank 0
        is not listening yet)."""
        n = _lib.COMM_ID_BYTES
        if self.world == 1:
            return uid
        seq = EpxComm._binds
        EpxComm._binds += 1
        got = self._store_exchange(uid, seq)
        if got is not None:
            return got
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((self.addr, self.port))
            srv.listen(self.world)
            srv.settimeout(300.0)
            try:
                seen = set()
                while len(seen) < self.world - 1:
                    conn, _ = srv.accept()
                    with conn:
                        try:
                            conn.settimeout(60.0)
                            peer = struct.unpack('<i', _recv_exact(conn, 4))[0]
                            conn.sendall(uid)
                            if _recv_exact(conn, 1) == b'k':      # counted only once the peer HAS the id (it reconnects otherwise)
                                seen.add(peer)
                        except (ConnectionResetError, socket.timeout, OSError):
                            pass
            finally:
                srv.close()
            return uid
        deadline = time.time() + 300.0
        while True:
            try:
                with socket.create_connection((self.addr, self.port), timeout=10.0) as conn:
                    conn.sendall(struct.pack('<i', self.rank))
                    got = _recv_exact(conn, n)
                    conn.sendall(b'k')
                    return got
            except (ConnectionRefusedError, ConnectionResetError, socket.timeout, OSError):
                if time.time() > deadline:
                    raise
                time.sleep(0.05)

    def bind(self, engine):
        """Create the RCCL communicator of `engine`'s context (collective)."""
        import ctypes
        lib = _lib.load()
        if not hasattr(engine, 'ctx'):
            raise TypeError('EpxComm needs the HIP engine (a context of libepx.so)')
        uid = b''
        if self.rank == 0:
            buf = ctypes.create_string_buffer(_lib.COMM_ID_BYTES)
            _lib.check(lib.epx_comm_unique_id(buf))
            uid = buf.raw
        uid = self._exchange_id(uid)
        _lib.check(lib.epx_comm_init(engine.ctx, uid, self.rank, self.world))
        self.engine = engine
        return self

    def size(self):
        """(rank, number of ranks) as RCCL reports them for the bound communicator."""
        return self.engine.comm_size()

    def close(self):
        if self.engine is not None and getattr(self.engine, 'ctx', None):
            _lib.load().epx_comm_destroy(self.engine.ctx)
        self.engine = None

    # ---- host-side collectives (small; the per-iteration reduction itself is inside epx_update_trial)
    def _reduce(self, arr, op):
        a = np.ascontiguousarray(arr, dtype=np.float64).copy()
        if a.size:
            self.engine.comm_allreduce(a.reshape(-1), op)
        return a

    def allreduce_sum(self, x):
        x[...] = self._reduce(x, _lib.OP_SUM).reshape(x.shape)
        return x

    def allreduce_min_int(self, v):
        return int(self._reduce(np.array([float(int(v))]), _lib.OP_MIN)[0])

    def allreduce_max(self, arr):
        return self._reduce(np.asarray(arr, dtype=np.float64), _lib.OP_MAX)

    def allgather_sites(self, local, K):
        """local: F-ordered (..., K_local) slice; returns the full (..., K) array."""
        lead, per, counts, kmax, buf = _gather_layout(local, K, self.world)
        out = self.engine.comm_allgather(buf, self.world)
        return _scatter_gathered([out[r] for r in range(self.world)], lead, per, counts, K)

    def barrier(self):
        self._reduce(np.zeros(1), _lib.OP_SUM)


class HostComm(EpxComm):
    """The library's collectives over a HOST transport (epx_comm_init_host): `Master` runs exactly the code of
    the RCCL case -- the fused update with its per-rank statistics slots, site offsets and flag reductions -- and
    every reduction is handed to `transport`, a TorchComm (gloo, MPI-backed groups, ...) or any object with
    `rank`, `world` and `reduce(array, op)` reducing a float64 array in place.  For nodes whose GPUs have no
    RCCL peer path, and for the world-size-2 tests on one GPU."""

    def __init__(self, transport):
        self.transport = transport
        self.rank, self.world = int(transport.rank), int(transport.world)
        self.engine = None
        self._cb = None

    def bind(self, engine):
        import ctypes
        lib = _lib.load()
        if not hasattr(engine, 'ctx'):
            raise TypeError('HostComm needs the HIP engine (a context of libepx.so)')
        tr = self.transport

        def reduce_cb(buf, n, op, _user):
            try:
                a = np.ctypeslib.as_array(buf, shape=(int(n),))
                tr.reduce(a, int(op))
                return 0
            except Exception:                      # an exception must not unwind through the C frames
                import traceback
                traceback.print_exc()
                return 1
        self._cb = _lib.HOST_ALLREDUCE_FN(reduce_cb)          # keep the thunk alive as long as the context
        _lib.check(lib.epx_comm_init_host(engine.ctx, self.rank, self.world, self._cb, None))
        self.engine = engine
        return self


def _recv_exact(conn, n):
    buf = b''
    while len(buf) < n:
        chunk = conn.recv(n - len(buf))
        if not chunk:
            raise ConnectionResetError('peer closed the id exchange')
        buf += chunk
    return buf


class TorchComm(object):
    """Host buffers through a torch.distributed process group that the caller initialised
    (gloo in the CPU tests; any backend works, device tensors are not involved)."""
    native = False

    def __init__(self, group=None, device=None):
        import torch
        import torch.distributed as dist
        self.torch = torch
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.device = device      # staging device of the collectives (None: host tensors, gloo)

    def _small(self, arr):
        t = self.torch.as_tensor(np.ascontiguousarray(arr))
        if self.device is not None:
            t = t.to(self.device)
        return t

    def allreduce_sum(self, x):
        t = self._small(x)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        x[...] = t.cpu().numpy()
        return x

    def allreduce_min_int(self, v):
        t = self._small(np.array([int(v)], dtype=np.int64))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.group)
        return int(t.cpu()[0])

    def allreduce_max(self, arr):
        t = self._small(np.asarray(arr, dtype=np.float64))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return t.cpu().numpy()

    def allgather_sites(self, local, K):
        """local: F-ordered (..., K_local) slice; returns the full (..., K) array."""
        lead, per, counts, kmax, buf = _gather_layout(local, K, self.world)
        t = self._small(buf)
        outs = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(outs, t, group=self.group)
        return _scatter_gathered([o.cpu().numpy() for o in outs], lead, per, counts, K)

    def barrier(self):
        self.dist.barrier(group=self.group)

    def reduce(self, a, op):
        """In-place reduction of a float64 array (the transport interface of HostComm)."""
        t = self._small(a)
        ops = {_lib.OP_SUM: self.dist.ReduceOp.SUM, _lib.OP_MIN: self.dist.ReduceOp.MIN, _lib.OP_MAX: self.dist.ReduceOp.MAX}
        self.dist.all_reduce(t, op=ops[op], group=self.group)
        a[...] = t.cpu().numpy()
