"""Rank plumbing for the parts of the path that shard: one process per GPU.

What shards, and how:
  * optimiser restarts (src/abstractMFGP.py:137): independent L-BFGS-B runs -> restart i on rank i % size (or the
    least-loaded rank), one all-gather of (f_opt, x_opt) -- a few dozen bytes per rank;
  * predictive panels K(X*, X): X* rows are split across ranks, every rank holds the (replicated, redundantly
    factorised) level state, one all-gather of 16 B per test row (mean + variance);
  * the K(X,X) row-block build of north_star: every rank builds its block of full rows, one in-place all-gather.
The Cholesky itself does not shard at N <= 16384 (sequential panel dependency): replicas only.

Transports.  The DEVICE collectives are RCCL inside libmfgp_hip.so (csrc/comm_rccl.hip: mfgp_allgather_rows,
mfgp_allgather_host), on the engine's own stream.  The HOST side needs only a rendezvous and a few tiny object
gathers; `SocketComm` does that over plain TCP on 127.0.0.1 (rank 0 is the hub) -- no PyTorch anywhere on the
product's multi-GPU path.  `TorchComm` remains as the gloo communicator of the CPU tests
(tests/test_sharding_gloo.py) and is the only place in the package that imports torch.
"""
import os
import pickle
import socket
import struct
import time

import numpy as np


class LocalComm:
    """size-1 communicator: the default everywhere."""
    rank, size = 0, 1
    transport = "local"

    def allgather_object(self, obj):
        return [obj]

    def allgather_rows(self, arr):
        return np.asarray(arr)

    def barrier(self):
        pass

    def bcast_object(self, obj, src=0):
        return obj

    def close(self):
        pass


def _pad_and_counts(comm, arr):
    """common part of the ragged row gathers: exchange the per-rank row counts, pad the block to the largest"""
    arr = np.ascontiguousarray(arr, dtype=np.float64)
    counts = comm.allgather_object(int(arr.shape[0]))
    m = max(counts) if counts else 0
    pad = np.zeros((m,) + arr.shape[1:])
    pad[:arr.shape[0]] = arr
    return pad, counts


class SocketComm:
    """Host-side communicator over TCP (hub and spokes, rank 0 = hub): rendezvous, object all-gather, barrier.

    Every collective is one round trip of pickled payloads to the hub and back; they carry restart results, row
    counts, timing scalars and the 128-byte RCCL unique id -- never matrices.  With `attach_engine` the row gathers of
    the data path go through RCCL on the engine's stream instead (`transport` says which one is in use)."""

    def __init__(self, rank, size, addr="127.0.0.1", port=29650, timeout=120.0):
        self.rank, self.size = int(rank), int(size)
        self.transport = "tcp"
        self._engine = None
        self._peers = []       # hub: sockets of ranks 1..size-1, in rank order
        self._hub = None       # spoke: socket to rank 0
        if self.size == 1:
            return
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, int(port)))
            srv.listen(self.size)
            srv.settimeout(timeout)
            got = {}
            while len(got) < self.size - 1:
                conn, _ = srv.accept()
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                conn.settimeout(None)
                peer = self._recv(conn)
                got[int(peer)] = conn
            srv.close()
            self._peers = [got[r] for r in range(1, self.size)]
        else:
            deadline = time.time() + timeout
            while True:
                try:
                    s = socket.create_connection((addr, int(port)), timeout=5.0)
                    break
                except OSError:
                    if time.time() > deadline:
                        raise
                    time.sleep(0.05)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.settimeout(None)
            self._send(s, self.rank)
            self._hub = s

    # ---- framing -----------------------------------------------------------------------------------------
    @staticmethod
    def _send(sock, obj):
        blob = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
        sock.sendall(struct.pack("<Q", len(blob)) + blob)

    @staticmethod
    def _recv(sock):
        def exactly(n):
            chunks, left = [], n
            while left:
                c = sock.recv(min(left, 1 << 20))
                if not c:
                    raise ConnectionError("peer closed the connection")
                chunks.append(c)
                left -= len(c)
            return b"".join(chunks)
        (n,) = struct.unpack("<Q", exactly(8))
        return pickle.loads(exactly(n))

    # ---- collectives -------------------------------------------------------------------------------------
    def allgather_object(self, obj):
        if self.size == 1:
            return [obj]
        if self.rank == 0:
            out = [obj] + [self._recv(p) for p in self._peers]
            for p in self._peers:
                self._send(p, out)
            return out
        self._send(self._hub, obj)
        return self._recv(self._hub)

    def bcast_object(self, obj, src=0):
        return self.allgather_object(obj if self.rank == src else None)[src]

    def barrier(self):
        self.allgather_object(None)

    def attach_engine(self, engine, init_timeout=120.0):
        """collective: create the RCCL communicator inside `engine`'s handle (unique id from rank 0 over TCP); from then
        on the row gathers run as ncclAllGather on that engine's stream.  All or nothing: if the initialisation fails --
        or does not return within `init_timeout` seconds -- on ANY rank, every rank stays on TCP (a half-attached job
        would hang in its first collective).  Only the ncclCommInitRank call itself runs under the watchdog; every TCP
        exchange stays on the calling thread."""
        import threading
        uid = None
        err = None
        if self.rank == 0:
            try:
                uid = engine.comm_unique_id()
            except Exception as e:  # noqa: BLE001 - reported, then agreed on by all ranks
                err = repr(e)
        uid, err = self.bcast_object((uid, err))
        if uid is not None:
            box = {}

            def init():
                try:
                    engine.comm_init(uid, self.rank, self.size)
                    box["ok"] = True
                except Exception as e:  # noqa: BLE001
                    box["err"] = repr(e)

            t = threading.Thread(target=init, daemon=True)   # daemon: a hung initialisation must not keep the process alive
            t.start()
            t.join(init_timeout)
            if t.is_alive():
                err = "ncclCommInitRank did not return within %.0f s on rank %d" % (init_timeout, self.rank)
            elif "err" in box:
                err = box["err"]
        errs = [e for e in self.allgather_object(err) if e]
        if errs:
            self.rccl_error = errs[0]
            return False
        self._engine = engine
        self.transport = "rccl"
        return True

    def allgather_rows(self, arr):
        """concatenate per-rank row blocks (ragged allowed: counts are exchanged first, blocks padded)"""
        if self.size == 1:
            return np.asarray(arr)
        pad, counts = _pad_and_counts(self, arr)
        if self._engine is not None:
            flat = self._engine.allgather_host(pad.reshape(-1))            # RCCL, on the engine's stream
            blocks = [flat[r].reshape(pad.shape) for r in range(self.size)]
        else:
            blocks = self.allgather_object(pad)
        return np.concatenate([b[:c] for b, c in zip(blocks, counts)], axis=0)

    def close(self):
        for s in self._peers + ([self._hub] if self._hub is not None else []):
            try:
                s.close()
            except OSError:
                pass
        self._peers, self._hub = [], None


def comm_from_env(timeout=120.0):
    """the communicator of a process launched one-per-GPU (torch.distributed.run / any launcher that exports RANK,
    WORLD_SIZE, MASTER_ADDR, MASTER_PORT).  The launcher's own store owns MASTER_PORT, so the hub listens on
    MFGP_COMM_PORT if set, else MASTER_PORT + 1000 (wrapped into the valid range)."""
    size = int(os.environ.get("WORLD_SIZE", "1"))
    if size == 1:
        return LocalComm()
    rank = int(os.environ["RANK"])
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    base = int(os.environ.get("MASTER_PORT", "29500"))
    port = int(os.environ.get("MFGP_COMM_PORT", base + 1000 if base + 1000 < 65536 else base - 1000))
    return SocketComm(rank, size, addr, port, timeout=timeout)


class TorchComm:
    """torch.distributed-backed communicator, kept for the gloo CPU tests only (process group already initialised)."""
    transport = "torch.distributed"

    def __init__(self, device=None):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self._torch, self._dist = torch, dist
        self.rank, self.size = dist.get_rank(), dist.get_world_size()
        self.backend = dist.get_backend()
        self.device = device or "cpu"

    def allgather_object(self, obj):
        out = [None] * self.size
        self._dist.all_gather_object(out, obj)
        return out

    def allgather_rows(self, arr):
        torch, dist = self._torch, self._dist
        pad, counts = _pad_and_counts(self, arr)
        t = torch.from_numpy(pad).to(self.device)
        outs = [torch.empty_like(t) for _ in range(self.size)]
        dist.all_gather(outs, t)
        return np.concatenate([o.cpu().numpy()[:c] for o, c in zip(outs, counts)], axis=0)

    def barrier(self):
        self._dist.barrier()

    def bcast_object(self, obj, src=0):
        box = [obj]
        self._dist.broadcast_object_list(box, src=src)
        return box[0]

    def close(self):
        pass


def eval_rowblock_allgather(engine, comm, theta, noise, jitter=1e-8, want_grad=True):
    """One objective(+gradient) evaluation with the K(X,X) build sharded by row blocks:
    rank r builds its block of full rows of Ky on its GPU, the blocks are all-gathered IN PLACE in every rank's device
    matrix -- ncclAllGather over xGMI inside the library when the engine carries an RCCL communicator
    (comm.attach_engine), through host memory and the communicator's object gather otherwise (multi-process tests on
    a one-GPU box) -- then every rank factorises (the Cholesky itself does not shard at these sizes).
    Needs the padded size to split into equal 64-row multiples per rank; falls back to the local build otherwise.
    At N = 8192 on 8 GPUs each rank receives 470 MB to save < 0.15 ms of local K-build: this path is provided
    because the layout is what a DISTRIBUTED factorisation would start from, not because it is faster here."""
    _, npad = engine.dev_matrix()
    size, rank = comm.size, comm.rank
    if size == 1:
        engine.kbuild_rows(theta, noise, jitter, 0, npad)
        return engine.eval_prebuilt(want_grad)
    if npad % (64 * size) != 0:
        return engine.eval(theta, noise, jitter, want_grad)
    rows = npad // size
    engine.kbuild_rows(theta, noise, jitter, rank * rows, (rank + 1) * rows)
    if getattr(comm, "_engine", None) is engine and engine.comm_size == size:
        engine.allgather_rows()
    else:
        blocks = comm.allgather_object(engine.rows_download(rank * rows, (rank + 1) * rows))
        for r, block in enumerate(blocks):
            if r != rank:
                engine.rows_upload(r * rows, block)
    return engine.eval_prebuilt(want_grad)


def split_rows(n, rank, size):
    """contiguous, balanced [begin, end) of n rows for `rank`"""
    base, rem = divmod(n, size)
    b = rank * base + min(rank, rem)
    return b, b + base + (1 if rank < rem else 0)
