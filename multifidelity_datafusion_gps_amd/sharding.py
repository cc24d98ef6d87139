"""Rank plumbing for the parts of the path that shard: one process per GPU,
`torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).

What shards, and how:
  * optimiser restarts (src/abstractMFGP.py:137): independent L-BFGS-B runs -> restart i on rank i % size,
    one all-gather of (f_opt, x_opt) -- a few dozen bytes per rank;
  * predictive panels K(X*, X): X* rows are split across ranks, every rank holds the (replicated,
    redundantly factorised) level state, one all-gather of 16 B per test row (mean + variance).
The Cholesky itself does not shard at N <= 16384 (sequential panel dependency): replicas only.
torch is imported lazily and only here: it is plumbing, never on the arithmetic path.
"""
import numpy as np


class LocalComm:
    """size-1 communicator: the default everywhere."""
    rank, size = 0, 1

    def allgather_object(self, obj):
        return [obj]

    def allgather_rows(self, arr):
        return np.asarray(arr)

    def barrier(self):
        pass

    def bcast_object(self, obj, src=0):
        return obj


class TorchComm:
    """torch.distributed-backed communicator (process group must already be initialised)."""

    def __init__(self, device=None):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self._torch, self._dist = torch, dist
        self.rank, self.size = dist.get_rank(), dist.get_world_size()
        self.backend = dist.get_backend()
        if device is None:
            device = "cuda:%d" % torch.cuda.current_device() if self.backend == "nccl" else "cpu"
        self.device = device

    def allgather_object(self, obj):
        out = [None] * self.size
        self._dist.all_gather_object(out, obj)
        return out

    def allgather_rows(self, arr):
        """concatenate per-rank row blocks (ragged allowed: counts are exchanged first, blocks padded)"""
        torch, dist = self._torch, self._dist
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        tail = arr.shape[1:]
        counts = self.allgather_object(int(arr.shape[0]))
        m = max(counts) if counts else 0
        pad = np.zeros((m,) + tail)
        pad[:arr.shape[0]] = arr
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


class _DevArray:
    """zero-copy view of a device buffer for torch (the CUDA array interface torch.as_tensor understands)"""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def eval_rowblock_allgather(engine, comm, theta, noise, jitter=1e-8, want_grad=True):
    """One objective(+gradient) evaluation with the K(X,X) build sharded by row blocks:
    rank r builds its block of full rows of Ky on its GPU, the blocks are all-gathered over RCCL/xGMI IN PLACE in
    every rank's device matrix, then every rank factorises (the Cholesky itself does not shard at these sizes).
    Needs the padded size to split into equal 64-row multiples per rank; falls back to the local build otherwise.
    At N = 8192 on 8 GPUs each rank receives 470 MB to save < 0.15 ms of local K-build: this path is provided
    because the layout is what a DISTRIBUTED factorisation would start from, not because it is faster here."""
    ptr, npad = engine.dev_matrix()
    size, rank = comm.size, comm.rank
    if size == 1 or npad % (64 * size) != 0:
        if size == 1:
            engine.kbuild_rows(theta, noise, jitter, 0, npad)
            return engine.eval_prebuilt(want_grad)
        return engine.eval(theta, noise, jitter, want_grad)
    rows = npad // size
    engine.kbuild_rows(theta, noise, jitter, rank * rows, (rank + 1) * rows)
    import torch
    import torch.distributed as dist
    full = torch.as_tensor(_DevArray(ptr, (npad, npad)), device="cuda")
    dist.all_gather_into_tensor(full, full[rank * rows:(rank + 1) * rows].clone())
    torch.cuda.synchronize()
    return engine.eval_prebuilt(want_grad)


def split_rows(n, rank, size):
    """contiguous, balanced [begin, end) of n rows for `rank`"""
    base, rem = divmod(n, size)
    b = rank * base + min(rank, rem)
    return b, b + base + (1 if rank < rem else 0)
