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

Wire safety (round 3).  Nothing received from a socket is ever unpickled: frames carry a closed, non-executable
encoding (`wire_encode` / `wire_decode`: None, bool, int, float, str, bytes, list, tuple, str-keyed dict, numpy
arrays of plain numeric dtypes).  A connection is admitted to the rendezvous only after a mutual HMAC-SHA256
challenge-response on a per-job token (MFGP_COMM_TOKEN; without it, a 0600 token file under a 0700 per-user
directory, single node only), and the hub binds loopback unless a multi-node address is explicitly requested.
"""
import hashlib
import hmac
import os
import secrets
import socket
import struct
import time

import numpy as np


# ---- wire format: a closed set of plain data types, nothing executable ----------------------------------------
_WIRE_DTYPES = ("<f8", "<f4", "<i8", "<i4", "<u8", "<u4", "|u1", "|b1")
_WIRE_MAX_FRAME = 1 << 32     # 4 GiB: row blocks of the one-GPU rehearsal travel through here; matrices never do


def _enc(obj, out):
    if obj is None:
        out.append(b"N")
    elif isinstance(obj, (bool, np.bool_)):
        out.append(b"T" if obj else b"F")
    elif isinstance(obj, (int, np.integer)):
        if not -(1 << 63) <= int(obj) < (1 << 63):
            raise TypeError("integer %d does not fit the wire format's int64" % int(obj))
        out.append(b"i" + struct.pack("<q", int(obj)))
    elif isinstance(obj, (float, np.floating)):
        out.append(b"d" + struct.pack("<d", float(obj)))
    elif isinstance(obj, str):
        b = obj.encode("utf-8")
        out.append(b"s" + struct.pack("<Q", len(b)) + b)
    elif isinstance(obj, (bytes, bytearray)):
        out.append(b"b" + struct.pack("<Q", len(obj)) + bytes(obj))
    elif isinstance(obj, np.ndarray):
        a = np.require(obj, requirements="C")        # (ascontiguousarray would turn a 0-d array into shape (1,))
        dt = a.dtype.newbyteorder("<").str if a.dtype.byteorder == ">" else a.dtype.str
        if dt not in _WIRE_DTYPES:
            raise TypeError("array dtype %s is not part of the wire format" % a.dtype)
        if a.ndim > 8:
            raise TypeError("arrays of more than 8 dimensions are not part of the wire format")
        a = a.astype(np.dtype(dt), copy=False)
        out.append(b"a" + struct.pack("<BB", _WIRE_DTYPES.index(dt), a.ndim) + struct.pack("<%dQ" % a.ndim, *a.shape))
        out.append(a.tobytes())
    elif isinstance(obj, (list, tuple)):
        out.append((b"l" if isinstance(obj, list) else b"t") + struct.pack("<Q", len(obj)))
        for v in obj:
            _enc(v, out)
    elif isinstance(obj, dict):
        out.append(b"m" + struct.pack("<Q", len(obj)))
        for k, v in obj.items():
            if not isinstance(k, str):
                raise TypeError("only str keys travel (got %r)" % type(k))
            _enc(k, out)
            _enc(v, out)
    else:
        raise TypeError("%r is not part of the wire format (plain data only)" % type(obj))


def wire_encode(obj):
    out = []
    _enc(obj, out)
    return b"".join(out)


_WIRE_MAX_DEPTH = 64      # nesting of lists / tuples / dicts a frame may carry (the payloads here nest three deep)


def _dec(buf, pos, depth=0):
    if depth > _WIRE_MAX_DEPTH:
        raise ValueError("frame nests deeper than %d levels" % _WIRE_MAX_DEPTH)
    tag = buf[pos:pos + 1]
    pos += 1
    if tag == b"N":
        return None, pos
    if tag == b"T":
        return True, pos
    if tag == b"F":
        return False, pos
    if tag == b"i":
        return struct.unpack_from("<q", buf, pos)[0], pos + 8
    if tag == b"d":
        return struct.unpack_from("<d", buf, pos)[0], pos + 8
    if tag in (b"s", b"b"):
        (n,) = struct.unpack_from("<Q", buf, pos)
        pos += 8
        if n > len(buf) - pos:
            raise ValueError("truncated frame")
        raw = bytes(buf[pos:pos + n])
        return (raw.decode("utf-8") if tag == b"s" else raw), pos + n
    if tag == b"a":
        di, nd = struct.unpack_from("<BB", buf, pos)
        pos += 2
        if di >= len(_WIRE_DTYPES) or nd > 8:
            raise ValueError("bad array header")
        shape = struct.unpack_from("<%dQ" % nd, buf, pos)
        pos += 8 * nd
        dt = np.dtype(_WIRE_DTYPES[di])
        count = 1
        for v in shape:
            count *= v                         # (python ints: no overflow; a header that claims more than the frame holds is refused)
        nbytes = count * dt.itemsize
        if nbytes > len(buf) - pos or any(v > len(buf) for v in shape if count == 0):
            raise ValueError("truncated frame")
        a = np.frombuffer(buf, dtype=dt, count=count, offset=pos).reshape(shape).copy()
        return a, pos + nbytes
    if tag in (b"l", b"t"):
        (n,) = struct.unpack_from("<Q", buf, pos)
        pos += 8
        if n > len(buf) - pos:      # every element takes at least one byte
            raise ValueError("truncated frame")
        items = []
        for _ in range(n):
            v, pos = _dec(buf, pos, depth + 1)
            items.append(v)
        return (items if tag == b"l" else tuple(items)), pos
    if tag == b"m":
        (n,) = struct.unpack_from("<Q", buf, pos)
        pos += 8
        if n > len(buf) - pos:
            raise ValueError("truncated frame")
        d = {}
        for _ in range(n):
            k, pos = _dec(buf, pos, depth + 1)
            if not isinstance(k, str):
                raise ValueError("bad map key")
            d[k], pos = _dec(buf, pos, depth + 1)
        return d, pos
    raise ValueError("unknown wire tag %r" % tag)


def wire_decode(blob):
    """-> the object; ValueError for ANY malformed frame (truncated, unknown tag, bad header, too deep): a collective that
    receives one fails with that, not with whatever the parser happened to trip over"""
    try:
        obj, pos = _dec(memoryview(blob), 0)
    except (struct.error, RecursionError, OverflowError, UnicodeDecodeError, MemoryError) as ex:
        raise ValueError("malformed frame: %s" % ex.__class__.__name__) from ex
    if pos != len(blob):
        raise ValueError("trailing bytes in frame")
    return obj


class ShardGroup:
    """a group of ranks that shares ONE evaluation at a time on `engine` (its handle holds the group's RCCL communicator)"""

    def __init__(self, engine, index, size):
        self.engine, self.index, self.size = engine, int(index), int(size)
        self.calibration = None          # the measured cost of a collective of this group (Engine.comm_calibrate), where it was formed here

    @property
    def leads(self):
        return self.index == 0

    def run(self, model, fn):
        """the leader runs `fn()` -- any sequence of optimiser runs on `model` -- with every evaluation of the model shared by
        the group; the other members serve until it is through.  -> fn()'s result on the leader, None on the others."""
        if not self.leads:
            self.engine.sharded_serve()
            return None
        model._eval_hook = lambda th, nz, jit: self.engine.sharded_lead(th, nz, jit, want_grad=True)
        try:
            return fn()
        finally:
            model._eval_hook = None
            # a pass that failed after the group had been told to start it has aborted the communicator (mfgp_comm_state -1): the
            # followers are somewhere inside that pass and a release would be one more collective nobody matches.  The exception
            # propagates, the rank ends with an error and its launcher stops the peers (bench.launch_ranks / torchrun); a follower
            # left alone gives up after MFGP_SHARD_TIMEOUT_S on its own.
            if not getattr(self.engine, "comm_aborted", False):
                self.engine.sharded_release()


class RcclInitError(RuntimeError):
    """the RCCL communicator could not be created on every rank.  `.hung` says that this rank's ncclCommInitRank never
    returned: the engine handle is then still in use by the stuck thread and must not be touched again -- the only
    safe continuation is to end the process (a launcher starts fresh ones)."""

    def __init__(self, msg, hung=False):
        super().__init__(msg)
        self.hung = hung


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

    def __init__(self, rank, size, addr="127.0.0.1", port=29650, timeout=120.0, token=None):
        self.rank, self.size = int(rank), int(size)
        self.transport = "tcp"
        self.calibration = None          # Engine.comm_calibrate of the world communicator (attach_engine)
        self._engine = None
        self._peers = []       # hub: sockets of ranks 1..size-1, in rank order
        self._hub = None       # spoke: socket to rank 0
        self._shard_group_cache = []     # (engine, members, ShardGroup | None): the groups formed so far (shard_group)
        self.groups_formed = 0           # communicators created for groups: a job that re-uses its engines forms each group once
        if self.size == 1:
            return
        loopback = addr in ("127.0.0.1", "localhost", "::1")
        self._token_file = None
        if self.rank == 0:
            _job_token.last_file = None
            key = _job_token(token, port, create=True, loopback=loopback)
            self._token_file = _job_token.last_file      # (None when the launcher provided the token)
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            # loopback unless a multi-node address was asked for explicitly (then that interface, never 0.0.0.0)
            srv.bind(("127.0.0.1" if loopback else addr, int(port)))
            srv.listen(self.size)
            deadline = time.time() + timeout
            got = {}
            while len(got) < self.size - 1:
                srv.settimeout(max(deadline - time.time(), 0.01))
                conn, _ = srv.accept()      # socket.timeout ends a rendezvous that never completes
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                conn.settimeout(10.0)
                try:
                    peer = self._admit(conn, key)
                except (OSError, ValueError, ConnectionError):
                    peer = None
                if peer is None or not (0 < peer < self.size) or peer in got:
                    conn.close()            # not one of ours (or a duplicate): dropped before anything is decoded
                    continue
                conn.settimeout(None)
                got[peer] = conn
            srv.close()
            self._peers = [got[r] for r in range(1, self.size)]
        else:
            deadline = time.time() + timeout
            while True:
                s = None
                try:
                    key = _job_token(token, port, create=False, loopback=loopback)
                    s = socket.create_connection((addr, int(port)), timeout=5.0)
                    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    s.settimeout(10.0)
                    self._introduce(s, key, self.rank)
                    break
                except (OSError, ValueError, ConnectionError):
                    # hub not up yet, token file not written yet, or a stale token of an earlier job: try again
                    if s is not None:
                        s.close()
                    if time.time() > deadline:
                        raise
                    time.sleep(0.05)
            s.settimeout(None)
            self._hub = s

    # ---- admission: mutual HMAC-SHA256 challenge-response on the job token, fixed-size fields only ----------------
    @staticmethod
    def _exactly(sock, n):
        chunks, left = [], n
        while left:
            c = sock.recv(min(left, 1 << 20))
            if not c:
                raise ConnectionError("peer closed the connection")
            chunks.append(c)
            left -= len(c)
        return b"".join(chunks)

    @classmethod
    def _admit(cls, conn, key):
        """hub side -> the peer's rank, or None when it does not hold the token"""
        nonce = secrets.token_bytes(16)
        conn.sendall(b"MFGP1" + nonce)
        msg = cls._exactly(conn, 4 + 16 + 32)
        rank = struct.unpack("<I", msg[:4])[0]
        peer_nonce, mac = msg[4:20], msg[20:]
        want = hmac.new(key, b"spoke" + nonce + msg[:4] + peer_nonce, hashlib.sha256).digest()
        if not hmac.compare_digest(mac, want):
            return None
        conn.sendall(hmac.new(key, b"hub" + peer_nonce + msg[:4], hashlib.sha256).digest())
        return rank

    @classmethod
    def _introduce(cls, sock, key, rank):
        hello = cls._exactly(sock, 5 + 16)
        if hello[:5] != b"MFGP1":
            raise ValueError("not an mfgp rendezvous")
        mine = secrets.token_bytes(16)
        r = struct.pack("<I", rank)
        sock.sendall(r + mine + hmac.new(key, b"spoke" + hello[5:] + r + mine, hashlib.sha256).digest())
        proof = cls._exactly(sock, 32)      # the hub holds the token too (or the connection is closed on us)
        if not hmac.compare_digest(proof, hmac.new(key, b"hub" + mine + r, hashlib.sha256).digest()):
            raise ValueError("the rendezvous hub does not hold the job token")

    # ---- framing -----------------------------------------------------------------------------------------
    @staticmethod
    def _send(sock, obj):
        blob = wire_encode(obj)
        sock.sendall(struct.pack("<Q", len(blob)) + blob)

    @classmethod
    def _recv(cls, sock):
        (n,) = struct.unpack("<Q", cls._exactly(sock, 8))
        if n > _WIRE_MAX_FRAME:
            raise ValueError("frame of %d bytes refused" % n)
        return wire_decode(cls._exactly(sock, n))

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

    def attach_engine(self, engine, init_timeout=120.0, required=False):
        """collective: create the RCCL communicator inside `engine`'s handle (unique id from rank 0 over TCP); from then
        on the row gathers run as ncclAllGather on that engine's stream.  All or nothing, agreed over TCP:
          * every rank's ncclCommInitRank returned and one of them reported an error -> every rank destroys what it
            created and stays on TCP: returns False (`rccl_error` holds the first reason), or raises RcclInitError when
            `required` (a benchmark of N GPUs must not quietly become a TCP job);
          * some rank's initialisation did not return within `init_timeout` seconds -> RcclInitError on EVERY rank, with
            `.hung` set where the call is still stuck.  There the engine handle stays in use by the stuck thread (the
            handle is not thread-safe; the thread may write its communicator at any later moment), so it is poisoned
            -- every further call on it raises -- and the caller must end the process rather than continue on TCP.
        Only the ncclCommInitRank call itself runs under the watchdog; every TCP exchange stays on the calling thread."""
        import threading
        uid = None
        err = None
        hung = False
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
                hung = True
                err = "ncclCommInitRank did not return within %.0f s on rank %d" % (init_timeout, self.rank)
                if hasattr(engine, "poison"):
                    engine.poison(err)
            elif "err" in box:
                err = box["err"]
        reports = self.allgather_object((err, hung))
        errs = [e for e, _ in reports if e]
        if not errs:
            self._engine = engine
            self.transport = "rccl"
            # first contact with the links: what a small collective of this communicator costs, measured (collective; ~20 + 20
            # repetitions, milliseconds).  The figure stays on the engine handle and decides whether a shared evaluation distributes
            # its Cholesky over the ranks as well (Engine.shard_decision); bench.py prints both.
            self.calibration = engine.comm_calibrate() if hasattr(engine, "comm_calibrate") else None
            return True
        self.rccl_error = errs[0]
        if any(h for _, h in reports):
            raise RcclInitError(errs[0], hung=hung)
        if uid is not None and err is None and getattr(engine, "comm_size", 0) > 1:
            try:
                engine.comm_destroy()       # this rank's half of a communicator other ranks failed to join
            except Exception:  # noqa: BLE001
                pass
        if required:
            raise RcclInitError(errs[0])
        return False

    def shard_group(self, engine, members, init_timeout=120.0):
        """collective over ALL ranks: an RCCL communicator of the ranks `members` (ascending, members[0] leads) inside `engine`'s
        handle, for evaluations shared by that group (Engine.sharded_lead / sharded_serve: one optimiser on the leader, the
        Cholesky on every member, the rows of L^-T / K^-1 split between them).  -> ShardGroup on the members (`.index` = rank
        within the group), None on the other ranks and wherever the group could not be formed (agreed by all ranks: nobody is
        left serving a leader that never calls).  The world communicator attached by `attach_engine` is reused when `members`
        is every rank and `engine` is the engine it lives in."""
        members = [int(m) for m in members]
        if len(members) < 2 or not hasattr(engine, "sharded_lead"):
            return None
        if members == list(range(self.size)) and self._engine is engine and getattr(engine, "comm_size", 1) == self.size:
            return ShardGroup(engine, self.rank, self.size)
        if engine is self._engine:
            return None              # one communicator per handle: the world's lives here
        # A group formed before on the SAME engine handle is reused while its communicator is alive: creating one is a collective
        # (ncclCommInitRank: a rendezvous over TCP + hundreds of milliseconds on real links) and a job that builds a fresh model
        # object per fit on the same engines (bench.py does, per timed step) must not pay it inside every fit.  Every rank -- members
        # and the others -- keeps the same record, so all of them reuse or none does.
        mine = self.rank in members
        cache = self._shard_group_cache
        found = None
        for k, (eng, mem, grp) in enumerate(cache):
            if eng is engine and mem == tuple(members):
                found = k
                break
        valid = found is not None and getattr(engine, "_h", None) is not None and not getattr(engine, "comm_aborted", False) and \
            (not mine or (cache[found][2] is not None and getattr(engine, "comm_size", 1) == len(members)))
        # (one small agreement over the host rendezvous: a rank whose record went stale -- its handle closed, its communicator destroyed or
        # replaced -- makes EVERY rank form the group again; forming it is collective, a lone re-former would wait for ever)
        if all(self.allgather_object(bool(valid))):
            return cache[found][2]
        if found is not None:
            del cache[found]
        group = self._form_group(engine, members, init_timeout)
        self.groups_formed += 1
        cache.append((engine, tuple(members), group))
        return group

    def _form_group(self, engine, members, init_timeout):
        import threading
        uid, err = None, None
        if self.rank == members[0]:
            try:
                uid = engine.comm_unique_id()
            except Exception as e:  # noqa: BLE001
                err = repr(e)
        uid, err = self.bcast_object((uid, err), src=members[0])
        mine = self.rank in members
        if uid is not None and mine:
            box = {}

            def init():
                try:
                    engine.comm_init(uid, members.index(self.rank), len(members))
                    box["ok"] = True
                except Exception as e:  # noqa: BLE001
                    box["err"] = repr(e)
            t = threading.Thread(target=init, daemon=True)
            t.start()
            t.join(init_timeout)
            if t.is_alive():
                err = "ncclCommInitRank (group %s) did not return within %.0f s on rank %d" % (members, init_timeout, self.rank)
                if hasattr(engine, "poison"):
                    engine.poison(err)
            elif "err" in box:
                err = box["err"]
        errs = [e for e in self.allgather_object(err) if e]
        if errs:
            self.rccl_error = errs[0]
            if mine and getattr(engine, "comm_size", 1) > 1:
                try:
                    engine.comm_destroy()
                except Exception:  # noqa: BLE001
                    pass
            return None
        if not mine:
            return None
        group = ShardGroup(engine, members.index(self.rank), len(members))
        group.calibration = engine.comm_calibrate() if hasattr(engine, "comm_calibrate") else None      # (collective over the members)
        return group

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
        if getattr(self, "_token_file", None):        # rank 0 of a token-file rendezvous: the file has served its job
            try:
                os.unlink(self._token_file)
            except OSError:
                pass
            self._token_file = None


def rehearsal_env(rank):
    """Environment under which RCCL accepts SEVERAL ranks on ONE device -- for rehearsing the N-rank path on a one-GPU box.
    RCCL refuses two ranks of a communicator that share a bus id on the same host ("Duplicate GPU detected"); the host
    identity it compares is NCCL_HOSTID when that is set, so ranks that name different hosts are taken for different
    machines and connect through the socket transport (loopback) -- a real communicator of N ranks: ncclCommInitRank on a
    shared unique id, ncclAllGather between processes.  Must be in os.environ before librccl is loaded (the library
    dlopens it lazily in mfgp_comm_unique_id / mfgp_comm_init).  Never used by a job with one GPU per rank."""
    return {"NCCL_HOSTID": "mfgp-rehearsal-host-%d" % int(rank), "NCCL_SOCKET_IFNAME": "lo", "NCCL_IB_DISABLE": "1",
            "NCCL_P2P_DISABLE": "1", "NCCL_SHM_DISABLE": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}


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
    return SocketComm(rank, size, addr, port, timeout=timeout, token=os.environ.get("MFGP_COMM_TOKEN"))


def _job_token(token, port, create, loopback):
    """-> the job's HMAC key.  MFGP_COMM_TOKEN (or the `token` argument) when the launcher provides one (bench.py's
    own launcher does); otherwise, on a single node, rank 0 writes 32 random bytes to a 0600 file in a 0700 per-user
    directory and the other ranks -- same user, same node -- read it.  A multi-node rendezvous needs the explicit token."""
    if token:
        return hashlib.sha256(("mfgp-comm:" + str(token)).encode("utf-8")).digest()
    if not loopback:
        raise RuntimeError("a multi-node rendezvous needs MFGP_COMM_TOKEN (a per-job secret shared by the launcher)")
    base = os.environ.get("XDG_RUNTIME_DIR") or os.path.join("/tmp", "mfgp-comm-%d" % os.getuid())
    d = os.path.join(base, "mfgp-comm") if os.environ.get("XDG_RUNTIME_DIR") else base
    path = os.path.join(d, "token-%d" % int(port))
    if create:
        os.makedirs(d, mode=0o700, exist_ok=True)
        st = os.stat(d)
        if st.st_uid != os.getuid() or (st.st_mode & 0o077):
            raise RuntimeError("%s is not a private directory of this user" % d)
        key = secrets.token_bytes(32)
        tmp = "%s.%d" % (path, os.getpid())
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o600)
        with os.fdopen(fd, "wb") as f:
            f.write(key)
        os.replace(tmp, path)
        _job_token.last_file = path
        return key
    st = os.stat(d)                          # FileNotFoundError (an OSError): rank 0 is not there yet -> the caller retries
    if st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise RuntimeError("%s is not a private directory of this user" % d)
    with open(path, "rb") as f:
        key = f.read()
    if len(key) != 32:
        raise ValueError("token file incomplete")
    return key


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
    every rank builds the full rows of ITS 128-row blocks of Ky on its GPU (serpentine block-cyclic deal, `Engine.row_block_owner`:
    every rank's blocks hold the same share of the lower triangle), the blocks are all-gathered into every rank's device matrix
    -- inside the library when the engine carries an RCCL communicator (comm.attach_engine): the lower part of each block packed
    by owner, ONE ncclAllGather over xGMI, 4 Np^2 bytes in all; through host memory and the communicator's object gather
    otherwise (multi-process tests on a one-GPU box) -- then every rank factorises (the Cholesky itself does not shard at these
    sizes).  At N = 8192 on 8 GPUs each rank still receives 235 MB to save < 0.15 ms of local K-build: this path is provided
    because the layout is what a DISTRIBUTED factorisation would start from, not because it is faster here."""
    _, npad = engine.dev_matrix()
    size, rank = comm.size, comm.rank
    if size == 1:
        engine.kbuild_rows(theta, noise, jitter, 0, npad)
        return engine.eval_prebuilt(want_grad)
    engine.kbuild_owned_rows(theta, noise, jitter, rank, size)
    if getattr(comm, "_engine", None) is engine and engine.comm_size == size:
        engine.allgather_rows()
    else:
        nblk = npad // 128
        mine = [b for b in range(nblk) if engine.row_block_owner(b, size) == rank]
        blocks = comm.allgather_object([(b, engine.rows_download(128 * b, 128 * (b + 1))) for b in mine])
        for r, owned in enumerate(blocks):
            if r != rank:
                for b, block in owned:
                    engine.rows_upload(128 * int(b), block)
    return engine.eval_prebuilt(want_grad)


def split_rows(n, rank, size):
    """contiguous, balanced [begin, end) of n rows for `rank`"""
    base, rem = divmod(n, size)
    b = rank * base + min(rank, rem)
    return b, b + base + (1 if rank < rem else 0)
