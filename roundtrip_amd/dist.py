"""Sharding of query points across ranks (one process per GPU) and the final gather.

Query points are independent (/root/reference/main_density_est.py:324: nothing crosses points until the
final np.array(...) at :342), so the path shards with no data-path collective: rank r evaluates a
contiguous block of query points and one all-gather assembles the N float64 results.  The proposal
sampler is keyed by the GLOBAL query index, so the result is bit-identical for any rank count.

The gather runs on RCCL behind the C ABI (rt_comm_* in include/rt_is.h, on the engine's own stream): `init(engine)`
makes the communicator from the launcher's environment (RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT — what
`python -m torch.distributed.run` and `bench.py --gpus N` export), exchanging RCCL's 128-byte id over a small TCP
rendezvous of its own.  Nothing here imports torch.  A caller that ALREADY runs a torch.distributed process group
(its own application, or the gloo-on-CPU tests of this repo) and has not called `init` gets its group used instead.
"""
import hashlib
import os
import socket
import struct
import sys
import time

import numpy as np

_MAGIC = b"RTID"
_MAGIC_ERR = b"RTER"     # rank 0 could not make the id: 128 bytes of its error text follow instead
_PORT_SPAN = 16          # candidate rendezvous ports: MASTER_PORT + 1 .. MASTER_PORT + _PORT_SPAN
_native = None           # (engine, rank, world_size) once init() has run
_epoch = 0               # rendezvous count of this process: keeps a second init() from meeting a stale first one
_weights = None          # per-rank shares of the query points (set_rank_weights); None: equal blocks


def shard_bounds(n, world_size, rank, weights=None):
    """Contiguous block [lo, hi) of rank `rank`.  Equal shares (the default): ceil(n / world_size) points per rank, last
    ranks may be short or empty.  `weights` (one positive number per rank, the same on every rank): block r gets
    round(n * w_r / sum(w)) of the points, cut at the rounded cumulative shares — for ranks of unequal speed."""
    n, world_size = int(n), int(world_size)
    if weights is not None:
        w = [float(v) for v in weights]
        if len(w) != world_size or min(w) <= 0.0:
            raise ValueError("need one positive weight per rank")
        tot = sum(w)
        cut = [0]
        acc = 0.0
        for v in w[:-1]:
            acc += v
            cut.append(min(n, max(cut[-1], int(n * (acc / tot) + 0.5))))
        cut.append(n)
        return cut[rank], cut[rank + 1]
    per = -(-n // world_size) if n > 0 else 0
    lo = min(n, rank * per)
    hi = min(n, lo + per)
    return lo, hi


def set_rank_weights(weights):
    """Switch from equal blocks to blocks proportional to `weights` (one per rank, identical on every rank; None: back to
    equal blocks) in sharded_map / all_gather_blocks.  The GPUs of a node do not run at one clock (the eight XCDs of ONE
    chip are 1-2 % apart); a caller that has measured its ranks' rates — bench.py --balance-ranks does, on a
    warm-up step — hands them over here.  Results cannot change: a query point's estimate depends on its GLOBAL index,
    not on the rank that evaluates it."""
    global _weights
    if weights is not None:
        weights = [float(v) for v in weights]
        if min(weights) <= 0.0:
            raise ValueError("weights must be positive")
    _weights = weights


def rank_weights():
    rank, ws = world()
    return _weights if _weights is not None and len(_weights) == ws else None


# ---- rendezvous: rank 0 hands RCCL's unique id to the other ranks ------------------------------------------------
def _token(addr, port, world_size, epoch):
    run = os.environ.get("TORCHELASTIC_RUN_ID", "") + "/" + os.environ.get("RT_COMM_RUN_ID", "")
    return hashlib.sha256(("%s:%d:%d:%d:%s" % (addr, port, world_size, epoch, run)).encode()).digest()[:16]


def _recv_exact(conn, n):
    buf = b""
    while len(buf) < n:
        chunk = conn.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("peer closed the rendezvous connection")
        buf += chunk
    return buf


def exchange_id(make_id, rank, world_size, addr, port, timeout=300.0, epoch=0):
    """Returns rank 0's `make_id()` bytes on every rank.  Rank 0 listens on the first free port of
    [port + 1, port + _PORT_SPAN] (MASTER_PORT itself belongs to the launcher's store) and serves every peer that
    presents the run's token; peers walk the same candidates until one answers with the magic — a foreign server on a
    candidate port is skipped, not trusted."""
    if world_size == 1:
        return make_id()
    tok = _token(addr, port, world_size, epoch)
    deadline = time.time() + timeout
    # RT_COMM_PORT (set for all ranks by whoever starts them; bench.py --gpus N does): exactly that port.  Otherwise the
    # first free one of MASTER_PORT + 1 .. + 16 — which can collide with a neighbouring job's MASTER_PORT when jobs on one
    # node use consecutive ones: give such jobs RT_COMM_PORT, or MASTER_PORTs further apart.
    fixed = os.environ.get("RT_COMM_PORT")
    cands = [int(fixed)] if fixed else [port + 1 + i for i in range(_PORT_SPAN)]
    if rank == 0:
        srv = None
        for p in cands:
            s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            try:
                s.bind((addr, p))
                s.listen(world_size + 8)
                srv = s
                break
            except OSError:
                s.close()
        if srv is None:
            raise RuntimeError("rendezvous: no free port in %d..%d on %s" % (cands[0], cands[-1], addr))
        failure = None
        try:
            uid = make_id()
            frame = _MAGIC + uid
        except Exception as e:          # the peers hear about it (and fail at once) instead of waiting out their timeout
            failure = e
            frame = _MAGIC_ERR + ("%s: %s" % (type(e).__name__, e)).encode("utf-8", "replace")[:128].ljust(128, b" ")
            deadline = min(deadline, time.time() + 15.0)
        served = set()
        try:
            while len(served) < world_size - 1:
                left = deadline - time.time()
                if left <= 0:
                    if failure is not None:
                        break
                    raise TimeoutError("rendezvous: %d of %d peers arrived within %.0f s"
                                       % (len(served), world_size - 1, timeout))
                srv.settimeout(min(left, 5.0))
                try:
                    conn, _ = srv.accept()
                except socket.timeout:
                    continue
                with conn:
                    conn.settimeout(10.0)
                    try:
                        hello = _recv_exact(conn, 20)
                        peer = struct.unpack("<i", hello[16:])[0]
                        if hello[:16] != tok or not 0 < peer < world_size:
                            continue                                    # not one of ours
                        conn.sendall(frame)
                        _recv_exact(conn, 1)                            # the peer has it
                        served.add(peer)
                    except (OSError, ConnectionError):
                        continue
        finally:
            srv.close()
        if failure is not None:
            raise failure
        return uid
    last = failed = None
    while time.time() < deadline:
        for p in cands:
            try:
                with socket.create_connection((addr, p), timeout=2.0) as conn:
                    conn.settimeout(10.0)
                    conn.sendall(tok + struct.pack("<i", rank))
                    head = _recv_exact(conn, 4)
                    if head not in (_MAGIC, _MAGIC_ERR):
                        continue
                    uid = _recv_exact(conn, 128)
                    conn.sendall(b"k")
                    if head == _MAGIC_ERR:
                        failed = uid.decode("utf-8", "replace").rstrip()
                        break
                    return uid
            except (OSError, ConnectionError) as e:
                last = e
        if failed is not None:
            raise RuntimeError("rendezvous: rank 0 could not make the communicator id (%s)" % failed)
        time.sleep(0.2)
    raise TimeoutError("rendezvous: rank 0 did not answer on %s:%d..%d within %.0f s (%r)"
                       % (addr, cands[0], cands[-1], timeout, last))


def init(engine, rank=None, world_size=None, addr=None, port=None, timeout=300.0):
    """Collective over all ranks: makes the RCCL communicator of `engine` (one rank per GPU).  Arguments default to
    the launcher's environment.  Afterwards world() and the gathers below use it."""
    global _native, _epoch
    rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
    world_size = int(os.environ.get("WORLD_SIZE", "1")) if world_size is None else int(world_size)
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(os.environ.get("MASTER_PORT", "29500")) if port is None else int(port)
    if not 0 <= rank < world_size:
        raise ValueError("need 0 <= rank < world_size, got %d / %d" % (rank, world_size))
    if _native is not None:
        raise RuntimeError("roundtrip_amd.dist.init was already called in this process (shutdown() first)")
    uid = exchange_id(engine.comm_unique_id, rank, world_size, addr, port, timeout, _epoch)
    _epoch += 1
    engine.comm_init(world_size, rank, uid)
    _native = (engine, rank, world_size)
    return rank, world_size


def shutdown():
    """Destroys the communicator made by init() (collective)."""
    global _native
    if _native is not None:
        eng = _native[0]
        _native = None
        eng.comm_destroy()


def _torch_group():
    """torch.distributed, if the CALLER has imported torch and initialised a process group; never imports torch."""
    td = sys.modules.get("torch.distributed")
    if td is not None and td.is_available() and td.is_initialized():
        return td
    return None


def world():
    """(rank, world_size): of the communicator made by init(), else of the caller's torch.distributed process group,
    else (0, 1)."""
    if _native is not None:
        return _native[1], _native[2]
    td = _torch_group()
    if td is not None:
        return td.get_rank(), td.get_world_size()
    return 0, 1


def active():
    """True when results must go through a gather: several ranks, or a communicator made by init() (a one-rank
    communicator still runs the RCCL call — that is how the path is exercised on a one-GPU box)."""
    return _native is not None or world()[1] > 1


def _gather_rows(rows, device=None):
    """(world, len(rows)) float64: every rank's `rows` (a flat float64 array, the same length on every rank)."""
    rows = np.ascontiguousarray(rows, dtype=np.float64).reshape(-1)
    if _native is not None:
        return _native[0].comm_all_gather(rows)           # RCCL on the engine's stream (rt_comm_all_gather_f64)
    td = _torch_group()
    torch = sys.modules["torch"]
    if td.get_backend() != "nccl":
        dev = torch.device("cpu")
    else:   # the caller's RCCL group: buffers on the engine's device, not torch's current one
        dev = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
    out = torch.empty(rows.size * td.get_world_size(), dtype=torch.float64, device=dev)
    td.all_gather_into_tensor(out, torch.from_numpy(rows).to(dev))
    return out.cpu().numpy().reshape(td.get_world_size(), rows.size)


def all_gather_blocks(local, n, device=None):
    """Assembles the full (n,) float64 result from every rank's contiguous block (see shard_bounds).

    One all-gather of ceil(n / world) float64 per rank — the only collective on the path.  Payloads are tiny
    (1M points = 8 MB in total), so this is latency- not bandwidth-bound; a single direct all-gather is right.
    With rank weights set (set_rank_weights) the blocks are unequal: every rank sends the LARGEST block's length, padded,
    and the blocks are cut out of the gathered rows by their bounds.
    `device` only matters for a caller-owned torch "nccl" group (the HIP device index of the engine)."""
    rank, ws = world()
    if not active():
        return np.asarray(local, dtype=np.float64)
    n = int(n)
    w = rank_weights()
    loc = np.asarray(local, dtype=np.float64).reshape(-1)
    if w is None:
        per = -(-n // ws)
        buf = np.zeros(per, dtype=np.float64)
        buf[: loc.size] = loc
        return _gather_rows(buf, device).reshape(-1)[:n].copy()
    bounds = [shard_bounds(n, ws, r, w) for r in range(ws)]
    per = max(1, max(hi - lo for lo, hi in bounds))
    buf = np.zeros(per, dtype=np.float64)
    buf[: loc.size] = loc
    rows = _gather_rows(buf, device)
    return np.concatenate([rows[r, : hi - lo] for r, (lo, hi) in enumerate(bounds)])


def sharded_map(points, fn, device=None):
    """Evaluates fn(block, global_index_of_block_start) on this rank's block of `points` and gathers.

    fn returns one float64 per row of its block.  With one rank it is simply fn(points, 0)."""
    points = np.asarray(points)
    n = points.shape[0]
    rank, ws = world()
    lo, hi = shard_bounds(n, ws, rank, rank_weights())
    local = fn(points[lo:hi], lo) if hi > lo else np.zeros(0, dtype=np.float64)
    return all_gather_blocks(local, n, device)


def all_gather_ranges(local, total, axis=1, device=None):
    """Every rank holds the slice shard_bounds(total, world, rank) of an array along `axis`; returns the whole array on
    every rank.  Used when ONE query point's samples are split over ranks (fewer query points than GPUs): the slices are
    ranges of a point's (max, sum) pairs, 16 bytes each — again a single small all-gather."""
    rank, ws = world()
    local = np.asarray(local, dtype=np.float64)
    if not active():
        return local
    per = -(-int(total) // ws)
    moved = np.moveaxis(local, axis, 0)                       # (mine, ...)
    rest = moved.shape[1:]
    buf = np.zeros((per,) + rest, dtype=np.float64)
    buf[: moved.shape[0]] = moved
    out = _gather_rows(buf, device).reshape((ws, per) + rest)
    parts = []
    for r in range(ws):
        lo, hi = shard_bounds(total, ws, r)
        parts.append(out[r, : hi - lo])
    return np.ascontiguousarray(np.moveaxis(np.concatenate(parts, axis=0), 0, axis))
