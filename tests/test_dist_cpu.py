"""The N > 1 path on CPU: channel sharding and the control plane, world_size 2.

Two spawned processes run the file-backed Comm (what bench.py falls back to when
RCCL is unavailable) and, beside it, torch.distributed/gloo doing the same
reductions, and must agree."""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest

from ghost_amd.dist import Comm, shard_channels


def test_shards_partition_the_channels():
    for n, w in [(1024, 8), (128, 1), (130, 8), (7, 8), (384, 5)]:
        spans = [shard_channels(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1
    assert shard_channels(1024, 3, 8) == (384, 512)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, session, port, q):
    try:
        import torch.distributed as td
        import torch
        comm = Comm(rank, world, use_rccl=False, session=session, timeout=60)
        assert comm.backend == "file"
        comm.barrier()
        mine = 10.0 + 3.0 * rank
        got = comm.allreduce_max(mine)
        # cross-check with gloo
        td.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank,
                              world_size=world)
        t = torch.tensor([mine], dtype=torch.float64)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        td.barrier()
        td.destroy_process_group()
        # each rank's shard of a synthetic job: weak scaling keeps per-rank work fixed
        a, b = shard_channels(16 * world, rank, world)
        comm.barrier()
        comm.close()                                  # signs off; rank 0 removes the session dir
        q.put((rank, got, float(t[0]), b - a))
    except Exception as e:  # pragma: no cover
        q.put((rank, "error", repr(e), 0))


def test_world_size_two_control_plane():
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    session = tempfile.mkdtemp(prefix="ghostcwt_test_")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, session, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
    assert all(r[1] != "error" for r in res), res
    assert [r[1] for r in res] == [13.0, 13.0]          # file backend
    assert [r[2] for r in res] == [13.0, 13.0]          # gloo agrees
    assert [r[3] for r in res] == [16, 16]
    assert not os.path.exists(session)                  # close() left nothing behind


def test_time_block_shards_tile_the_recording():
    from ghost_amd.dist import shard_time_blocks
    # 11 blocks of uneven length, as the planner cuts two epochs
    edges = [0, 3968, 7936, 11904, 15000, 18968, 22936, 26904, 30872, 34840, 38808, 40000]
    segs = [(a, b, 8192) for a, b in zip(edges[:-1], edges[1:])]
    for world in (1, 2, 3, 4, 8, 16):
        covered = []
        for r in range(world):
            a, b = shard_time_blocks(segs, r, world)
            assert a <= b
            if b > a:
                assert a in edges and b in edges          # whole blocks only
                covered.append((a, b))
        assert covered[0][0] == 0 and covered[-1][1] == 40000
        assert all(x[1] == y[0] for x, y in zip(covered[:-1], covered[1:]))
        if world <= 4:
            sizes = [b - a for a, b in covered]
            assert max(sizes) - min(sizes) <= 2 * 3968
    assert shard_time_blocks([], 0, 2) == (0, 0)


def test_cpulist_parser_and_pinning_is_best_effort(tmp_path):
    from ghost_amd.dist import parse_cpulist, pin_to_device_numa
    assert parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11}
    assert parse_cpulist("5") == {5} and parse_cpulist("") == set()
    before = os.sched_getaffinity(0)
    assert pin_to_device_numa(0, sysfs=str(tmp_path)) == (None, 0)     # no GPU / no sysfs entry: nothing changes
    assert os.sched_getaffinity(0) == before


def test_single_rank_comm_is_a_noop():
    c = Comm(0, 1)
    assert c.backend == "single"
    c.barrier()
    assert c.allreduce_max(2.5) == 2.5


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` run bare starts N fresh rank processes itself (before any
    HIP call), each with its own RANK / LOCAL_RANK and a shared rendezvous address; rank 0's
    line is the only thing on stdout.  --dry-run stops every rank before it touches a GPU."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3",
                          "--dry-run", str(tmp_path)], env=env, capture_output=True, timeout=120)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    lines = [l for l in res.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1 and json.loads(lines[0]) == {"dry_run": True, "n_gpus": 3}
    ranks = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(3)]
    assert [r["rank"] for r in ranks] == [0, 1, 2]
    assert [r["local_rank"] for r in ranks] == [0, 1, 2]
    assert all(r["world"] == 3 for r in ranks)
    assert len({r["master"] for r in ranks}) == 1 and ranks[0]["master"].startswith("127.0.0.1:")
    # a mismatch between the launcher's world and --gpus is refused, not folded
    env2 = dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4",
                          "--dry-run", str(tmp_path)], env=env2, capture_output=True, timeout=120)
    assert res.returncode != 0 and b"WORLD_SIZE=2 but --gpus 4" in res.stderr


def test_a_failing_rank_stops_the_launcher(tmp_path):
    """One rank dies while the others would still wait (an RCCL collective whose peer is gone
    looks like this): the launcher must end the remaining ranks and exit non-zero, promptly --
    not block reading rank 0's pipe.  Either rank 0 or another rank may be the one that dies."""
    import subprocess
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dead in ("0", "2"):
        env = {k: v for k, v in os.environ.items()
               if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
        env["GHOSTCWT_BENCH_FAIL_RANK"] = dead
        t0 = time.time()
        res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3",
                              "--dry-run", str(tmp_path)], env=env, capture_output=True, timeout=100)
        assert res.returncode != 0
        assert time.time() - t0 < 60, "the launcher waited for ranks that would never finish"
        assert ("rank %s exited with 3" % dead).encode() in res.stderr
        assert res.stdout.strip() == b""


def test_rccl_failure_after_setup(tmp_path, monkeypatch):
    """An RCCL call that fails when it is ENQUEUED (GCWT_ERR_COMM: every rank sees it) must not
    take the run down: the operation is served by the file backend, the communicator is aborted
    -- never destroyed, which would wait for work in flight -- and the reason is kept (bench.py
    prints it as config.rccl_error).  A collective that was enqueued and did not complete
    (GCWT_ERR_COMM_INCOMPLETE) means a peer is gone: the communicator is aborted and the error is
    raised, so that the rank exits non-zero and the launcher stops the others.  Any other code is the
    rank's own failure (a HIP error while the plan uploads, say): re-raised as it is, not reported as
    a lost peer; and the short time-out of the fall-back applies to its first operation only."""
    import threading
    import ghost_amd._lib as _lib
    from ghost_amd.dist import Comm

    class Broken:                       # stands in for libghostcwt's RCCL entry points
        destroyed = aborted = 0
        code = _lib.ERR_COMM

        def gcwt_comm_barrier(self, h):
            return Broken.code

        def gcwt_comm_allreduce_max(self, h, v):
            return Broken.code

        def gcwt_comm_destroy(self, h):
            Broken.destroyed += 1

        def gcwt_comm_abort(self, h):
            Broken.aborted += 1

        def gcwt_last_error(self):
            return b"ncclAllReduce: unhandled system error"

    monkeypatch.setattr(_lib, "lib", Broken())
    out = {}

    def rank(r):
        c = Comm(r, 2, use_rccl=False, session=str(tmp_path), timeout=20.0)
        c._handle, c.backend = object(), "rccl"          # as if set-up had succeeded
        c.barrier()
        out[r] = (c.allreduce_max(10.0 + r), c.backend, c.rccl_error, c._handle, c.allgather(float(r)))

    ts = [threading.Thread(target=rank, args=(r,)) for r in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(30)
    for r in range(2):
        value, backend, err, handle, gathered = out[r]
        assert value == 11.0 and backend == "file" and handle is None and gathered == [0.0, 1.0]
        assert "barrier" in err and "unhandled system error" in err
    assert Broken.aborted == 2 and Broken.destroyed == 0
    # a collective that did not complete: no fall-back, the rank gives up
    Broken.code = _lib.ERR_COMM_INCOMPLETE
    lone = Comm.__new__(Comm)
    lone.rank, lone.world, lone.timeout, lone.dir, lone._seq = 0, 2, 500.0, str(tmp_path), 0
    lone._handle, lone.backend, lone.rccl_error = object(), "rccl", None
    with pytest.raises(RuntimeError, match="peer rank is gone"):
        lone.barrier()
    assert lone._handle is None and Broken.aborted == 3
    assert lone.timeout == 500.0                     # the ranks' own time-out is never shortened for good
    # a local failure is reported as what it is
    Broken.code = _lib.ERR_HIP
    lone._handle, lone.backend = object(), "rccl"
    with pytest.raises(Exception) as ei:
        lone.barrier()
    assert "peer rank" not in str(ei.value) and getattr(ei.value, "code", None) == _lib.ERR_HIP
    assert Broken.aborted == 4
