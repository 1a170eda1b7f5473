"""N>1 control plane on CPU: world_size 2 over gloo (the GPU job uses the same code over RCCL).
Each rank writes its result to its own file: two ranks printing to one pipe can interleave lines."""
import json
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import json, os, sys, time
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from libav_amd import shard
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    out = {"rank": rank, "backend_world_size": world}
    # --- static deal -------------------------------------------------------------------------
    table = shard.make_stream_table(7, 0x264) if rank == 0 else None
    table = shard.broadcast_stream_table(table, 7, "cpu")
    mine = shard.my_streams(table, rank, world)
    dist.barrier()
    elapsed, units = shard.reduce_counters(1.0 + rank, len(mine) * 8160, "cpu")
    out.update(mine=mine, elapsed=elapsed, units=units)
    # --- work queue: streams of uneven length, ranks of uneven speed ----------------------------
    gops = [5, 1, 9, 3, 3, 7, 2]
    items = shard.make_work_items(gops)
    q = shard.WorkQueue(len(items), batch=2)
    done = []
    while True:
        r = q.next()
        if r is None:
            break
        for i in r:
            done.append(list(items[i]))
            time.sleep(0.02 if rank == 0 else 0.002)      # rank 0 is ten times slower per item
    dist.barrier()
    _, total = shard.reduce_counters(0.0, len(done), "cpu")
    out.update(done=done, total=total, per_rank=shard.gather_counts(len(done), "cpu"), n_items=len(items))
    # a second queue in the same job starts from zero again
    q2 = shard.WorkQueue(3, batch=1)
    got = []
    while True:
        r = q2.next()
        if r is None:
            break
        got += list(r)
    out["second"] = got
    json.dump(out, open(os.path.join(%r, "rank%%d.json" %% rank), "w"))
    dist.destroy_process_group()
''')


def _run(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % (ROOT, str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    with socket.socket() as sock:                            # a free port: a fixed one collides with a lingering run
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    return [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(2)]


def test_two_ranks_static_deal_and_work_queue(tmp_path):
    rows = _run(tmp_path)
    assert all(r["backend_world_size"] == 2 for r in rows)
    # static deal: every stream exactly once, seeds follow the table, counters reduce to (max, sum)
    streams = sorted(s for r in rows for s, _ in r["mine"])
    assert streams == list(range(7))
    assert all(seed == 0x264 + s for r in rows for s, seed in r["mine"])
    assert all(r["elapsed"] == 2.0 and r["units"] == 7 * 8160 for r in rows)
    # work queue: every (stream, gop) item done exactly once, by whichever rank had room
    gops = [5, 1, 9, 3, 3, 7, 2]
    want = sorted([s, g] for s, n in enumerate(gops) for g in range(n))
    assert sorted(i for r in rows for i in r["done"]) == want
    assert all(r["total"] == len(want) and r["n_items"] == len(want) for r in rows)
    assert rows[0]["per_rank"] == rows[1]["per_rank"] == [len(rows[0]["done"]), len(rows[1]["done"])]
    assert len(rows[1]["done"]) > len(rows[0]["done"])       # the faster rank took more of the queue
    # per stream, GOPs leave the queue in order
    for r in rows:
        for s in range(len(gops)):
            g = [i[1] for i in r["done"] if i[0] == s]
            assert g == sorted(g)
    assert sorted(rows[0]["second"] + rows[1]["second"]) == [0, 1, 2]


def test_work_queue_single_process():
    sys.path.insert(0, ROOT)
    from libav_amd import shard
    items = shard.make_work_items([2, 0, 3])
    assert items == [(0, 0), (2, 0), (0, 1), (2, 1), (2, 2)]
    q = shard.WorkQueue(len(items), batch=2)
    got = []
    while True:
        r = q.next()
        if r is None:
            break
        got += list(r)
    assert got == list(range(5))
