"""N>1 control plane on CPU: world_size 2 over gloo (the GPU job uses the same code over RCCL)."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import json, os, sys
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    from libav_amd import shard
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    table = shard.make_stream_table(7, 0x264) if rank == 0 else None
    table = shard.broadcast_stream_table(table, 7, "cpu")
    mine = shard.my_streams(table, rank, world)
    dist.barrier()
    elapsed, units = shard.reduce_counters(1.0 + rank, len(mine) * 8160, "cpu")
    print(json.dumps({"rank": rank, "mine": mine, "elapsed": elapsed, "units": units}), flush=True)
    dist.destroy_process_group()
''') % ROOT


def test_two_ranks_share_streams_without_overlap(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    import socket
    with socket.socket() as sock:                            # a free port: a fixed one collides with a lingering run
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    rows = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(rows) == 2
    streams = sorted(s for r in rows for s, _ in r["mine"])
    assert streams == list(range(7))                       # every stream exactly once
    assert all(seed == 0x264 + s for r in rows for s, seed in r["mine"])
    assert all(r["elapsed"] == 2.0 and r["units"] == 7 * 8160 for r in rows)   # max time, summed units
