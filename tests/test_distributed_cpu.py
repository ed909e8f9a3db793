"""The N > 1 path of bench.py on CPU: two processes, gloo backend — rendezvous, per-rank seeds, MAX of the wall time and SUM of
the work counters (the only things ranks exchange: there is no collective on the data path)."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import json, os, sys
    sys.path.insert(0, %r)
    from alphagomoku_amd import distributed
    rank, local_rank, world = distributed.env_ranks()
    dist = distributed.init(backend="gloo")
    dist.barrier()
    elapsed, counters = distributed.combine(dist, 1.0 + rank, [100 * (rank + 1), 7, rank])
    dist.barrier()
    print(json.dumps(dict(rank=rank, world=world, elapsed=elapsed, counters=counters, seed=distributed.rank_seed_base(rank))))
    dist.destroy_process_group()
''')


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_ranks_gloo(tmp_path):
    import json
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        out, err = p.communicate(timeout=180)
        assert p.returncode == 0, err
        outs.append(json.loads(out.strip().splitlines()[-1]))
    for o in outs:
        assert o["world"] == 2
        assert o["elapsed"] == 2.0                       # MAX over ranks
        assert o["counters"] == [300.0, 14.0, 1.0]      # SUM over ranks
    assert outs[0]["seed"] != outs[1]["seed"]


def test_single_process_is_identity():
    from alphagomoku_amd import distributed
    assert distributed.combine(None, 1.5, [3, 4]) == (1.5, [3.0, 4.0])
