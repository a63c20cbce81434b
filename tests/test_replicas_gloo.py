"""The N > 1 path of bench.py on CPU: two processes over gloo run the replica timing protocol
(ntransformer_amd/replica.py) with a stand-in step function.  Checks: both ranks leave the timed region together,
the reported time is the MAX over ranks, the whole-job rate is N*K/max, and request sharding is a partition."""
import json
import os
import socket
import subprocess
import sys
import textwrap

from conftest import ROOT
from ntransformer_amd import replica


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    from ntransformer_amd import replica
    rank, local, world = replica.env_ranks()
    dist, backend = replica.init_distributed("gloo", local)
    assert backend == "gloo"
    per_step = 0.01 * (1 + 3 * rank)            # rank 1 is 4x slower: the job runs at its pace
    def run(k):
        time.sleep(per_step * k)
        return k
    elapsed, rate, res = replica.timed_steps(run, 10, lambda: None, dist, backend)
    print(json.dumps({"rank": rank, "elapsed": elapsed, "rate": rate, "mine": replica.shard_requests(7, rank, world)}), flush=True)
    dist.barrier(); dist.destroy_process_group()
""") % ROOT


def test_two_replicas_over_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=180)
        assert p.returncode == 0, e[-2000:]
        outs.append(json.loads(o.strip().splitlines()[-1]))
    a, b = sorted(outs, key=lambda d: d["rank"])
    assert abs(a["elapsed"] - b["elapsed"]) < 1e-6          # all-reduced MAX: identical on every rank
    assert 0.38 <= a["elapsed"] < 1.5                        # the slow replica's 10 * 0.04 s, not the fast one's 0.1 s
    assert abs(a["rate"] - 2 * 10 / a["elapsed"]) < 1e-6     # whole-job steps/s
    assert sorted(a["mine"] + b["mine"]) == list(range(7)) and not set(a["mine"]) & set(b["mine"])


def test_single_process_path():
    elapsed, rate, res = replica.timed_steps(lambda k: k * 2, 5, lambda: None, None)
    assert res == 10 and rate == 5 / elapsed
    assert replica.shard_requests(5, 0, 1) == [0, 1, 2, 3, 4]
