"""Tensor-parallel decoding on ONE GPU: the ranks share the device (the GPU boxes of this project have a single MI355X), which
exercises everything except the xGMI link itself -- the sliced weights, the local heads and KV cache, the exchange kernel with
its flags, slots and epochs, hipGraph replay, the hipIpc mapping across processes.  Every rank's logits are compared with the
ORACLE (oracle.OracleModel: the CPU restatement of the reference on the same file and token stream, north-star tolerance 1e-3 --
these models are 1-4 layers deep, far below the depth where half-rounding flips dominate, tests/test_parity_depth.py) and with the
unsliced engine (same kernels, other summation order), and must be bit-identical on all ranks."""
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from ntransformer_amd import engine as E
from ntransformer_amd import gguf as G
from oracle import oracle as O
from test_oracle_golden import golden_model

pytestmark = pytest.mark.gpu

ORACLE_TOL = 1e-3   # against the oracle: the north-star tolerance
TOL = 5e-4   # against the unsliced engine (same kernels; partial sums added in a different order, other launch geometries); the logits bar is 1e-3


def _run_rank(eng, prompt, fed, graph, out, key):
    try:
        lg = [eng.forward(prompt, 0)]
        pos = len(prompt)
        for t in fed:
            lg.append(eng.decode_fused(int(t), pos, graph))
            pos += 1
        toks = eng.decode_greedy_steps(int(fed[-1]), pos, 8)
        out[key] = (np.stack(lg), toks, eng.tp_error())
    except Exception as e:   # noqa: BLE001 -- reported by the main thread
        out[key] = e


def _oracle_logits(path, ctx, prompt, fed):
    """the oracle on the same token stream (teacher-forced: the ranks are fed `fed`, not their own arg-max)"""
    m = O.OracleModel(path, ctx)
    lg = [m.forward(prompt, 0)]
    pos = len(prompt)
    for t in fed:
        lg.append(m.forward([int(t)], pos))
        pos += 1
    return np.stack(lg)


def _single(path, ctx, prompt, fed, graph):
    eng = E.Engine()
    eng.load(path, ctx)
    out = {}
    _run_rank(eng, prompt, fed, graph, out, 0)
    eng.close()
    assert not isinstance(out[0], Exception), out[0]
    return out[0]


@pytest.mark.parametrize("name,shape,mix,world", [("tiny_q8_0", G.TINY, "Q8_0", 2), ("small_q8_0", G.SMALL, "Q8_0", 2),
                                                  ("small_q4_k_m", G.SMALL, "Q4_K_M", 2), ("small_q6_k", G.SMALL, "Q6_K", 2)])
@pytest.mark.parametrize("graph", [False, True])
def test_ranks_sharing_a_process_match_the_unsliced_engine(name, shape, mix, world, graph, tmp_path):
    path, z = golden_model(name, shape, mix, tmp_path)
    ctx = int(z["ctx"])
    r = np.random.Generator(np.random.Philox(key=[20260925, 7]))
    prompt = [int(z["prompt"][0])] + [int(t) for t in r.integers(0, 256, 20)]   # > 16 tokens: the FP16 prompt GEMM under slices
    fed = [int(t) for t in r.integers(0, 256, 5)]
    ref_logits, ref_toks, _ = _single(path, ctx, prompt, fed, graph)
    want = _oracle_logits(path, ctx, prompt, fed)

    engines = []
    for rank in range(world):
        eng = E.Engine()
        eng.tp_configure(rank, world)
        eng.load(path, ctx)
        engines.append(eng)
    raws = [eng.tp_export()[1] for eng in engines]
    for eng in engines:
        eng.tp_connect(raws=raws)
    out = {}
    threads = [threading.Thread(target=_run_rank, args=(engines[k], prompt, fed, graph, out, k)) for k in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert all(not t.is_alive() for t in threads), "a rank did not finish"
    for eng in engines:
        eng.close()
    for k in range(world):
        assert not isinstance(out[k], Exception), out[k]
        lg, toks, err = out[k]
        assert err == 0, "rank %d: a wait for a peer gave up (%d)" % (k, err)
        assert np.isfinite(lg).all()
        assert np.abs(lg - want).max() <= ORACLE_TOL, ("vs oracle", k, np.abs(lg - want).max())
        assert np.abs(lg - ref_logits).max() <= TOL, (k, np.abs(lg - ref_logits).max())
    for k in range(1, world):   # the ranks add the same numbers in the same order
        assert np.array_equal(out[k][0], out[0][0])
        assert out[k][1] == out[0][1]
    assert out[0][1] == ref_toks or np.abs(out[0][0] - ref_logits).max() > 0   # greedy stream: equal unless a near-tie flips


@pytest.mark.parametrize("world", [4])
def test_four_way_slices_of_a_70b_width_layer(world):
    """Llama-3.1-70B width (H 8192, 64 / 8 heads, FFN 28672), one layer, Q4_K_M, built by the seeded generator, 4 ranks against
    the unsliced engine.  (8 ranks as 8 threads on ONE GPU oversubscribe its hardware queues: the waiting kernels of some ranks
    are time-sliced against the producing kernels of others and the bounded waits give up -- an artefact of the emulation, seen
    once and not kept as a test; on 8 GPUs every rank has its own queues.)"""
    spec = E.synth_spec("70b", "Q4_K_M", layers=1)
    r = np.random.Generator(np.random.Philox(key=[20260925, 9]))
    prompt = [128000] + [int(t) for t in r.integers(0, 128000, 17)]
    fed = [int(t) for t in r.integers(0, 128000, 3)]

    def single():
        eng = E.Engine()
        eng.load_synthetic(spec, 128)
        out = {}
        _run_rank(eng, prompt, fed, True, out, 0)
        eng.close()
        assert not isinstance(out[0], Exception), out[0]
        return out[0]
    ref_logits, _, _ = single()
    path = os.path.join("/dev/shm" if os.access("/dev/shm", os.W_OK) else "/tmp", "_tp_70b_l1.gguf")
    E.synth_write_gguf(path, spec)       # the same seeded tensors as load_synthetic (test_synthetic_loader_equals_file_loader)
    try:
        want = _oracle_logits(path, 128, prompt, fed)
    finally:
        os.remove(path)
    engines = []
    for rank in range(world):
        eng = E.Engine()
        eng.tp_configure(rank, world)
        eng.load_synthetic(spec, 128)
        engines.append(eng)
    raws = [eng.tp_export()[1] for eng in engines]
    for eng in engines:
        eng.tp_connect(raws=raws)
    out = {}
    threads = [threading.Thread(target=_run_rank, args=(engines[k], prompt, fed, True, out, k)) for k in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=180)
    assert all(not t.is_alive() for t in threads), "a rank did not finish"
    for eng in engines:
        eng.close()
    for k in range(world):
        assert not isinstance(out[k], Exception), out[k]
        assert out[k][2] == 0
        assert np.abs(out[k][0] - want).max() <= ORACLE_TOL, ("vs oracle", k, np.abs(out[k][0] - want).max())
        assert np.abs(out[k][0] - ref_logits).max() <= TOL, (k, np.abs(out[k][0] - ref_logits).max())
        assert np.array_equal(out[k][0], out[0][0])


def test_a_model_that_does_not_divide_is_refused(tmp_path):
    path, z = golden_model("tiny_q4_k_m", G.TINY, "Q4_K_M", tmp_path)   # Wo has 256 columns: half a Q4_K super-block per rank
    eng = E.Engine()
    eng.tp_configure(0, 2)
    with pytest.raises(Exception):
        eng.load(path, int(z["ctx"]))
    eng.close()
    eng = E.Engine()
    eng.tp_configure(0, 4)                                               # 2 KV heads over 4 ranks
    path8, z8 = golden_model("small_q8_0", G.SMALL, "Q8_0", tmp_path)
    with pytest.raises(Exception):
        eng.load(path8, int(z8["ctx"]))
    eng.close()


RANK_SCRIPT = r"""
import sys, json, numpy as np
sys.path.insert(0, sys.argv[1])
from ntransformer_amd import engine as E, tp
rank, world, path, ctx, d = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5]), sys.argv[6]
job = json.load(open(d + "/job.json"))
eng = E.Engine()
eng.tp_configure(rank, world)
eng.load(path, ctx)
tp.connect_over_files(eng, rank, world, d, timeout_s=100, run_id=job["run_id"])
lg = [eng.forward(job["prompt"], 0)]
pos = len(job["prompt"])
for t in job["fed"]:
    lg.append(eng.decode_fused(int(t), pos, True))
    pos += 1
np.save(d + "/logits_%d.npy" % rank, np.stack(lg))
json.dump({"tp_error": eng.tp_error()}, open(d + "/done_%d.json" % rank, "w"))
eng.close()
"""


def test_two_processes_over_hipipc_match_the_unsliced_engine(tmp_path):
    """the deployment form: one process per rank, communication buffers mapped with hipIpc handles (exchanged through files)"""
    path, z = golden_model("small_q8_0", G.SMALL, "Q8_0", tmp_path)
    ctx = int(z["ctx"])
    r = np.random.Generator(np.random.Philox(key=[20260925, 8]))
    prompt = [int(z["prompt"][0])] + [int(t) for t in r.integers(0, 256, 20)]
    fed = [int(t) for t in r.integers(0, 256, 4)]
    ref_logits, _, _ = _single(path, ctx, prompt, fed, True)
    want = _oracle_logits(path, ctx, prompt, fed)
    d = str(tmp_path)
    json.dump({"prompt": prompt, "fed": fed, "run_id": "t2"}, open(os.path.join(d, "job.json"), "w"))
    for k in range(2):   # handle files of an earlier run in the same directory must not be picked up (run_id namespaces them)
        open(os.path.join(d, "tp_handle_%d" % k), "wb").write(b"\0" * 64)
    script = os.path.join(d, "rank.py")
    open(script, "w").write(RANK_SCRIPT)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, script, ROOT, str(k), "2", path, str(ctx), d], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for k in range(2)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        outs.append(o.decode(errors="replace"))
    for k, p in enumerate(procs):
        assert p.returncode == 0, outs[k][-2000:]
    lgs = [np.load(os.path.join(d, "logits_%d.npy" % k)) for k in range(2)]
    for k in range(2):
        assert json.load(open(os.path.join(d, "done_%d.json" % k)))["tp_error"] == 0
        assert np.abs(lgs[k] - want).max() <= ORACLE_TOL, ("vs oracle", np.abs(lgs[k] - want).max())
        assert np.abs(lgs[k] - ref_logits).max() <= TOL, np.abs(lgs[k] - ref_logits).max()
    assert np.array_equal(lgs[0], lgs[1])


def test_eight_processes_over_hipipc_match_the_oracle(tmp_path):
    """The 8-rank deployment form on ONE device: eight PROCESSES (each with its own HIP context and hardware queues -- the 8-thread
    emulation oversubscribed one process's queues, see above), handles over files, a model with 8 KV heads so that every rank owns a
    whole KV head (hidden 1024, 8 / 8 heads, FFN 2048, 2 layers, Q8_0).  Every rank against the oracle; ranks bit-identical.  A rank
    whose bounded wait for a peer gives up reports it (tp_error) -- the run then fails here instead of hanging the device."""
    spec = E.SynthSpec(1024, 2048, 2, 8, 8, 2048, 512, 1e-5, 500000.0, 256, 257, b"Q8_0", 20260925)
    path = str(tmp_path / "tp8.gguf")
    E.synth_write_gguf(path, spec)
    r = np.random.Generator(np.random.Philox(key=[20260925, 88]))
    prompt = [256] + [int(t) for t in r.integers(0, 2048, 20)]
    fed = [int(t) for t in r.integers(0, 2048, 3)]
    want = _oracle_logits(path, 256, prompt, fed)
    d = str(tmp_path)
    json.dump({"prompt": prompt, "fed": fed, "run_id": "t8"}, open(os.path.join(d, "job.json"), "w"))
    script = os.path.join(d, "rank.py")
    open(script, "w").write(RANK_SCRIPT)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, script, ROOT, str(k), "8", path, "256", d], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for k in range(8)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        outs.append(o.decode(errors="replace"))
    for k, p in enumerate(procs):
        assert p.returncode == 0, (k, outs[k][-2000:])
    lgs = [np.load(os.path.join(d, "logits_%d.npy" % k)) for k in range(8)]
    for k in range(8):
        assert json.load(open(os.path.join(d, "done_%d.json" % k)))["tp_error"] == 0, "rank %d gave up waiting for a peer" % k
        assert np.isfinite(lgs[k]).all()
        assert np.abs(lgs[k] - want).max() <= ORACLE_TOL, ("vs oracle", k, np.abs(lgs[k] - want).max())
        assert np.array_equal(lgs[k], lgs[0])
