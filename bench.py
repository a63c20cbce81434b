#!/usr/bin/env python3
"""bench.py -- decode tokens/s of the MI355X-native engine on BASELINE.json's headline configuration.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one greedy decode token (one pass of the hot path over the resident weights, batch 1).  The
workload at N=1 is BASELINE.json configs[1]: Llama-3.1-8B-shaped Q8_0, weights resident in HBM, greedy decode
after a 16-token prompt (seeded synthetic weights: no checkpoint exists offline).  W untimed warm-up tokens, then
exactly K tokens timed between (barrier + device synchronize) pairs; rank 0 prints ONE JSON line.  With N > 1
every rank runs an independent whole-model replica on its own GPU (the path has no exchange step),
`value` = N*K / max-over-ranks time, scaling = weak.

Besides the contract fields the line carries
  roofline     -- HBM roofline of the dominant kernel (the dequant-fused GEMV): algorithmic bytes per launch /
                  average launch duration measured with HIP event pairs on the compute stream
  cpu_baseline -- the oracle (CPU restatement of the reference kernels) timed on this host on a bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling 6290 GB/s
REF_3090_TOK_S = 48.9          # reference, RTX 3090, same metric and model config


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--model", default="8b", choices=["tiny", "small", "8b", "70b"])
    ap.add_argument("--mix", default="Q8_0")
    ap.add_argument("--ctx", type=int, default=4096)
    ap.add_argument("--prompt-len", type=int, default=16)
    ap.add_argument("--no-fuse", action="store_true", help="the reference's 15-launch/layer sequence")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline sample")
    ap.add_argument("--no-pmc-note", action="store_true")
    return ap.parse_args()


def cpu_baseline(args, spec_full, prompt):
    """Oracle decode on a bounded sample: SAMPLE_LAYERS layers of the same shape + the full LM head, timed per
    section and scaled to the full depth.  The oracle is the CPU restatement of the reference's kernels driven in
    the reference's launch order (oracle/oracle.py); OpenMP over output rows."""
    import numpy as np
    from ntransformer_amd import engine as E
    from oracle import oracle as O
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    threads = O.pick_threads()      # what this host really schedules, not what it advertises
    sample_layers = 2 if args.model in ("8b", "70b") else E.PRESETS[args.model]["layers"]
    spec = E.synth_spec(args.model, args.mix, layers=sample_layers)
    path = "/dev/shm/_bench_cpu_sample.gguf" if os.path.isdir("/dev/shm") else "/tmp/_bench_cpu_sample.gguf"
    E.synth_write_gguf(path, spec)
    try:
        m = O.OracleModel(path, max_context=256)
        full_layers = E.PRESETS[args.model]["layers"]
        m.forward(prompt[:4], 0)                               # warm page cache / threads
        # time one decode step split into [layers] and [final norm + LM head] by differencing two model depths
        t_steps, pos, tok = [], 4, int(prompt[4])
        t_budget = time.perf_counter() + args.cpu_seconds
        while len(t_steps) < 2 or (time.perf_counter() < t_budget and len(t_steps) < 64):
            t0 = time.perf_counter()
            lg = m.forward([tok], pos)
            t_steps.append(time.perf_counter() - t0)
            tok, pos = int(np.argmax(lg)), pos + 1
        m1 = O.OracleModel(path, max_context=256, n_layers=1)
        m1.forward(prompt[:4], 0)
        t1 = []
        for i in range(max(2, min(8, len(t_steps)))):
            t0 = time.perf_counter()
            m1.forward([tok], 4 + i)
            t1.append(time.perf_counter() - t0)
        t_full, t_one = float(np.median(t_steps)), float(np.median(t1))
        t_layer = max((t_full - t_one) / max(sample_layers - 1, 1), 1e-9)
        t_head = max(t_one - t_layer, 0.0)
        tok_s = 1.0 / (t_head + full_layers * t_layer)
        cores = threads
        return {"value": round(tok_s, 4), "unit": "tokens/s", "cores": cores, "kind": "port",
                "sample": "%d of %d layers + full LM head of the same %s %s model, %d decode steps on the CPU oracle "
                          "(%.3f s/layer, %.3f s head+embed); layers scaled to full depth"
                          % (sample_layers, full_layers, args.model, args.mix, len(t_steps), t_layer, t_head),
                "host": _cpu_model()}
    finally:
        try:
            os.remove(path)
        except OSError:
            pass


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and world != args.gpus:
        print("bench.py: WORLD_SIZE=%d but --gpus %d" % (world, args.gpus), file=sys.stderr)
    os.environ["NTK_DEVICE"] = str(local_rank)

    from ntransformer_amd import replica
    # torch only as rendezvous / barrier / max-reduce plumbing: replicas exchange no data, so 8 bytes per run go over gloo
    # (NT_DIST_BACKEND=nccl for RCCL)
    dist, backend = replica.init_distributed("gloo", local_rank) if world > 1 else (None, None)

    import numpy as np
    from ntransformer_amd import _lib
    from ntransformer_amd import engine as E
    L = _lib.lib()
    ndev = L.ntk_device_count()
    dev = local_rank
    if ndev > 0 and local_rank >= ndev:   # more ranks than GPUs (only ever a test of the multi-rank plumbing): share devices
        dev = local_rank % ndev
        print("bench.py: rank %d shares GPU %d (%d visible)" % (rank, dev, ndev), file=sys.stderr)
    os.environ["NTK_DEVICE"] = str(dev)
    _lib.check(L.ntk_device_init(dev), "ntk_device_init(%d)" % dev)

    spec = E.synth_spec(args.model, args.mix)
    eng = E.Engine()
    eng.set_option("fused", not args.no_fuse)
    eng.set_option("graph", not args.no_graph)
    t_load = time.perf_counter()
    eng.load_synthetic(spec, args.ctx)
    t_load = time.perf_counter() - t_load

    rng = np.random.Generator(np.random.Philox(key=[20260925, 99]))
    prompt = [spec.bos] + [int(t) for t in rng.integers(0, spec.vocab, args.prompt_len - 1)]
    # prefill + first token (untimed), exactly Engine::generate's first half
    first = eng.generate_tokens(prompt, 1, temperature=0.0, repeat_penalty=1.0, stop_at_eos=False)
    tok, pos = first[0], len(prompt)
    warm = eng.decode_greedy_steps(tok, pos, args.warmup) if args.warmup > 0 else []
    if warm:
        tok = warm[-1]
    pos += args.warmup

    elapsed, _, out = replica.timed_steps(lambda k: eng.decode_greedy_steps(tok, pos, k), args.steps,
                                          lambda: _lib.check(L.ntk_device_synchronize(), "device sync"), dist, backend)
    pos_end = pos + args.steps

    if rank == 0:
        tok_s = world * args.steps / elapsed
        b_tok = eng.bytes_per_token(pos + args.steps // 2)
        # ---- roofline of the dominant kernel, measured live with HIP events on the compute stream over a few eagerly
        # launched tokens.  `coarse`: one event per run of GEMV launches (o, gate|up, down, next qkv between two attention
        # launches), so the events' own cost (~2 us each) is paid once per 4 launches; `fine` (an event pair per launch,
        # reads ~2 us high per launch) is reported beside it.  rocprofv3's kernel trace of this command is in profiles/.
        def prof(coarse, n_prof=4):
            ms, calls = [0.0] * 4, [0] * 4
            for i in range(n_prof):
                m_, c_ = eng.profile_token(out[-1] if out else tok, min(pos_end + i, args.ctx - 1), coarse)
                ms = [a + b for a, b in zip(ms, m_)]
                calls = [a + b for a, b in zip(calls, c_)]
            return [m / n_prof for m in ms], [c / n_prof for c in calls]
        ms, calls = prof(True)
        ms_fine, calls_fine = prof(False)
        gemv_bytes_tok = _gemv_bytes_per_token(eng, spec, args.mix)
        launches_tok = calls[0]
        avg_launch_ms = ms[0] / max(calls[0], 1)
        achieved = (gemv_bytes_tok / max(launches_tok, 1)) / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
        traffic, traffic_src = _pmc_traffic(args)
        line = {
            "metric": "decode tokens/sec (Llama-3.1-8B Q8_0 class, resident weights, greedy, batch 1)" if (args.model, args.mix) == ("8b", "Q8_0")
                      else "decode tokens/sec (%s %s, resident weights, greedy, batch 1)" % (args.model, args.mix),
            "value": round(tok_s, 3), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": round(tok_s / world / REF_3090_TOK_S, 3) if (args.model, args.mix) == ("8b", "Q8_0") else None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Llama-3.1-%s-shaped %s GGUF tensors (seed 20260925), resident in HBM, %d-token prompt, greedy decode"
                                   % (args.model.upper(), args.mix, args.prompt_len),
                       "ctx": args.ctx, "decode_positions": [pos, pos_end], "replicas": world,
                       "path": "1:1 launchers (15/layer)" if args.no_fuse else "fused (5 launches/layer)",
                       "hipgraph": not args.no_graph and not args.no_fuse,
                       "algorithmic_bytes_per_token": b_tok, "load_seconds": round(t_load, 2)},
            "hbm_fraction_of_8TBs_end_to_end": round(b_tok * tok_s / world / (HBM_PEAK_GBS * 1e9), 4),
            "roofline": {"bound": "hbm", "kernel": "gemv_quant_kernel<%s> (all projection launches of a token pooled)" % args.mix,
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "bytes_per_launch": int(gemv_bytes_tok / max(launches_tok, 1)), "launches_per_token": launches_tok,
                         "avg_launch_us": round(avg_launch_ms * 1e3, 2),
                         "avg_launch_us_event_pair_per_launch": round(ms_fine[0] / max(calls_fine[0], 1) * 1e3, 2),
                         "timed_runs_per_token": calls[3],
                         "token_ms_by_class_eager": {"gemv": round(ms[0], 4), "attention": round(ms[1], 4), "other": round(ms[2], 4)}},
        }
        if not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(args, spec, prompt)
            except Exception as e:   # the GPU number must survive a CPU-side problem
                line["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(line), flush=True)
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def _pmc_traffic(args):
    """HBM bytes per GEMV launch from the committed rocprofv3 PMC passes (FETCH_SIZE + WRITE_SIZE, separate runs of this
    command, decode window, gfx950 x2 correction on FETCH_SIZE: tools/pmc_summary.py -> profiles/pmc_traffic.json).
    PMC collection needs rocprofv3 around the process, so the number is read back, not measured in this run."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic.json")
    key = "%s_%s" % (args.model, args.mix.lower())
    try:
        g = json.load(open(path))[key]["ntk::gemv_quant_kernel"]
        return int(g["fetch_bytes_per_launch"] + g["write_bytes_per_launch_raw"]), "profiles/pmc_traffic.json[%s]" % key
    except Exception:
        return None, None


def _gemv_bytes_per_token(eng, spec, mix):
    """Bytes of the matrices the GEMV launches stream per token = weights minus the embedding table (gathered, 1 row)."""
    from ntransformer_amd import gguf as G
    shape = G.LlamaShape("b", spec.hidden, spec.inter, spec.layers, spec.heads, spec.kv_heads, spec.vocab)
    types = G.tensor_types(shape, mix if mix == "Q4_K_M" else {"Q4_K": "Q4_K", "Q5_K": "Q5_K"}.get(mix, mix))
    hd = spec.hidden // spec.heads
    dims = {"attn_q": (spec.hidden, spec.heads * hd), "attn_k": (spec.hidden, spec.kv_heads * hd), "attn_v": (spec.hidden, spec.kv_heads * hd),
            "attn_output": (spec.heads * hd, spec.hidden), "ffn_gate": (spec.hidden, spec.inter), "ffn_up": (spec.hidden, spec.inter),
            "ffn_down": (spec.inter, spec.hidden)}
    total = G.row_bytes(types["output.weight"], spec.hidden) * spec.vocab
    for name, t in types.items():
        if name.startswith("blk."):
            i, o = dims[name.split(".")[2]]
            total += G.row_bytes(t, i) * o
    return total


if __name__ == "__main__":
    main()
