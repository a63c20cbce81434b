"""oracle/oracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes view of oracle/liboracle.so (the CPU restatement of the reference's CUDA kernels,
oracle/ref_launchers_cpu.cpp) plus `OracleModel`, a restatement of the reference's host orchestration
of those kernels for the resident path:

  Transformer::forward      reference src/model/transformer.cpp:604-669
  Attention::forward        reference src/model/attention.cpp:120-211
  FFN::forward              reference src/model/ffn.cpp:85-134
  RMSNorm::forward          reference src/model/norm.cpp:27-35
  Transformer::embed_tokens reference src/model/transformer.cpp:419-599
  buffers / KV layout       reference src/model/transformer.cpp:330-391

`OracleModel` is pinned bit-for-bit against the reference's own unmodified host code linked with the same
kernels (oracle/_ref/ref_logits; tests/test_oracle_golden.py), so either can be the golden side.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB: Optional[C.CDLL] = None

c_fp = C.POINTER(C.c_float)


def build(force: bool = False) -> str:
    """Compile liboracle.so (and oracle/_ref when /root/reference is present). Returns the .so path."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "ref_launchers_cpu.cpp")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return so


def build_ref() -> bool:
    """Build oracle/_ref from /root/reference when it is present (this container only)."""
    if not os.path.exists("/root/reference/src/main.cpp"):
        return os.path.exists(os.path.join(_HERE, "_ref", "ref_logits"))
    subprocess.check_call(["make", "-s", "-j8", "-C", _HERE, "ref"])
    return True


def host_cpu_budget():
    """(threads, info): the physical cores this process may run on (affinity mask, SMT siblings counted once), capped by the
    cgroup CPU quota when there is one.  The GPU boxes advertise 256 CPUs behind a 16-CPU quota: OpenMP's default of one
    thread per advertised CPU is 16x oversubscribed there (a 12-token tiny-model check took 38 s)."""
    aff = sorted(os.sched_getaffinity(0))
    phys = set()
    for c in aff:
        try:
            sib = open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c).read().strip()
        except OSError:
            sib = str(c)
        phys.add(sib)
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(p)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    threads = len(phys)
    if quota is not None:
        threads = max(1, min(threads, int(quota)))
    try:
        load1 = os.getloadavg()[0]
    except OSError:
        load1 = None
    info = {"nproc_online": os.cpu_count(), "affinity_cpus": len(aff), "physical_cores_in_affinity": len(phys),
            "cgroup_cpu_quota": quota, "loadavg_1m": load1, "threads_used": threads}
    return threads, info


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        _LIB = C.CDLL(build())
        _LIB.oracle_set_threads(C.c_int(host_cpu_budget()[0]))
        _LIB.oracle_h2f.restype = C.c_float
        _LIB.oracle_h2f.argtypes = [C.c_uint16]
        _LIB.oracle_f2h.restype = C.c_uint16
        _LIB.oracle_f2h.argtypes = [C.c_float]
        _LIB.oracle_embed_row.restype = C.c_int
    return _LIB


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


# ---- one function per launcher in reference src/cuda/kernels.h (same argument meaning) -------------
def gemv(W: np.ndarray, x: np.ndarray, out_f: int, in_f: int, dtype: int) -> np.ndarray:
    y = np.zeros(out_f, np.float32)
    x = _f32(x)
    W = np.ascontiguousarray(W)
    lib().oracle_gemv(_p(y), _p(W), _p(x), C.c_int(out_f), C.c_int(in_f), C.c_int(dtype))
    return y


def gemv_add(y: np.ndarray, W: np.ndarray, x: np.ndarray, out_f: int, in_f: int, dtype: int) -> np.ndarray:
    y = _f32(y).copy()
    lib().oracle_gemv_add(_p(y), _p(np.ascontiguousarray(W)), _p(_f32(x)), C.c_int(out_f), C.c_int(in_f), C.c_int(dtype))
    return y


def silu_mul(g: np.ndarray, u: np.ndarray) -> np.ndarray:
    g, u = _f32(g), _f32(u)
    o = np.empty_like(g)
    lib().oracle_silu_mul(_p(o), _p(g), _p(u), C.c_int(g.size))
    return o


def rmsnorm(x: np.ndarray, w: np.ndarray, eps: float) -> np.ndarray:
    x = _f32(x)
    batch = 1 if x.ndim == 1 else x.shape[0]
    hidden = x.shape[-1]
    o = np.empty_like(x)
    lib().oracle_rmsnorm(_p(o), _p(x), _p(_f32(w)), C.c_int(batch), C.c_int(hidden), C.c_float(eps))
    return o


def rope(q: np.ndarray, k: np.ndarray, positions: Sequence[int], nh: int, nkv: int, hd: int, theta: float,
         fscale: float = 1.0, interleaved: bool = False):
    q, k = _f32(q).copy(), _f32(k).copy()
    pos = np.ascontiguousarray(positions, dtype=np.int32)
    lib().oracle_rope(_p(q), _p(k), _p(pos), C.c_int(len(pos)), C.c_int(nh), C.c_int(nkv), C.c_int(hd),
                      C.c_float(theta), C.c_float(fscale), C.c_int(int(interleaved)))
    return q, k


def copy_to_kv_cache(kc: np.ndarray, vc: np.ndarray, k: np.ndarray, v: np.ndarray, seq_len: int, nkv: int,
                     hd: int, start_pos: int, max_seq: int) -> None:
    assert kc.dtype == np.uint16 and vc.dtype == np.uint16
    lib().oracle_copy_to_kv_cache(_p(kc), _p(vc), _p(_f32(k)), _p(_f32(v)), C.c_int(seq_len), C.c_int(nkv),
                                  C.c_int(hd), C.c_int(start_pos), C.c_int(max_seq))


def attention_decode(q: np.ndarray, kc: np.ndarray, vc: np.ndarray, seq_len: int, nh: int, nkv: int, hd: int,
                     max_seq: int, scale: float) -> np.ndarray:
    o = np.zeros(nh * hd, np.float32)
    lib().oracle_attention_decode(_p(o), _p(_f32(q)), _p(kc), _p(vc), C.c_int(seq_len), C.c_int(nh), C.c_int(nkv),
                                  C.c_int(hd), C.c_int(max_seq), C.c_float(scale))
    return o


def attention_prefill(Q: np.ndarray, kc: np.ndarray, vc: np.ndarray, seq_len: int, start_pos: int, nh: int,
                      nkv: int, hd: int, max_seq: int, scale: float) -> np.ndarray:
    o = np.zeros(seq_len * nh * hd, np.float32)
    lib().oracle_attention_prefill(_p(o), _p(_f32(Q)), _p(kc), _p(vc), C.c_int(seq_len), C.c_int(start_pos),
                                   C.c_int(nh), C.c_int(nkv), C.c_int(hd), C.c_int(max_seq), C.c_float(scale))
    return o


def add_inplace(a: np.ndarray, b: np.ndarray) -> None:
    lib().oracle_add_inplace(_p(a), _p(_f32(b)), C.c_int(a.size))


def cosine_similarity(a: np.ndarray, b: np.ndarray) -> float:
    r = np.zeros(1, np.float32)
    lib().oracle_cosine_similarity(_p(r), _p(_f32(a)), _p(_f32(b)), C.c_int(np.size(a)))
    return float(r[0])


def embed_row(table: np.ndarray, token: int, hidden: int, dtype: int) -> np.ndarray:
    o = np.empty(hidden, np.float32)
    lib().oracle_embed_row(_p(o), _p(table), C.c_int(token), C.c_int(hidden), C.c_int(dtype))
    return o


def set_threads(n: int) -> None:
    lib().oracle_set_threads(C.c_int(n))


def max_threads() -> int:
    return int(lib().oracle_get_max_threads())


def pick_threads(candidates=(1, 2, 4, 8, 16, 32, 64, 128)) -> int:
    """Thread count this host actually sustains (containers often expose more CPUs than they schedule):
    time one 2048x4096 Q8_0 GEMV per candidate, keep the fastest."""
    import time
    rng = np.random.default_rng(0)
    W = rng.integers(0, 255, 2048 * 4352, dtype=np.uint8)
    x = rng.standard_normal(4096).astype(np.float32)
    best, best_t = 1, float("inf")
    limit = host_cpu_budget()[0]
    for n in candidates:
        if n > limit:
            break
        set_threads(n)
        gemv(W, x, 2048, 4096, 2)
        t0 = time.perf_counter()
        gemv(W, x, 2048, 4096, 2)
        dt = time.perf_counter() - t0
        if dt < best_t * 0.95:
            best, best_t = n, dt
    set_threads(best)
    return best


def h2f(h: int) -> float:
    return float(lib().oracle_h2f(C.c_uint16(h)))


def f2h(f: float) -> int:
    return int(lib().oracle_f2h(C.c_float(f)))


# ---- host orchestration restated ------------------------------------------------------------------
class OracleModel:
    """Resident-weights Llama forward on the CPU, call for call the reference's launcher sequence."""

    def __init__(self, gguf_path: str, max_context: int = 4096, n_layers: Optional[int] = None):
        import sys
        sys.path.insert(0, os.path.dirname(_HERE))
        from ntransformer_amd import gguf as G
        self.G = G
        self.f = G.read_gguf(gguf_path)
        m = self.f.meta
        arch = m.get("general.architecture", b"llama")
        arch = arch.decode() if isinstance(arch, bytes) else arch
        g = lambda k, d=None: m.get(arch + "." + k, d)
        self.hidden = int(g("embedding_length", 4096))
        self.inter = int(g("feed_forward_length", 11008))
        self.n_layers = int(g("block_count", 32))
        if n_layers is not None:
            self.n_layers = min(self.n_layers, n_layers)   # bounded CPU-baseline sample
        self.nh = int(g("attention.head_count", 32))
        self.nkv = int(g("attention.head_count_kv", self.nh))
        self.hd = self.hidden // self.nh
        self.eps = float(g("attention.layer_norm_rms_epsilon", 1e-5))
        self.theta = float(g("rope.freq_base", 10000.0))
        ctx = int(g("context_length", 4096))
        self.max_seq = min(ctx, max_context)                       # transformer.cpp:70-74
        toks = m.get("tokenizer.ggml.tokens")
        self.vocab = len(toks) if toks else int(g("vocab_size", 32000))   # loader.cpp:139-141
        self.scale = np.float32(1.0) / np.sqrt(np.float32(self.hd))     # attention.cpp:21
        self.out_name = "output.weight" if "output.weight" in self.f.tensors else "token_embd.weight"
        L, S, per = self.n_layers, self.max_seq, self.nkv * self.hd
        self.k_cache = np.zeros((L, S * per), np.uint16)           # transformer.cpp:340-346 (zeroed F16)
        self.v_cache = np.zeros((L, S * per), np.uint16)

    def _gemv(self, name: str, x: np.ndarray, out_f: int, in_f: int) -> np.ndarray:
        return gemv(self.f.raw(name), x, out_f, in_f, self.f.dtype(name))

    def embed(self, tokens: Sequence[int]) -> np.ndarray:
        t = "token_embd.weight"
        return np.stack([embed_row(self.f.raw(t), int(tok), self.hidden, self.f.dtype(t)) for tok in tokens])

    def forward(self, tokens: Sequence[int], start_pos: int, trace: Optional[dict] = None) -> np.ndarray:
        """trace (tests): receives "layer_in" / "layer_out" -- per layer the hidden states [T, H] entering / leaving it"""
        T, H = len(tokens), self.hidden
        q_dim, kv_dim = self.nh * self.hd, self.nkv * self.hd
        hidden = self.embed(tokens)                                            # [T, H]
        positions = [start_pos + i for i in range(T)]
        if trace is not None:
            trace["layer_in"], trace["layer_out"] = [], []
        for i in range(self.n_layers):
            p = "blk.%d." % i
            if trace is not None:
                trace["layer_in"].append(hidden.copy())
            resid = rmsnorm(hidden, self.f.f32(p + "attn_norm.weight"), self.eps)   # :636
            q = np.stack([self._gemv(p + "attn_q.weight", resid[t], q_dim, H) for t in range(T)])
            k = np.stack([self._gemv(p + "attn_k.weight", resid[t], kv_dim, H) for t in range(T)])
            v = np.stack([self._gemv(p + "attn_v.weight", resid[t], kv_dim, H) for t in range(T)])
            q, k = rope(q.reshape(-1), k.reshape(-1), positions, self.nh, self.nkv, self.hd, self.theta)
            copy_to_kv_cache(self.k_cache[i], self.v_cache[i], k, v.reshape(-1), T, self.nkv, self.hd,
                             start_pos, self.max_seq)
            total = start_pos + T
            if T == 1:
                att = attention_decode(q, self.k_cache[i], self.v_cache[i], total, self.nh, self.nkv, self.hd,
                                       self.max_seq, self.scale)
            else:
                att = attention_prefill(q, self.k_cache[i], self.v_cache[i], T, start_pos, self.nh, self.nkv,
                                        self.hd, self.max_seq, self.scale)
            att = att.reshape(T, q_dim)
            resid = np.stack([self._gemv(p + "attn_output.weight", att[t], H, q_dim) for t in range(T)])
            hidden = hidden + resid                                            # add_inplace :647
            resid = rmsnorm(hidden, self.f.f32(p + "ffn_norm.weight"), self.eps)
            outs = []
            for t in range(T):
                gate = self._gemv(p + "ffn_gate.weight", resid[t], self.inter, H)
                up = self._gemv(p + "ffn_up.weight", resid[t], self.inter, H)
                act = silu_mul(gate, up)
                outs.append(self._gemv(p + "ffn_down.weight", act, H, self.inter))
            hidden = hidden + np.stack(outs)                                   # :654
            if trace is not None:
                trace["layer_out"].append(hidden.copy())
        last = rmsnorm(hidden[T - 1], self.f.f32("output_norm.weight"), self.eps)   # :658-659
        return self._gemv(self.out_name, last, self.vocab, H)                  # :662-665

    @staticmethod
    def argmax(logits: np.ndarray) -> int:
        return int(np.argmax(logits))   # first max, as sampler.cpp:18-28 (strict >)


def run_ref_logits(gguf_path: str, prompt: Sequence[int], forced: Sequence[int] = (), n_greedy: int = 0,
                   ctx: int = 4096, out_path: Optional[str] = None, exe_name: str = "ref_logits"):
    """Run oracle/_ref/ref_logits (reference host code + CPU kernels). Returns (fed, argmax, logits[steps,V]).
    exe_name="ref_logits_hip": the same reference host code linked over the product's HIP library (GPU tests)."""
    exe = os.path.join(_HERE, "_ref", exe_name)
    if not os.path.exists(exe):
        raise FileNotFoundError(exe)
    out_path = out_path or (gguf_path + ".ref_logits.bin")
    cmd = [exe, gguf_path, str(ctx), out_path, str(len(prompt)), str(n_greedy)] + [str(t) for t in list(prompt) + list(forced)]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    raw = np.fromfile(out_path, dtype=np.uint8)
    n_steps, V = np.frombuffer(raw[:8].tobytes(), "<i4")
    rec = raw[8:].reshape(n_steps, 8 + 4 * V)
    fed = rec[:, 0:4].copy().view("<i4").reshape(-1)
    am = rec[:, 4:8].copy().view("<i4").reshape(-1)
    logits = rec[:, 8:].copy().view("<f4").reshape(n_steps, V)
    os.remove(out_path)
    return fed, am, logits
