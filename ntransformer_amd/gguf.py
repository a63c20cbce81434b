"""GGUF v3 container: writer, reader and seeded synthetic Llama-shaped models (numpy only).

The reference can only *read* GGUF (reference src/model/loader.cpp:56-187); it ships no writer and no
model files exist in this environment, so every parity/bench input is a synthetic file produced here
(small fixtures, tests) or by the C++ generator in csrc/synth.cpp (8B/70B shapes, same container).

Layout facts this file must agree with the reference loader on:
  * header: magic "GGUF", version 3, n_tensors u64, n_kv u64                      (loader.cpp:60-83)
  * kv: key string, type u32, value; arrays = elem type u32 + count u64 + elems   (loader.cpp:91-131)
  * keys read: general.architecture/name/alignment, <arch>.{vocab_size, embedding_length,
    feed_forward_length, block_count, attention.head_count, attention.head_count_kv, context_length,
    attention.layer_norm_rms_epsilon, rope.freq_base}, tokenizer.ggml.{tokens, scores, token_type,
    bos_token_id, eos_token_id}                                  (reference src/model/config.cpp:24-49)
  * tensor info: name, n_dims u32, dims u64[n] in ggml order (fastest first = in_features first),
    ggml type u32, offset u64 relative to the aligned data section      (loader.cpp:146-170,177-185)
  * quant block byte layouts                                     (reference src/core/types.h:96-137)
"""
from __future__ import annotations

import mmap
import struct
from dataclasses import dataclass, field
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

GGUF_MAGIC = 0x46554747

# GGUF metadata value types (reference src/core/types.h:156-170)
T_U8, T_I8, T_U16, T_I16, T_U32, T_I32, T_F32, T_BOOL, T_STR, T_ARR, T_U64, T_I64, T_F64 = range(13)

# ggml tensor types the reference maps (types.h:173-215)
GGML_F32, GGML_F16, GGML_Q4_0, GGML_Q8_0, GGML_Q4_K, GGML_Q5_K, GGML_Q6_K = 0, 1, 2, 8, 12, 13, 14

# nt::DType numeric values (types.h:24-35) -- the C ABI contract
DT_F32, DT_F16, DT_Q8_0, DT_Q4_0, DT_Q4_K, DT_Q6_K, DT_Q5_K, DT_Q2_K, DT_I32 = range(9)

GGML_TO_DT = {GGML_F32: DT_F32, GGML_F16: DT_F16, GGML_Q4_0: DT_Q4_0, GGML_Q8_0: DT_Q8_0,
              GGML_Q4_K: DT_Q4_K, GGML_Q5_K: DT_Q5_K, GGML_Q6_K: DT_Q6_K}
DT_TO_GGML = {v: k for k, v in GGML_TO_DT.items()}
DT_NAME = {DT_F32: "F32", DT_F16: "F16", DT_Q8_0: "Q8_0", DT_Q4_0: "Q4_0", DT_Q4_K: "Q4_K",
           DT_Q6_K: "Q6_K", DT_Q5_K: "Q5_K"}
NAME_TO_GGML = {"F32": GGML_F32, "F16": GGML_F16, "Q4_0": GGML_Q4_0, "Q8_0": GGML_Q8_0,
                "Q4_K": GGML_Q4_K, "Q5_K": GGML_Q5_K, "Q6_K": GGML_Q6_K}

# (elements per block, bytes per block)
BLOCK = {GGML_F32: (1, 4), GGML_F16: (1, 2), GGML_Q4_0: (32, 18), GGML_Q8_0: (32, 34),
         GGML_Q4_K: (256, 144), GGML_Q5_K: (256, 176), GGML_Q6_K: (256, 210)}


def row_bytes(ggml_type: int, n: int) -> int:
    be, bb = BLOCK[ggml_type]
    assert n % be == 0, (ggml_type, n)
    return (n // be) * bb


# ----------------------------------------------------------------------------------------------
# writer
# ----------------------------------------------------------------------------------------------
def _s(b: bytes | str) -> bytes:
    if isinstance(b, str):
        b = b.encode("utf-8")
    return struct.pack("<Q", len(b)) + b


def _kv(key: str, vtype: int, payload: bytes) -> bytes:
    return _s(key) + struct.pack("<I", vtype) + payload


def kv_u32(key, v): return _kv(key, T_U32, struct.pack("<I", v))
def kv_f32(key, v): return _kv(key, T_F32, struct.pack("<f", v))
def kv_str(key, v): return _kv(key, T_STR, _s(v))


def kv_arr_str(key, items: Sequence[bytes | str]) -> bytes:
    body = b"".join(_s(x) for x in items)
    return _kv(key, T_ARR, struct.pack("<IQ", T_STR, len(items)) + body)


def kv_arr_i32(key, items: Sequence[int]) -> bytes:
    return _kv(key, T_ARR, struct.pack("<IQ", T_I32, len(items)) + np.asarray(items, "<i4").tobytes())


def kv_arr_f32(key, items: Sequence[float]) -> bytes:
    return _kv(key, T_ARR, struct.pack("<IQ", T_F32, len(items)) + np.asarray(items, "<f4").tobytes())


@dataclass
class TensorSpec:
    name: str
    dims: Tuple[int, ...]      # ggml order: (in_features, out_features) for matrices, (n,) for vectors
    ggml_type: int
    data: Optional[bytes] = None   # raw bytes; or a callable returning bytes (lazy, for big files)

    @property
    def nbytes(self) -> int:
        n = 1
        for d in self.dims:
            n *= d
        return row_bytes(self.ggml_type, n)


def write_gguf(path: str, kvs: Sequence[bytes], tensors: Sequence[TensorSpec], alignment: int = 32) -> None:
    """Write a GGUF v3 file. Tensor data is laid out in order, each tensor aligned to `alignment`."""
    header = struct.pack("<IIQQ", GGUF_MAGIC, 3, len(tensors), len(kvs)) + b"".join(kvs)
    infos = []
    off = 0
    offsets = []
    for t in tensors:
        offsets.append(off)
        infos.append(_s(t.name) + struct.pack("<I", len(t.dims)) + struct.pack("<%dQ" % len(t.dims), *t.dims)
                     + struct.pack("<IQ", t.ggml_type, off))
        off += (t.nbytes + alignment - 1) // alignment * alignment
    head = header + b"".join(infos)
    pad = (-len(head)) % alignment
    with open(path, "wb") as f:
        f.write(head + b"\0" * pad)
        for t, o in zip(tensors, offsets):
            data = t.data() if callable(t.data) else t.data
            assert data is not None and len(data) == t.nbytes, (t.name, len(data or b""), t.nbytes)
            f.write(data)
            f.write(b"\0" * ((-t.nbytes) % alignment))


# ----------------------------------------------------------------------------------------------
# reader (independent of the C++ loader; used by the oracle-side forward and by tests)
# ----------------------------------------------------------------------------------------------
@dataclass
class TensorInfo:
    name: str
    dims: Tuple[int, ...]
    ggml_type: int
    offset: int
    nbytes: int


@dataclass
class GGUFFile:
    path: str
    version: int
    meta: Dict[str, object]
    tensors: Dict[str, TensorInfo]
    data_offset: int
    _mm: Optional[mmap.mmap] = field(default=None, repr=False)
    _buf: Optional[np.ndarray] = field(default=None, repr=False)

    def raw(self, name: str) -> np.ndarray:
        ti = self.tensors[name]
        a = self.data_offset + ti.offset
        return self._buf[a:a + ti.nbytes]

    def dtype(self, name: str) -> int:
        return GGML_TO_DT[self.tensors[name].ggml_type]

    def f32(self, name: str) -> np.ndarray:
        assert self.tensors[name].ggml_type == GGML_F32
        return self.raw(name).view("<f4")

    def close(self):
        self._buf = None
        if self._mm is not None:
            self._mm.close()
            self._mm = None


def read_gguf(path: str) -> GGUFFile:
    f = open(path, "rb")
    mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
    buf = np.frombuffer(mm, dtype=np.uint8)
    pos = 0

    def rd(fmt):
        nonlocal pos
        v = struct.unpack_from("<" + fmt, mm, pos)
        pos += struct.calcsize("<" + fmt)
        return v if len(v) > 1 else v[0]

    def rstr():
        nonlocal pos
        n = rd("Q")
        s = bytes(mm[pos:pos + n])
        pos += n
        return s

    scalar = {T_U8: "B", T_I8: "b", T_U16: "H", T_I16: "h", T_U32: "I", T_I32: "i", T_F32: "f",
              T_BOOL: "?", T_U64: "Q", T_I64: "q", T_F64: "d"}

    def rval(t):
        if t == T_STR:
            return rstr()
        if t == T_ARR:
            et, n = rd("I"), rd("Q")
            return [rval(et) for _ in range(n)]
        return rd(scalar[t])

    magic, version = rd("I"), rd("I")
    if magic != GGUF_MAGIC:
        raise ValueError("bad GGUF magic 0x%08x" % magic)
    if version not in (2, 3):
        raise ValueError("unsupported GGUF version %d" % version)
    n_tensors, n_kv = rd("Q"), rd("Q")
    meta: Dict[str, object] = {}
    for _ in range(n_kv):
        k = rstr().decode("utf-8")
        t = rd("I")
        meta[k] = rval(t)
    tensors: Dict[str, TensorInfo] = {}
    for _ in range(n_tensors):
        name = rstr().decode("utf-8")
        nd = rd("I")
        dims = tuple(rd("Q") for _ in range(nd))
        gt, off = rd("I"), rd("Q")
        n = 1
        for d in dims:
            n *= d
        tensors[name] = TensorInfo(name, dims, gt, off, row_bytes(gt, n))
    align = int(meta.get("general.alignment", 32))
    data_offset = (pos + align - 1) // align * align
    return GGUFFile(path, version, meta, tensors, data_offset, mm, buf)


# ----------------------------------------------------------------------------------------------
# synthetic quantised tensors (blocks generated directly)
# ----------------------------------------------------------------------------------------------
def _f16_bytes(x: np.ndarray) -> np.ndarray:
    return x.astype("<f2").view(np.uint8).reshape(x.shape + (2,))


def synth_tensor(rng: np.random.Generator, ggml_type: int, out_f: int, in_f: int, sigma: float = 1.0) -> bytes:
    """Random [out_f, in_f] matrix in `ggml_type` encoding with dequantised RMS ~ sigma/sqrt(in_f)."""
    be, bb = BLOCK[ggml_type]
    nb = out_f * in_f // be
    tgt = sigma / np.sqrt(in_f)
    jit = (1.0 + 0.25 * rng.uniform(-1, 1, nb)).astype(np.float32)
    if ggml_type == GGML_F32:
        return (rng.standard_normal(out_f * in_f) * tgt).astype("<f4").tobytes()
    if ggml_type == GGML_F16:
        return (rng.standard_normal(out_f * in_f) * tgt).astype("<f2").tobytes()
    blk = np.empty((nb, bb), np.uint8)
    if ggml_type == GGML_Q8_0:
        blk[:, 0:2] = _f16_bytes(tgt / 73.3 * jit)
        blk[:, 2:] = rng.integers(-127, 128, (nb, 32), dtype=np.int8).view(np.uint8)
    elif ggml_type == GGML_Q4_0:
        blk[:, 0:2] = _f16_bytes(tgt / 4.6 * jit)
        blk[:, 2:] = rng.integers(0, 256, (nb, 16), dtype=np.uint8)
    elif ggml_type in (GGML_Q4_K, GGML_Q5_K):
        mean_q = 7.5 if ggml_type == GGML_Q4_K else 15.5
        dev_q = 4.6 if ggml_type == GGML_Q4_K else 9.2
        d = tgt / (32.0 * dev_q * 1.6) * jit
        blk[:, 0:2] = _f16_bytes(d)
        blk[:, 2:4] = _f16_bytes(d * mean_q)             # dmin*m ~ d*sc*mean(q): roughly centred
        blk[:, 4:16] = rng.integers(0, 256, (nb, 12), dtype=np.uint8)
        blk[:, 16:] = rng.integers(0, 256, (nb, bb - 16), dtype=np.uint8)
    elif ggml_type == GGML_Q6_K:
        blk[:, 0:192] = rng.integers(0, 256, (nb, 192), dtype=np.uint8)
        blk[:, 192:208] = rng.integers(-64, 64, (nb, 16), dtype=np.int8).view(np.uint8)
        blk[:, 208:210] = _f16_bytes(tgt / (37.0 * 18.5) * jit)
    else:
        raise ValueError(ggml_type)
    return blk.tobytes()


# ----------------------------------------------------------------------------------------------
# dequantisation (numpy, written from the block definitions in SURVEY Appendix A; an independent
# statement used to cross-check the oracle, the HIP embedding gather and the golden fixtures)
# ----------------------------------------------------------------------------------------------
def _kq_scales(sc: np.ndarray):
    """sc: [nb, 12] uint8 -> (scale[nb,8], min[nb,8]) 6-bit values."""
    sc = sc.astype(np.int32)
    s = np.empty(sc.shape[:-1] + (8,), np.int32)
    m = np.empty_like(s)
    s[..., :4] = sc[..., 0:4] & 63
    m[..., :4] = sc[..., 4:8] & 63
    s[..., 4:] = (sc[..., 8:12] & 0xF) | ((sc[..., 0:4] >> 6) << 4)
    m[..., 4:] = (sc[..., 8:12] >> 4) | ((sc[..., 4:8] >> 6) << 4)
    return s, m


def dequantize(raw: np.ndarray | bytes, ggml_type: int, n: int) -> np.ndarray:
    """raw bytes of n elements -> float32[n] (same element order as the row)."""
    raw = np.frombuffer(raw, np.uint8) if not isinstance(raw, np.ndarray) else raw.view(np.uint8).reshape(-1)
    be, bb = BLOCK[ggml_type]
    nb = n // be
    if ggml_type == GGML_F32:
        return raw.view("<f4").astype(np.float32)
    if ggml_type == GGML_F16:
        return raw.view("<f2").astype(np.float32)
    b = raw[: nb * bb].reshape(nb, bb)
    if ggml_type == GGML_Q8_0:
        d = b[:, 0:2].copy().view("<f2").astype(np.float32)
        return (d * b[:, 2:].view(np.int8).astype(np.float32)).reshape(-1)
    if ggml_type == GGML_Q4_0:
        d = b[:, 0:2].copy().view("<f2").astype(np.float32)
        q = b[:, 2:]
        lo = (q & 0xF).astype(np.float32) - 8
        hi = (q >> 4).astype(np.float32) - 8
        return (d * np.concatenate([lo, hi], 1)).reshape(-1)
    if ggml_type in (GGML_Q4_K, GGML_Q5_K):
        d = b[:, 0:2].copy().view("<f2").astype(np.float32)
        dm = b[:, 2:4].copy().view("<f2").astype(np.float32)
        s, m = _kq_scales(b[:, 4:16])
        if ggml_type == GGML_Q4_K:
            qs = b[:, 16:144].reshape(nb, 4, 32)
            lo = (qs & 0xF).astype(np.float32)
            hi = (qs >> 4).astype(np.float32)
        else:
            qh = b[:, 16:48].astype(np.int32)[:, None, :]
            ql = b[:, 48:176].reshape(nb, 4, 32).astype(np.int32)
            c = np.arange(4)[None, :, None]
            lo = ((ql & 0xF) + (((qh >> (2 * c)) & 1) << 4)).astype(np.float32)
            hi = ((ql >> 4) + (((qh >> (2 * c + 1)) & 1) << 4)).astype(np.float32)
        q = np.stack([lo, hi], 2).reshape(nb, 8, 32)         # sub-block order 0..7
        w = (d * s.astype(np.float32))[:, :, None] * q - (dm * m.astype(np.float32))[:, :, None]
        return w.reshape(-1).astype(np.float32)
    if ggml_type == GGML_Q6_K:
        ql = b[:, 0:128].reshape(nb, 2, 64).astype(np.int32)
        qh = b[:, 128:192].reshape(nb, 2, 32).astype(np.int32)
        sc = b[:, 192:208].view(np.int8).reshape(nb, 2, 8).astype(np.float32)
        d = b[:, 208:210].copy().view("<f2").astype(np.float32)
        q1 = ((ql[:, :, :32] & 0xF) | (((qh >> 0) & 3) << 4)) - 32
        q2 = ((ql[:, :, 32:] & 0xF) | (((qh >> 2) & 3) << 4)) - 32
        q3 = ((ql[:, :, :32] >> 4) | (((qh >> 4) & 3) << 4)) - 32
        q4 = ((ql[:, :, 32:] >> 4) | (((qh >> 6) & 3) << 4)) - 32
        q = np.stack([q1, q2, q3, q4], 2).astype(np.float32)          # [nb, 2, 4, 32]
        isx = np.arange(32) // 16
        scl = np.stack([sc[:, :, isx + 2 * g] for g in range(4)], 2)  # [nb, 2, 4, 32]
        w = d[:, :, None, None] * scl * q
        return w.reshape(-1).astype(np.float32)
    raise ValueError(ggml_type)


# ----------------------------------------------------------------------------------------------
# synthetic Llama-shaped models
# ----------------------------------------------------------------------------------------------
@dataclass
class LlamaShape:
    name: str
    hidden: int
    inter: int
    layers: int
    heads: int
    kv_heads: int
    vocab: int
    ctx: int = 131072
    eps: float = 1e-5
    theta: float = 500000.0
    bos: int = 1
    eos: int = 2


TINY = LlamaShape("tiny", 256, 512, 2, 4, 2, 512, ctx=256, bos=256, eos=257)
SMALL = LlamaShape("small", 1024, 2048, 4, 8, 2, 2048, ctx=2048, bos=256, eos=257)       # hd=128, GQA 4
LLAMA_8B = LlamaShape("llama3.1-8b", 4096, 14336, 32, 32, 8, 128256, bos=128000, eos=128009)
LLAMA_70B = LlamaShape("llama3.1-70b", 8192, 28672, 80, 64, 8, 128256, bos=128000, eos=128009)


def use_more_bits(i: int, n: int) -> bool:
    """llama.cpp's Q4_K_M rule."""
    return i < n // 8 or i >= 7 * n // 8 or (i - n // 8) % 3 == 2


def tensor_types(shape: LlamaShape, mix: str) -> Dict[str, int]:
    """name -> ggml type for a quantisation mix: a plain type name or 'Q4_K_M'."""
    L = shape.layers
    out: Dict[str, int] = {}
    mats = ["attn_q", "attn_k", "attn_v", "attn_output", "ffn_gate", "ffn_up", "ffn_down"]
    if mix == "Q4_K_M":
        gqa = shape.heads // shape.kv_heads
        out["token_embd.weight"] = GGML_Q4_K
        out["output.weight"] = GGML_Q6_K
        for i in range(L):
            for m in mats:
                t = GGML_Q4_K
                if m in ("attn_v", "ffn_down") and use_more_bits(i, L):
                    t = GGML_Q6_K
                elif m == "attn_v" and gqa >= 4 and shape.hidden >= 8192:
                    t = GGML_Q5_K  # 70B-class GQA: remaining attn_v -> Q5_K
                out["blk.%d.%s.weight" % (i, m)] = t
    elif mix == "MIXED":  # every quant type in one small file (tests)
        cyc = [GGML_Q8_0, GGML_Q4_0, GGML_Q4_K, GGML_Q5_K, GGML_Q6_K, GGML_F16, GGML_F32]
        out["token_embd.weight"] = GGML_Q4_K
        out["output.weight"] = GGML_Q6_K
        k = 0
        for i in range(L):
            for m in mats:
                out["blk.%d.%s.weight" % (i, m)] = cyc[k % len(cyc)]
                k += 1
    else:
        t = NAME_TO_GGML[mix]
        # the reference has no Q5_K branch in embed_tokens (it zero-fills, transformer.cpp:595-598),
        # so a plain-Q5_K test file keeps a Q4_K embedding like real llama.cpp K-quant files do
        out["token_embd.weight"] = GGML_Q4_K if t == GGML_Q5_K else t
        out["output.weight"] = t
        for i in range(L):
            for m in mats:
                out["blk.%d.%s.weight" % (i, m)] = t
    return out


def gpt2_byte_alphabet() -> List[bytes]:
    """The 256 single-'character' tokens of GPT-2 byte-level BPE (reference tokenizer.cpp:14-52)."""
    out, n = [], 0
    for b in range(256):
        ident = (33 <= b <= 126) or (161 <= b <= 172) or (174 <= b <= 255)
        cp = b if ident else 256 + n
        if not ident:
            n += 1
        out.append(chr(cp).encode("utf-8"))
    return out


def synth_vocab(shape: LlamaShape, rng: np.random.Generator) -> Tuple[List[bytes], List[int]]:
    """GPT2-BPE style vocab: 256 byte tokens (so 'Ġ' exists and GPT2 mode triggers, tokenizer.cpp:80-81),
    special BOS/EOS as control tokens, remaining ids = unique multi-byte merges of ASCII letters."""
    alpha = gpt2_byte_alphabet()
    toks: List[bytes] = list(alpha)
    types = [1] * 256
    letters = [alpha[c] for c in b"abcdefghijklmnopqrstuvwxyz"] + [alpha[0x20]]
    seen = set(toks)
    i = 0
    while len(toks) < shape.vocab:
        tid = len(toks)
        if tid in (shape.bos, shape.eos):
            toks.append(b"<|special_%d|>" % tid)
            types.append(3)
            continue
        ln = 2 + (i % 4)
        cand = b"".join(letters[int(k)] for k in rng.integers(0, len(letters), ln))
        i += 1
        if cand in seen:
            cand = cand + b"_%d" % tid
        seen.add(cand)
        toks.append(cand)
        types.append(1)
    return toks, types


def make_synthetic_llama(path: str, shape: LlamaShape, mix: str = "Q8_0", seed: int = 20260925,
                         with_scores: bool = False) -> Dict[str, int]:
    """Write a seeded synthetic Llama-architecture GGUF; returns {tensor name: ggml type}."""
    types = tensor_types(shape, mix)
    hd = shape.hidden // shape.heads
    q_dim, kv_dim = shape.heads * hd, shape.kv_heads * hd
    dims = {"attn_q": (shape.hidden, q_dim), "attn_k": (shape.hidden, kv_dim), "attn_v": (shape.hidden, kv_dim),
            "attn_output": (q_dim, shape.hidden), "ffn_gate": (shape.hidden, shape.inter),
            "ffn_up": (shape.hidden, shape.inter), "ffn_down": (shape.inter, shape.hidden)}

    def rng_for(name: str) -> np.random.Generator:
        h = 1469598103934665603
        for c in name.encode():
            h = ((h ^ c) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return np.random.Generator(np.random.Philox(key=[seed & 0xFFFFFFFFFFFFFFFF, h]))

    def norm_w(name: str) -> bytes:
        r = rng_for(name)
        return (1.0 + 0.05 * r.uniform(-1, 1, shape.hidden)).astype("<f4").tobytes()

    specs: List[TensorSpec] = []

    def mat(name: str, in_f: int, out_f: int, sigma: float = 1.0):
        t = types[name]
        specs.append(TensorSpec(name, (in_f, out_f), t,
                                (lambda n=name, tt=t, o=out_f, i=in_f, s=sigma: synth_tensor(rng_for(n), tt, o, i, s))))

    mat("token_embd.weight", shape.hidden, shape.vocab, sigma=float(np.sqrt(shape.hidden)))  # rows RMS ~ 1
    for i in range(shape.layers):
        p = "blk.%d." % i
        specs.append(TensorSpec(p + "attn_norm.weight", (shape.hidden,), GGML_F32, norm_w(p + "attn_norm.weight")))
        for m in ("attn_q", "attn_k", "attn_v", "attn_output"):
            mat(p + m + ".weight", *dims[m])
        specs.append(TensorSpec(p + "ffn_norm.weight", (shape.hidden,), GGML_F32, norm_w(p + "ffn_norm.weight")))
        for m in ("ffn_gate", "ffn_up", "ffn_down"):
            mat(p + m + ".weight", *dims[m])
    specs.append(TensorSpec("output_norm.weight", (shape.hidden,), GGML_F32, norm_w("output_norm.weight")))
    mat("output.weight", shape.hidden, shape.vocab, sigma=2.0)

    toks, ttypes = synth_vocab(shape, np.random.Generator(np.random.Philox(key=[seed, 7])))
    kvs = [
        kv_str("general.architecture", "llama"),
        kv_str("general.name", "synthetic-%s-%s" % (shape.name, mix)),
        kv_u32("general.alignment", 32),
        kv_u32("llama.vocab_size", shape.vocab),
        kv_u32("llama.embedding_length", shape.hidden),
        kv_u32("llama.feed_forward_length", shape.inter),
        kv_u32("llama.block_count", shape.layers),
        kv_u32("llama.attention.head_count", shape.heads),
        kv_u32("llama.attention.head_count_kv", shape.kv_heads),
        kv_u32("llama.context_length", shape.ctx),
        kv_f32("llama.attention.layer_norm_rms_epsilon", shape.eps),
        kv_f32("llama.rope.freq_base", shape.theta),
        kv_arr_str("tokenizer.ggml.tokens", toks),
        kv_arr_i32("tokenizer.ggml.token_type", ttypes),
        kv_u32("tokenizer.ggml.bos_token_id", shape.bos),
        kv_u32("tokenizer.ggml.eos_token_id", shape.eos),
    ]
    if with_scores:
        sc = -np.arange(shape.vocab, dtype=np.float32)  # earlier id = better merge
        kvs.append(kv_arr_f32("tokenizer.ggml.scores", sc))
    write_gguf(path, kvs, specs)
    return types


def write_vocab_gguf(path: str, tokens: Sequence[bytes], scores: Optional[Sequence[float]], types: Sequence[int],
                     bos: int, eos: int) -> None:
    """A GGUF that carries only a vocabulary (plus one dummy tensor): tokenizer fixtures."""
    kvs = [kv_str("general.architecture", "llama"), kv_str("general.name", "vocab-only"), kv_u32("general.alignment", 32),
           kv_u32("llama.embedding_length", 32), kv_u32("llama.attention.head_count", 1), kv_u32("llama.block_count", 0),
           kv_arr_str("tokenizer.ggml.tokens", tokens), kv_arr_i32("tokenizer.ggml.token_type", types),
           kv_u32("tokenizer.ggml.bos_token_id", bos), kv_u32("tokenizer.ggml.eos_token_id", eos)]
    if scores is not None:
        kvs.append(kv_arr_f32("tokenizer.ggml.scores", scores))
    write_gguf(path, kvs, [TensorSpec("dummy.weight", (32,), GGML_F32, np.zeros(32, "<f4").tobytes())])


def algorithmic_bytes_per_token(shape: LlamaShape, types: Dict[str, int], pos: int = 0) -> int:
    """B_tok of SURVEY.md section 8(d): matrices once in GGUF encoding + norms + KV r/w + 1 embedding row."""
    hd = shape.hidden // shape.heads
    total = 0
    for name, t in types.items():
        if name == "token_embd.weight":
            total += row_bytes(t, shape.hidden)
            continue
        if name == "output.weight":
            total += row_bytes(t, shape.hidden) * shape.vocab
            continue
        m = name.split(".")[2]
        in_f, out_f = {"attn_q": (shape.hidden, shape.heads * hd), "attn_k": (shape.hidden, shape.kv_heads * hd),
                       "attn_v": (shape.hidden, shape.kv_heads * hd), "attn_output": (shape.heads * hd, shape.hidden),
                       "ffn_gate": (shape.hidden, shape.inter), "ffn_up": (shape.hidden, shape.inter),
                       "ffn_down": (shape.inter, shape.hidden)}[m]
        total += row_bytes(t, in_f) * out_f
    total += (2 * shape.layers + 1) * shape.hidden * 4
    total += 2 * shape.layers * shape.kv_heads * hd * 2 * (pos + 1)   # KV read, once per KV head
    total += 2 * shape.layers * shape.kv_heads * hd * 2               # KV write
    return total
