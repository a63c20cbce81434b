// engine/synth.h -- seeded synthetic Llama-shaped weights in GGUF block encodings.
// No reference counterpart: the reference only ever loads real checkpoints, none of which exist in this
// environment (no network).  Blocks are generated directly (quants uniform over their integer range,
// FP16 super-scales chosen so dequantised rows have RMS ~ sigma/sqrt(in)), SURVEY.md section 8(d).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace nt {

struct SynthTensor {
    std::string name;
    int ggml_type;
    int64_t in_f, out_f;   // vectors: in_f = n, out_f = 1
    double sigma;
    size_t nbytes;
};

struct SynthSpec {
    int hidden = 4096, inter = 14336, layers = 32, heads = 32, kv_heads = 8, vocab = 128256;
    int ctx = 131072;
    float eps = 1e-5f, theta = 500000.0f;
    int bos = 128000, eos = 128009;
    std::string mix = "Q8_0";   // Q8_0 | Q4_0 | Q4_K | Q5_K | Q6_K | F16 | F32 | Q4_K_M
    uint64_t seed = 20260925;
};

// tensor list in file order (token_embd, per-layer 9 tensors, output_norm, output); false if `mix` is unknown
bool synth_plan(const SynthSpec& s, std::vector<SynthTensor>& out);
// deterministic in (seed, name, element index); multi-threaded
void synth_fill(void* dst, const SynthTensor& t, uint64_t seed, int nthreads);
// GPT-2 style synthetic vocabulary: 256 byte tokens first, specials as control tokens
void synth_vocab(const SynthSpec& s, std::vector<std::string>& tokens, std::vector<int>& types);
// write a complete GGUF v3 file; 0 on success
int synth_write_gguf(const std::string& path, const SynthSpec& s, int nthreads);

}  // namespace nt
