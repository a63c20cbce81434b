// engine/sampler.h -- host-side token sampling with the reference's semantics
// (reference src/inference/sampler.{h,cpp}): greedy when temperature <= 0, otherwise temperature ->
// top-k (partial sort) -> softmax -> top-p cut + renormalise -> one draw from std::mt19937.
#pragma once
#include <cstdint>
#include <random>
#include <utility>
#include <vector>

namespace nt {

struct SamplerConfig {
    float temperature = 0.7f;
    int top_k = 40;
    float top_p = 0.9f;
    float repeat_penalty = 1.1f;
    int repeat_window = 64;
    uint64_t seed = 42;
};

class Sampler {
public:
    void init(const SamplerConfig& c) { cfg_ = c; rng_.seed(c.seed); }
    void set_seed(uint64_t s) { cfg_.seed = s; rng_.seed(s); }
    static int argmax(const float* logits, int n);
    void apply_repeat_penalty(float* logits, int n, const std::vector<int>& recent) const;
    int sample(const float* logits, int n);
    bool is_pure_greedy() const { return cfg_.temperature <= 0.0f && cfg_.repeat_penalty <= 1.0f; }
    // the uniform draw sample() takes from the generator (sampler.cpp:103-104), for a sampler that runs elsewhere (the device):
    // exactly one per sampled token, so the generator stays in step with the reference's
    float draw() { std::uniform_real_distribution<float> uni(0.0f, 1.0f); return uni(rng_); }
    const SamplerConfig& config() const { return cfg_; }

private:
    SamplerConfig cfg_;
    std::mt19937 rng_;
    std::vector<std::pair<float, int>> cand_;
};

}  // namespace nt
