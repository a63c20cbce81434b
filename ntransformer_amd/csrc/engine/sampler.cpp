// engine/sampler.cpp -- see sampler.h
#include "sampler.h"

#include <algorithm>
#include <cmath>

namespace nt {

int Sampler::argmax(const float* l, int n) {   // first maximum (sampler.cpp:18-28)
    int best = 0;
    for (int i = 1; i < n; ++i)
        if (l[i] > l[best]) best = i;
    return best;
}

void Sampler::apply_repeat_penalty(float* l, int n, const std::vector<int>& recent) const {
    // one application per occurrence inside the window (sampler.cpp:30-45)
    if (cfg_.repeat_penalty <= 1.0f) return;
    const int have = (int)recent.size();
    for (int i = have - std::min(have, cfg_.repeat_window); i < have; ++i) {
        const int t = recent[i];
        if (t < 0 || t >= n) continue;
        l[t] = l[t] > 0 ? l[t] / cfg_.repeat_penalty : l[t] * cfg_.repeat_penalty;
    }
}

int Sampler::sample(const float* l, int n) {   // sampler.cpp:47-117
    if (cfg_.temperature <= 0.0f) return argmax(l, n);
    cand_.resize(n);
    for (int i = 0; i < n; ++i) cand_[i] = {l[i] / cfg_.temperature, i};
    const auto desc = [](const std::pair<float, int>& a, const std::pair<float, int>& b) { return a.first > b.first; };
    if (cfg_.top_k > 0 && cfg_.top_k < n) {
        std::partial_sort(cand_.begin(), cand_.begin() + cfg_.top_k, cand_.end(), desc);
        cand_.resize(cfg_.top_k);
    } else {
        std::sort(cand_.begin(), cand_.end(), desc);
    }
    const float top = cand_[0].first;
    float z = 0.0f;
    for (auto& c : cand_) { c.first = expf(c.first - top); z += c.first; }
    for (auto& c : cand_) c.first /= z;
    if (cfg_.top_p < 1.0f && cfg_.top_p > 0.0f) {
        float cum = 0.0f;
        size_t keep = cand_.size();
        for (size_t i = 0; i < cand_.size(); ++i) {
            cum += cand_[i].first;
            if (cum >= cfg_.top_p) { keep = i + 1; break; }
        }
        cand_.resize(keep);
        z = 0.0f;
        for (auto& c : cand_) z += c.first;
        for (auto& c : cand_) c.first /= z;
    }
    std::uniform_real_distribution<float> uni(0.0f, 1.0f);
    const float r = uni(rng_);
    float cum = 0.0f;
    for (const auto& c : cand_) {
        cum += c.first;
        if (r <= cum) return c.second;
    }
    return cand_.back().second;
}

}  // namespace nt
