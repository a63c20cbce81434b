// engine/tokenizer.h -- prompt <-> token ids with the reference's exact behaviour
// (reference src/inference/tokenizer.{h,cpp}): vocabulary-only, `tokenizer.ggml.merges` is ignored;
// text is mapped to GPT-2 byte-unicode (when the vocab contains "Ġ") or SentencePiece "▁" form, split by
// greedy longest match (<= 64 bytes) with byte fallback, then adjacent pieces are merged best-score-first.
#pragma once
#include <string>
#include <unordered_map>
#include <vector>
#include "gguf.h"

namespace nt {

class Tokenizer {
public:
    void init(const GgufVocab& vocab, int bos_id, int eos_id);
    std::vector<int> encode(const std::string& text, bool add_bos = true) const;
    std::string decode(const std::vector<int>& ids) const;
    std::string decode_token(int id) const;
    int bos_id() const { return bos_; }
    int eos_id() const { return eos_; }
    int vocab_size() const { return (int)pieces_.size(); }
    bool gpt2_mode() const { return gpt2_; }

private:
    int lookup(const std::string& s) const;
    int byte_fallback(unsigned char b) const;
    std::vector<std::string> pieces_;
    std::vector<float> scores_;
    std::vector<int> types_;
    std::unordered_map<std::string, int> ids_;
    int bos_ = 1, eos_ = 2;
    bool gpt2_ = false;
};

}  // namespace nt
