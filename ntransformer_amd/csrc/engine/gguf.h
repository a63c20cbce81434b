// engine/gguf.h -- GGUF v2/v3 reader (mmap, zero-copy tensor views) and model hyper-parameters.
// Replaces reference src/model/loader.{h,cpp} and src/model/config.{h,cpp}; same file-format behaviour
// (cited per function in gguf.cpp), error codes instead of abort().
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

namespace nt {

struct ModelConfig {   // reference src/model/config.h:16-60 (same defaults)
    std::string architecture = "llama";
    std::string model_name = "unknown";
    int vocab_size = 32000;
    int hidden_size = 4096;
    int intermediate_size = 11008;
    int n_layers = 32;
    int n_heads = 32;
    int n_kv_heads = 32;
    int head_dim = 128;
    float norm_eps = 1e-5f;
    float rope_theta = 10000.0f;
    float rope_freq_scale = 1.0f;   // never read from GGUF by the reference (config.h:35): stays 1
    bool rope_interleaved = false;  // never set by the reference (config.h:36): NeoX pairing always
    int max_seq_len = 4096;
    int bos_token_id = 1;
    int eos_token_id = 2;
    void print() const;
};

struct GgufValue {
    enum Kind { NONE, INT, FLOAT, STRING, BOOL } kind = NONE;
    int64_t i = 0;
    double f = 0.0;
    std::string s;
};

struct GgufTensor {
    std::string name;
    std::vector<int64_t> dims;   // ggml order: dims[0] = in_features (fastest)
    uint32_t ggml_type = 0;
    int dtype = 0;               // NTK_DT_* (nt::DType numeric value)
    uint64_t offset = 0;         // relative to the data section
    size_t nbytes = 0;
    bool known_type = true;      // false: a ggml type outside types.h:202-215 (the reference reads those as F32; the model
                                 // loader refuses to USE such a tensor instead of reading quantised bytes as floats)
    // product of the dims; -1 if a dim is <= 0 or the product overflows (file-controlled values)
    int64_t numel() const {
        int64_t n = 1;
        for (auto d : dims) {
            if (d <= 0 || n > INT64_MAX / d) return -1;
            n *= d;
        }
        return n;
    }
};

struct GgufVocab {
    std::vector<std::string> tokens;
    std::vector<float> scores;
    std::vector<int> token_types;
};

class GgufFile {
public:
    GgufFile() = default;
    ~GgufFile();
    GgufFile(const GgufFile&) = delete;
    GgufFile& operator=(const GgufFile&) = delete;

    // 0 on success, NTK_E_IO / NTK_E_FORMAT otherwise (message in error())
    int open(const std::string& path);
    void close();

    const ModelConfig& config() const { return config_; }
    const GgufVocab& vocab() const { return vocab_; }
    const std::vector<GgufTensor>& tensors() const { return tensors_; }
    const GgufTensor* find(const std::string& name) const;
    const void* data(const GgufTensor& t) const { return base_ + data_offset_ + t.offset; }
    const GgufValue* meta(const std::string& key) const;
    size_t file_size() const { return size_; }
    size_t data_offset() const { return data_offset_; }
    uint32_t version() const { return version_; }
    const std::string& error() const { return err_; }
    void print_info() const;

private:
    int parse();
    int fd_ = -1;
    const uint8_t* base_ = nullptr;
    size_t size_ = 0;
    size_t data_offset_ = 0;
    uint32_t version_ = 0;
    std::string path_, err_;
    ModelConfig config_;
    GgufVocab vocab_;
    std::vector<GgufTensor> tensors_;
    std::unordered_map<std::string, size_t> index_;
    std::unordered_map<std::string, GgufValue> meta_;
};

int ggml_type_to_dtype(uint32_t ggml_type);   // reference src/core/types.h:202-215 (unknown -> F32)
bool ggml_type_known(uint32_t ggml_type);     // one of the types that table lists
const char* dtype_name(int dtype);

}  // namespace nt
