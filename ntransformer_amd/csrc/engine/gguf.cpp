// engine/gguf.cpp -- see gguf.h.  Bounds-checked cursor over the mmap instead of raw pointer walks.
#include "gguf.h"
#include "../../../include/ntk.h"

#include <cinttypes>
#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace nt {

enum : uint32_t { GT_U8, GT_I8, GT_U16, GT_I16, GT_U32, GT_I32, GT_F32, GT_BOOL, GT_STR, GT_ARR, GT_U64, GT_I64, GT_F64 };

int ggml_type_to_dtype(uint32_t t) {
    switch (t) {   // reference src/core/types.h:202-215
        case 0: return NTK_DT_F32;
        case 1: return NTK_DT_F16;
        case 2: return NTK_DT_Q4_0;
        case 8: return NTK_DT_Q8_0;
        case 10: return NTK_DT_Q2_K;
        case 12: return NTK_DT_Q4_K;
        case 13: return NTK_DT_Q5_K;
        case 14: return NTK_DT_Q6_K;
        case 26: return NTK_DT_I32;
        default: return NTK_DT_F32;   // the reference's fallback
    }
}

bool ggml_type_known(uint32_t t) {
    switch (t) { case 0: case 1: case 2: case 8: case 10: case 12: case 13: case 14: case 26: return true; default: return false; }
}

const char* dtype_name(int dt) {
    static const char* n[] = {"F32", "F16", "Q8_0", "Q4_0", "Q4_K_M", "Q6_K", "Q5_K", "Q2_K", "I32"};
    return dt >= 0 && dt <= 8 ? n[dt] : "UNKNOWN";
}

void ModelConfig::print() const {   // same fields the reference prints (config.cpp:52-65)
    fprintf(stderr, "=== Model Config ===\nArchitecture: %s\nName: %s\n", architecture.c_str(), model_name.c_str());
    fprintf(stderr, "Vocab: %d, Hidden: %d, Intermediate: %d\n", vocab_size, hidden_size, intermediate_size);
    fprintf(stderr, "Layers: %d, Heads: %d, KV Heads: %d, Head dim: %d\n", n_layers, n_heads, n_kv_heads, head_dim);
    fprintf(stderr, "Max seq: %d, Norm eps: %e\n", max_seq_len, norm_eps);
    fprintf(stderr, "RoPE theta: %.1f, GQA: %s (group=%d)\n", rope_theta, n_kv_heads < n_heads ? "yes" : "no",
            n_kv_heads > 0 ? n_heads / n_kv_heads : 0);
    fprintf(stderr, "BOS: %d, EOS: %d\n", bos_token_id, eos_token_id);
}

namespace {
struct Cursor {
    const uint8_t* p;
    const uint8_t* end;
    bool ok = true;
    template <typename T> T get() {
        T v{};
        if ((size_t)(end - p) < sizeof(T)) { ok = false; return v; }
        memcpy(&v, p, sizeof(T));
        p += sizeof(T);
        return v;
    }
    std::string str() {
        const uint64_t n = get<uint64_t>();
        if (!ok || n > (uint64_t)(end - p)) { ok = false; return {}; }
        std::string s(reinterpret_cast<const char*>(p), (size_t)n);
        p += n;
        return s;
    }
    void skip(size_t n) {
        if ((size_t)(end - p) < n) ok = false; else p += n;
    }
};

size_t scalar_size(uint32_t t) {
    switch (t) {
        case GT_U8: case GT_I8: case GT_BOOL: return 1;
        case GT_U16: case GT_I16: return 2;
        case GT_U32: case GT_I32: case GT_F32: return 4;
        case GT_U64: case GT_I64: case GT_F64: return 8;
        default: return 0;
    }
}

// scalar / string value -> GgufValue; integers are narrowed to int like the reference (loader.cpp:197-215)
GgufValue read_value(Cursor& c, uint32_t t) {
    GgufValue v;
    switch (t) {
        case GT_U8: v.kind = GgufValue::INT; v.i = c.get<uint8_t>(); break;
        case GT_I8: v.kind = GgufValue::INT; v.i = c.get<int8_t>(); break;
        case GT_U16: v.kind = GgufValue::INT; v.i = c.get<uint16_t>(); break;
        case GT_I16: v.kind = GgufValue::INT; v.i = c.get<int16_t>(); break;
        case GT_U32: v.kind = GgufValue::INT; v.i = (int)c.get<uint32_t>(); break;
        case GT_I32: v.kind = GgufValue::INT; v.i = c.get<int32_t>(); break;
        case GT_U64: v.kind = GgufValue::INT; v.i = (int)c.get<uint64_t>(); break;
        case GT_I64: v.kind = GgufValue::INT; v.i = (int)c.get<int64_t>(); break;
        case GT_F32: v.kind = GgufValue::FLOAT; v.f = c.get<float>(); break;
        case GT_F64: v.kind = GgufValue::FLOAT; v.f = (float)c.get<double>(); break;
        case GT_BOOL: v.kind = GgufValue::BOOL; v.i = c.get<uint8_t>() != 0; break;
        case GT_STR: v.kind = GgufValue::STRING; v.s = c.str(); break;
        default: c.ok = false; break;
    }
    return v;
}

void skip_value(Cursor& c, uint32_t t, int depth = 0) {
    if (t == GT_STR) { (void)c.str(); return; }
    if (t == GT_ARR) {
        const uint32_t et = c.get<uint32_t>();
        const uint64_t n = c.get<uint64_t>();
        if (depth > 4) { c.ok = false; return; }
        if (const size_t sz = scalar_size(et)) { c.skip((size_t)n * sz); return; }
        for (uint64_t i = 0; i < n && c.ok; ++i) skip_value(c, et, depth + 1);
        return;
    }
    const size_t sz = scalar_size(t);
    if (!sz) { c.ok = false; return; }
    c.skip(sz);
}
}  // namespace

GgufFile::~GgufFile() { close(); }

void GgufFile::close() {
    if (base_) munmap(const_cast<uint8_t*>(base_), size_);
    if (fd_ >= 0) ::close(fd_);
    base_ = nullptr;
    fd_ = -1;
}

int GgufFile::open(const std::string& path) {   // reference loader.cpp:23-54: open + fstat + mmap + madvise
    close();
    path_ = path;
    fd_ = ::open(path.c_str(), O_RDONLY);
    if (fd_ < 0) { err_ = "Failed to open " + path; return NTK_E_IO; }
    struct stat st;
    if (fstat(fd_, &st) != 0 || st.st_size < 24) { err_ = "Failed to stat " + path; return NTK_E_IO; }
    size_ = (size_t)st.st_size;
    void* m = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
    if (m == MAP_FAILED) { err_ = "Failed to mmap " + path; return NTK_E_IO; }
    base_ = static_cast<const uint8_t*>(m);
    madvise(m, size_, MADV_SEQUENTIAL);
    return parse();
}

int GgufFile::parse() {   // reference loader.cpp:56-187
    Cursor c{base_, base_ + size_};
    const uint32_t magic = c.get<uint32_t>();
    if (magic != 0x46554747u) {
        char b[96];
        snprintf(b, sizeof b, "Invalid GGUF magic: 0x%08X (expected 0x46554747)", magic);
        err_ = b;
        return NTK_E_FORMAT;
    }
    version_ = c.get<uint32_t>();
    if (version_ < 2 || version_ > 3) { err_ = "Unsupported GGUF version: " + std::to_string(version_); return NTK_E_FORMAT; }
    const uint64_t n_tensors = c.get<uint64_t>();
    const uint64_t n_kv = c.get<uint64_t>();
    if (!c.ok || n_tensors > (1u << 24) || n_kv > (1u << 24)) { err_ = "Corrupt GGUF header"; return NTK_E_FORMAT; }
    fprintf(stderr, "GGUF v%u: %" PRIu64 " tensors, %" PRIu64 " metadata entries\n", version_, n_tensors, n_kv);

    for (uint64_t k = 0; k < n_kv && c.ok; ++k) {
        const std::string key = c.str();
        const uint32_t type = c.get<uint32_t>();
        if (type == GT_ARR) {   // only the three vocabulary arrays are kept (loader.cpp:102-124): merges are ignored
            const uint32_t et = c.get<uint32_t>();
            const uint64_t n = c.get<uint64_t>();
            if (!c.ok) break;
            if (key == "tokenizer.ggml.tokens" && et == GT_STR) {
                vocab_.tokens.reserve((size_t)n);
                for (uint64_t j = 0; j < n && c.ok; ++j) vocab_.tokens.push_back(c.str());
            } else if (key == "tokenizer.ggml.scores" && et == GT_F32) {
                vocab_.scores.resize((size_t)n);
                for (uint64_t j = 0; j < n && c.ok; ++j) vocab_.scores[j] = c.get<float>();
            } else if (key == "tokenizer.ggml.token_type" && et == GT_I32) {
                vocab_.token_types.resize((size_t)n);
                for (uint64_t j = 0; j < n && c.ok; ++j) vocab_.token_types[j] = c.get<int32_t>();
            } else if (const size_t sz = scalar_size(et)) {
                c.skip((size_t)n * sz);
            } else {
                for (uint64_t j = 0; j < n && c.ok; ++j) skip_value(c, et);
            }
        } else {
            meta_[key] = read_value(c, type);
        }
    }
    if (!c.ok) { err_ = "Truncated GGUF metadata"; return NTK_E_FORMAT; }

    // hyper-parameters (reference config.cpp:24-49)
    auto geti = [&](const std::string& k, int d) { auto it = meta_.find(k); return it != meta_.end() && it->second.kind == GgufValue::INT ? (int)it->second.i : d; };
    auto getf = [&](const std::string& k, float d) { auto it = meta_.find(k); return it != meta_.end() && it->second.kind == GgufValue::FLOAT ? (float)it->second.f : d; };
    auto gets = [&](const std::string& k, const char* d) { auto it = meta_.find(k); return it != meta_.end() && it->second.kind == GgufValue::STRING ? it->second.s : std::string(d); };
    ModelConfig& m = config_;
    m.architecture = gets("general.architecture", "llama");
    m.model_name = gets("general.name", "unknown");
    const std::string pre = m.architecture + ".";
    m.vocab_size = geti(pre + "vocab_size", m.vocab_size);
    m.hidden_size = geti(pre + "embedding_length", m.hidden_size);
    m.intermediate_size = geti(pre + "feed_forward_length", m.intermediate_size);
    m.n_layers = geti(pre + "block_count", m.n_layers);
    m.n_heads = geti(pre + "attention.head_count", m.n_heads);
    m.n_kv_heads = geti(pre + "attention.head_count_kv", m.n_heads);
    if (m.n_heads <= 0 || m.n_kv_heads <= 0 || m.hidden_size <= 0) { err_ = "Bad model dimensions"; return NTK_E_FORMAT; }
    m.head_dim = m.hidden_size / m.n_heads;
    m.max_seq_len = geti(pre + "context_length", m.max_seq_len);
    m.norm_eps = getf(pre + "attention.layer_norm_rms_epsilon", m.norm_eps);
    m.rope_theta = getf(pre + "rope.freq_base", m.rope_theta);
    m.bos_token_id = geti("tokenizer.ggml.bos_token_id", m.bos_token_id);
    m.eos_token_id = geti("tokenizer.ggml.eos_token_id", m.eos_token_id);
    if (!vocab_.tokens.empty() && (int)vocab_.tokens.size() != m.vocab_size) m.vocab_size = (int)vocab_.tokens.size();  // loader.cpp:139-141

    tensors_.resize((size_t)n_tensors);
    for (uint64_t t = 0; t < n_tensors && c.ok; ++t) {
        GgufTensor& ti = tensors_[t];
        ti.name = c.str();
        const uint32_t nd = c.get<uint32_t>();
        if (nd > 8) { c.ok = false; break; }
        ti.dims.resize(nd);
        for (uint32_t d = 0; d < nd; ++d) ti.dims[d] = (int64_t)c.get<uint64_t>();
        ti.ggml_type = c.get<uint32_t>();
        ti.offset = c.get<uint64_t>();
        ti.dtype = ggml_type_to_dtype(ti.ggml_type);
        ti.known_type = ggml_type_known(ti.ggml_type);
        const int64_t ne = ti.numel();
        if (ne < 0) { err_ = "Tensor '" + ti.name + "' has a non-positive or overflowing shape"; return NTK_E_FORMAT; }
        ti.nbytes = ntk_row_bytes(ti.dtype, ne);
        index_[ti.name] = (size_t)t;
    }
    if (!c.ok) { err_ = "Truncated GGUF tensor table"; return NTK_E_FORMAT; }

    size_t align = 32;   // loader.cpp:172-183
    if (auto it = meta_.find("general.alignment"); it != meta_.end() && it->second.kind == GgufValue::INT && it->second.i > 0)
        align = (size_t)it->second.i;
    const size_t header = (size_t)(c.p - base_);
    data_offset_ = (header + align - 1) / align * align;
    if (data_offset_ > size_) { err_ = "GGUF data section starts beyond the file"; return NTK_E_FORMAT; }
    const uint64_t data_size = size_ - data_offset_;
    for (const auto& ti : tensors_) {
        // offsets and sizes are file-controlled 64-bit values: compare without forming sums that can wrap
        if (ti.offset > data_size || (uint64_t)ti.nbytes > data_size - ti.offset) {   // the reference aborts here (loader.cpp:266-274)
            err_ = "Tensor '" + ti.name + "' extends beyond the file";
            return NTK_E_FORMAT;
        }
    }
    return NTK_OK;
}

const GgufTensor* GgufFile::find(const std::string& name) const {
    auto it = index_.find(name);
    return it == index_.end() ? nullptr : &tensors_[it->second];
}

const GgufValue* GgufFile::meta(const std::string& key) const {
    auto it = meta_.find(key);
    return it == meta_.end() ? nullptr : &it->second;
}

void GgufFile::print_info() const {   // loader.cpp:291-315
    fprintf(stderr, "=== GGUF File: %s ===\nFile size: %.2f GB\n", path_.c_str(), size_ / 1073741824.0);
    fprintf(stderr, "Tensor data: %.2f GB at offset 0x%zX\nTensors: %zu\nVocab: %zu tokens\n",
            (size_ - data_offset_) / 1073741824.0, data_offset_, tensors_.size(), vocab_.tokens.size());
}

}  // namespace nt
