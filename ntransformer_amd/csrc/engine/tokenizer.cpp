// engine/tokenizer.cpp -- see tokenizer.h
#include "tokenizer.h"

#include <array>
#include <cstdio>
#include <cstdlib>
#include <limits>

namespace nt {
namespace {

// GPT-2 bytes<->unicode table (reference tokenizer.cpp:14-52): printable Latin-1 bytes map to themselves,
// the remaining 68 bytes to U+0100.. in order.
struct ByteTable {
    std::array<std::string, 256> fwd;
    std::unordered_map<std::string, unsigned char> inv;
    ByteTable() {
        int next = 0;
        for (int b = 0; b < 256; ++b) {
            const bool keep = (b >= 33 && b <= 126) || (b >= 161 && b <= 172) || (b >= 174);
            const unsigned cp = keep ? (unsigned)b : 256u + (unsigned)next++;
            std::string u;
            if (cp < 0x80) u.push_back((char)cp);
            else if (cp < 0x800) { u.push_back((char)(0xC0 | (cp >> 6))); u.push_back((char)(0x80 | (cp & 0x3F))); }
            else { u.push_back((char)(0xE0 | (cp >> 12))); u.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); u.push_back((char)(0x80 | (cp & 0x3F))); }
            fwd[b] = u;
            inv[u] = (unsigned char)b;
        }
    }
};
const ByteTable& table() { static const ByteTable t; return t; }

int utf8_len(unsigned char c) {
    if (c < 0x80) return 1;
    if ((c & 0xE0) == 0xC0) return 2;
    if ((c & 0xF0) == 0xE0) return 3;
    if ((c & 0xF8) == 0xF0) return 4;
    return 0;
}
}  // namespace

void Tokenizer::init(const GgufVocab& v, int bos_id, int eos_id) {
    pieces_ = v.tokens;
    scores_ = v.scores;
    types_ = v.token_types;
    bos_ = bos_id;
    eos_ = eos_id;
    ids_.clear();
    ids_.reserve(pieces_.size() * 2);
    for (int i = 0; i < (int)pieces_.size(); ++i) ids_[pieces_[i]] = i;   // later duplicates win, as in the reference
    gpt2_ = ids_.count(table().fwd[0x20]) != 0;                          // "Ġ" present -> GPT-2 byte BPE (tokenizer.cpp:77-81)
    fprintf(stderr, "Tokenizer: %d tokens, BOS=%d, EOS=%d, encoding=%s\n", (int)pieces_.size(), bos_, eos_,
            gpt2_ ? "GPT2-BPE" : "SentencePiece");
}

int Tokenizer::lookup(const std::string& s) const {
    auto it = ids_.find(s);
    return it == ids_.end() ? -1 : it->second;
}

int Tokenizer::byte_fallback(unsigned char b) const {   // tokenizer.cpp:355-372
    if (gpt2_) {
        const int id = lookup(table().fwd[b]);
        if (id >= 0) return id;
    }
    char name[8];
    snprintf(name, sizeof name, "<0x%02X>", b);
    const int id = lookup(name);
    return id >= 0 ? id : 0;
}

std::vector<int> Tokenizer::encode(const std::string& text, bool add_bos) const {
    std::vector<int> out;
    if (add_bos) out.push_back(bos_);
    if (text.empty()) return out;

    // 1. text -> the vocabulary's surface form
    std::string enc;
    if (gpt2_) {
        for (unsigned char ch : text) enc += table().fwd[ch];
    } else {
        for (char ch : text) {
            if (ch == ' ') enc += "\xE2\x96\x81"; else enc.push_back(ch);
        }
    }

    // 2. greedy longest match (window 64 bytes), byte fallback for unmatched bytes (tokenizer.cpp:132-175)
    struct Piece { int id; std::string text; int next; };
    std::vector<Piece> ps;
    for (size_t pos = 0; pos < enc.size();) {
        size_t take = 0;
        int id = -1;
        for (size_t len = std::min<size_t>(enc.size() - pos, 64); len >= 1; --len) {
            id = lookup(enc.substr(pos, len));
            if (id >= 0) { take = len; break; }
        }
        if (id < 0) { id = byte_fallback((unsigned char)enc[pos]); take = 1; }
        if (!ps.empty()) ps.back().next = (int)ps.size();
        ps.push_back({id, enc.substr(pos, take), -1});
        pos += take;
    }

    // 3. repeatedly merge the adjacent pair whose concatenation is the best-scored vocabulary entry;
    //    first pair wins ties; no scores -> every merge scores 0 (tokenizer.cpp:177-209)
    for (;;) {
        float best = -std::numeric_limits<float>::infinity();
        int at = -1;
        for (int i = 0; i < (int)ps.size(); ++i) {
            if (ps[i].next < 0 || ps[i].id < 0) continue;
            const int m = lookup(ps[i].text + ps[ps[i].next].text);
            if (m < 0) continue;
            const float sc = m < (int)scores_.size() ? scores_[m] : 0.0f;
            if (sc > best) { best = sc; at = i; }
        }
        if (at < 0) break;
        Piece& a = ps[at];
        Piece& b = ps[a.next];
        a.text += b.text;
        a.id = lookup(a.text);
        a.next = b.next;
        b.id = -1;
    }
    for (const Piece& p : ps)
        if (p.id >= 0) out.push_back(p.id);
    return out;
}

std::string Tokenizer::decode(const std::vector<int>& ids) const {
    std::string s;
    for (int id : ids) s += decode_token(id);
    return s;
}

std::string Tokenizer::decode_token(int id) const {   // tokenizer.cpp:284-353
    if (id < 0 || id >= (int)pieces_.size()) return "";
    if (id < (int)types_.size() && (types_[id] == 3 || types_[id] == 4)) return "";   // control / user-defined pieces print nothing
    const std::string& tok = pieces_[id];
    std::string out;
    if (gpt2_) {
        for (size_t pos = 0; pos < tok.size();) {
            const int n = utf8_len((unsigned char)tok[pos]);
            if (n == 0) { ++pos; continue; }
            if (pos + n > tok.size()) break;
            const std::string ch = tok.substr(pos, n);
            auto it = table().inv.find(ch);
            if (it != table().inv.end()) out.push_back((char)it->second); else out += ch;
            pos += n;
        }
        return out;
    }
    if (tok.size() == 6 && tok[0] == '<' && tok[1] == '0' && tok[2] == 'x' && tok[5] == '>') {
        const char hex[3] = {tok[3], tok[4], 0};
        return std::string(1, (char)strtol(hex, nullptr, 16));
    }
    for (size_t pos = 0; pos < tok.size();) {
        if (pos + 2 < tok.size() && (unsigned char)tok[pos] == 0xE2 && (unsigned char)tok[pos + 1] == 0x96 && (unsigned char)tok[pos + 2] == 0x81) {
            out.push_back(' ');
            pos += 3;
        } else {
            out.push_back(tok[pos++]);
        }
    }
    return out;
}

}  // namespace nt
