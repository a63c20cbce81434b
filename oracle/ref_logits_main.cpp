// oracle/ref_logits_main.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Driver around the reference's own, unmodified host classes (nt::Transformer from
// /root/reference/src/model/transformer.h, compiled from where it lies by oracle/Makefile) linked with
// the CPU restatement of its kernels (oracle/ref_launchers_cpu.cpp).  It exists because the reference
// CLI only prints text: parity needs the logits.  Output is the golden side of the logits tests.
//
//   ref_logits <model.gguf> <ctx> <out.bin> <n_prompt> <n_greedy> tok0 tok1 ...
//
// The first n_prompt ids are prefetched as one prefill call (Transformer::forward(tokens, n, 0),
// reference src/inference/engine.cpp:70-73); every further id given on the command line is fed
// teacher-forced as one decode step (engine.cpp:109); then n_greedy more steps continue with
// first-max argmax (reference src/inference/sampler.cpp:18-28).  out.bin layout (little endian):
//   int32 n_steps, int32 vocab, then n_steps x { int32 token_fed_last, int32 argmax, float32 logits[vocab] }.
#include "model/transformer.h"
#include "core/device.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

static int argmax_first(const float* l, int n) {
    int best = 0;
    for (int i = 1; i < n; ++i) if (l[i] > l[best]) best = i;
    return best;
}

int main(int argc, char** argv) {
    if (argc < 7) {
        fprintf(stderr, "usage: %s model.gguf ctx out.bin n_prompt n_greedy tok...\n", argv[0]);
        return 2;
    }
    const std::string path = argv[1];
    const int ctx = atoi(argv[2]);
    const char* out_path = argv[3];
    const int n_prompt = atoi(argv[4]);
    const int n_greedy = atoi(argv[5]);
    std::vector<int> toks;
    for (int i = 6; i < argc; ++i) toks.push_back(atoi(argv[i]));
    if (n_prompt < 1 || (int)toks.size() < n_prompt) { fprintf(stderr, "need >= n_prompt tokens\n"); return 2; }

    nt::Transformer model;
    if (!model.load(path, ctx, false)) { fprintf(stderr, "load failed\n"); return 1; }
    const int V = model.config().vocab_size;

    FILE* f = fopen(out_path, "wb");
    if (!f) { perror("fopen"); return 1; }
    const int n_steps = 1 + ((int)toks.size() - n_prompt) + n_greedy;
    fwrite(&n_steps, 4, 1, f);
    fwrite(&V, 4, 1, f);

    std::vector<float> host(V);
    auto dump = [&](const float* dev_logits, int last_tok) {
        nt_cuda_memcpy_d2h(host.data(), dev_logits, (size_t)V * sizeof(float));
        const int am = argmax_first(host.data(), V);
        fwrite(&last_tok, 4, 1, f);
        fwrite(&am, 4, 1, f);
        fwrite(host.data(), sizeof(float), V, f);
        return am;
    };

    float* lg = model.forward(toks.data(), n_prompt, 0);
    int next = dump(lg, toks[n_prompt - 1]);
    int pos = n_prompt;
    for (size_t i = n_prompt; i < toks.size(); ++i) {
        int t = toks[i];
        lg = model.forward(&t, 1, pos++);
        next = dump(lg, t);
    }
    for (int g = 0; g < n_greedy; ++g) {
        int t = next;
        lg = model.forward(&t, 1, pos++);
        next = dump(lg, t);
    }
    fclose(f);
    return 0;
}
