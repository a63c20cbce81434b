// oracle/ref_host_main.cpp -- TEST INFRASTRUCTURE ONLY.  Golden generator for the HOST logic next to the
// hot path: drives the reference's own unmodified GGUFLoader / Tokenizer / Sampler classes (compiled from
// /root/reference by oracle/Makefile) and prints what they produce.
//   ref_host tok    <model.gguf> <text-file>      one line per input line: ids separated by spaces
//   ref_host detok  <model.gguf> id id ...        decoded bytes
//   ref_host sample <logits.f32> n temp top_k top_p penalty window seed n_draws [recent ids...]
#include "model/loader.h"
#include "inference/tokenizer.h"
#include "inference/sampler.h"
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    const std::string mode = argv[1];
    if (mode == "tok" || mode == "detok") {
        nt::GGUFLoader loader;
        if (!loader.load(argv[2])) return 1;
        nt::Tokenizer tk;
        tk.init(loader.vocab(), loader.config().bos_token_id, loader.config().eos_token_id);
        if (mode == "tok") {
            std::ifstream in(argv[3]);
            std::string line;
            while (std::getline(in, line)) {
                for (int id : tk.encode(line, true)) printf("%d ", id);
                printf("\n");
            }
        } else {
            std::vector<int> ids;
            for (int i = 3; i < argc; ++i) ids.push_back(atoi(argv[i]));
            const std::string s = tk.decode(ids);
            fwrite(s.data(), 1, s.size(), stdout);
        }
        return 0;
    }
    if (mode == "sample" && argc >= 11) {
        const int n = atoi(argv[3]);
        std::vector<float> logits(n);
        FILE* f = fopen(argv[2], "rb");
        if (!f || fread(logits.data(), 4, n, f) != (size_t)n) return 1;
        fclose(f);
        nt::SamplerConfig c;
        c.temperature = (float)atof(argv[4]); c.top_k = atoi(argv[5]); c.top_p = (float)atof(argv[6]);
        c.repeat_penalty = (float)atof(argv[7]); c.repeat_window = atoi(argv[8]); c.seed = strtoull(argv[9], nullptr, 10);
        const int draws = atoi(argv[10]);
        std::vector<int> recent;
        for (int i = 11; i < argc; ++i) recent.push_back(atoi(argv[i]));
        nt::Sampler s;
        s.init(c);
        for (int d = 0; d < draws; ++d) {
            std::vector<float> l = logits;
            s.apply_repeat_penalty(l.data(), n, recent);
            const int t = s.sample(l.data(), n);
            recent.push_back(t);
            printf("%d ", t);
        }
        printf("\n");
        return 0;
    }
    return 2;
}
