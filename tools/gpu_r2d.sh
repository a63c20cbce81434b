#!/bin/bash
mkdir -p gpurun_out/r02d
timeout 120 python tools/persistent_trace.py 2>&1 | grep -v "^Model\|^Free\|^Tokenizer\|^  \|^===\|^GGUF\|^Note\|^Arch\|^Name\|^Vocab\|^Layers\|^Max\|^RoPE\|^BOS\|^File\|^Tensor\|^Loading" > gpurun_out/r02d/persistent_trace_8b_q8_0.txt
tail -22 gpurun_out/r02d/persistent_trace_8b_q8_0.txt | cut -c1-200
for mix in Q8_0 Q4_K_M; do
  timeout 120 python bench.py --no-also --no-cpu-baseline --steps 64 --persistent --mix $mix 2>/dev/null > gpurun_out/r02d/bench_persistent_8b_$mix.json
  python3 -c "import sys,json; d=json.load(open('gpurun_out/r02d/bench_persistent_8b_$mix.json')); print('$mix', d['value'], 'tok/s', d['ms_per_step'], 'ms', d['config']['path'][:20])"
done
timeout 300 python -m pytest tests/test_engine_gpu.py -q -k "persistent_token_kernel or logits_match_reference_host_code" 2>&1 | tail -3
