#!/bin/bash
timeout 600 python -m pytest tests/test_tp_gpu.py -x -q -p no:cacheprovider 2>&1 | grep -v "^Tensor\|^Model\|^Free\|^Token\|^===\|^File\|^Vocab\|^Max\|^RoPE\|^BOS\|^Arch\|^Layers\|^Heads\|^Embed\|^FFN\|^Quant\|^Loading\|^GGUF\|^Note\|^Name\|^Hidden\|^Inter\|^Norm" | tail -40
