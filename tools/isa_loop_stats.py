#!/usr/bin/env python3
"""Instruction mix of the hottest loop of a kernel in hipcc's -S output (tuning aid, not part of the product).

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o /tmp/gemv.s ntransformer_amd/csrc/gemv.hip
  python tools/isa_loop_stats.py /tmp/gemv.s 'gemv_quant_kernelILi4ELb0ELb1'

Prints, for every loop (label ... backward branch to it) in the kernel: length and a VALU / SALU / DS / VMEM / wait
breakdown, so the cost of a decode change can be read off before a GPU run."""
import re, sys, collections

def classify(op):
    if op.startswith(('v_',)): return 'valu'
    if op.startswith(('ds_',)): return 'ds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'vmem'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith(('s_barrier',)): return 'barrier'
    if op.startswith(('s_cbranch', 's_branch')): return 'branch'
    if op.startswith('s_'): return 'salu'
    return 'other'

def main():
    path, pat = sys.argv[1], sys.argv[2]
    lines = open(path).read().split('\n')
    start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\w*' + re.escape(pat) + r'\w*:', l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm'))
    body = lines[start:end + 1]
    labels = {}
    insns = []   # (index, op, text)
    for l in body:
        t = l.strip()
        m = re.match(r'^(\.LBB\w+):', t)
        if m: labels[m.group(1)] = len(insns); continue
        if not t or t.startswith((';', '.', '//')) or t.endswith(':'): continue
        insns.append((t.split()[0], t))
    print(f'{pat}: {len(insns)} instructions')
    loops = []
    for i, (op, t) in enumerate(insns):
        if op.startswith(('s_cbranch', 's_branch')):
            tgt = t.split()[-1]
            if tgt in labels and labels[tgt] <= i: loops.append((labels[tgt], i, tgt))
    for a, b, tgt in sorted(loops, key=lambda x: x[0] - x[1]):
        c = collections.Counter(classify(op) for op, _ in insns[a:b + 1])
        ops = collections.Counter(op for op, _ in insns[a:b + 1] if classify(op) == 'valu')
        print(f'  loop {tgt}: {b - a + 1:5d} insns  ' + ' '.join(f'{k}={v}' for k, v in sorted(c.items())))
        if '-v' in sys.argv: print('     ' + ', '.join(f'{k}:{v}' for k, v in ops.most_common(25)))

if __name__ == '__main__':
    main()
