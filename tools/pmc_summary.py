#!/usr/bin/env python3
"""HBM traffic per launch from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE are collected in SEPARATE runs: the TCC
block has 4 counter slots, FETCH_SIZE takes 3 and WRITE_SIZE 2 -- MI355X_MICROARCH.md, 'rocprofv3 PMC slots').

  python tools/pmc_summary.py <fetch_counter_collection.csv> [<write_counter_collection.csv>] [--json out.json --key 8b_q8_0]

Corrections applied (same guide, 'HBM'): FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts 128-byte
requests at 64 bytes, i.e. exactly half of the bytes of a wide coalesced streaming read -> doubled here (raw value
printed beside it).  WRITE_SIZE is uncalibrated and reported raw."""
import csv, sys, json, gzip, collections, argparse

def load(path, counter):
    """kernel -> per-dispatch values, DECODE window only (as tools/prof_summary.py): dispatches after the last prefill
    attention launch, starting at the embedding lookup of the first decode token."""
    op = gzip.open if path.endswith('.gz') else open
    rows = []
    with op(path, 'rt', newline='') as f:
        for r in csv.DictReader(f):
            if r.get('Counter_Name') != counter: continue
            name = r['Kernel_Name'].split('(')[0].replace('void ', '')
            rows.append((int(r['Dispatch_Id']), name, float(r['Counter_Value'])))
    rows.sort()
    last_prefill = max((i for i, r in enumerate(rows) if 'attention_kernel<' in r[1]), default=-1)
    dec = rows[last_prefill + 1:]
    first = next((i for i, r in enumerate(dec) if 'attention_decode_fused' in r[1]), 0)
    first_embed = max((i for i in range(first) if 'embed_rows' in dec[i][1]), default=0)
    per = collections.defaultdict(list)
    for _, name, v in dec[first_embed:]:
        if 'sclk_' not in name: per[name].append(v)   # (the clock probes of bench.py are not part of the token)
    return per

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('fetch_csv'); ap.add_argument('write_csv', nargs='?')
    ap.add_argument('--json'); ap.add_argument('--key', default='default')
    ap.add_argument('--algorithmic-bytes-per-launch', type=float, default=0.0)
    a = ap.parse_args()
    fetch = load(a.fetch_csv, 'FETCH_SIZE')
    write = load(a.write_csv, 'WRITE_SIZE') if a.write_csv else {}
    out = {}
    print(f"{'kernel':60s} {'launches':>8s} {'FETCH raw MB':>13s} {'FETCH x2 MB':>12s} {'WRITE raw MB':>13s}")
    pooled = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for k in sorted(fetch, key=lambda k: -sum(fetch[k])):
        n = len(fetch[k]); fr = sum(fetch[k]) / n * 1024.0
        wr = (sum(write[k]) / len(write[k]) * 1024.0) if k in write else float('nan')
        print(f"{k[:60]:60s} {n:8d} {fr / 1e6:13.3f} {2 * fr / 1e6:12.3f} {wr / 1e6:13.3f}")
        base = k.split('<')[0]
        pooled[base][0] += n; pooled[base][1] += sum(fetch[k]) * 1024.0
        if k in write: pooled[base][2] += sum(write[k]) * 1024.0
    print()
    for base, (n, fb, wb) in pooled.items():
        print(f"{base}: {n} launches, FETCH_SIZE {fb / n / 1e6:.3f} MB raw -> {2 * fb / n / 1e6:.3f} MB corrected per launch, WRITE_SIZE {wb / n / 1e6:.4f} MB raw per launch")
        out[base] = {"launches": n, "fetch_bytes_per_launch_raw": fb / n, "fetch_bytes_per_launch": 2 * fb / n,
                     "write_bytes_per_launch_raw": wb / n}
    # every form of the GEMV (plain, integer-activation, two-format, matrix-core over the repack) pooled: what bench.py's roofline block calls
    # "the GEMV launches" (the key keeps its round-1 name)
    gk = [k for k in out if k.startswith('ntk::gemv_quant_') or k.startswith('ntk::rp_gemv_kernel')]
    if gk:
        n = sum(out[k]['launches'] for k in gk)
        out['ntk::gemv_quant_*'] = {"launches": n,
                                    "fetch_bytes_per_launch_raw": sum(out[k]['fetch_bytes_per_launch_raw'] * out[k]['launches'] for k in gk) / n,
                                    "fetch_bytes_per_launch": sum(out[k]['fetch_bytes_per_launch'] * out[k]['launches'] for k in gk) / n,
                                    "write_bytes_per_launch_raw": sum(out[k]['write_bytes_per_launch_raw'] * out[k]['launches'] for k in gk) / n}
        g = out['ntk::gemv_quant_*']
        print(f"GEMV launches (ntk::gemv_quant_* + ntk::rp_gemv_kernel) pooled: {n} launches, FETCH_SIZE {g['fetch_bytes_per_launch'] / 1e6:.3f} MB corrected per launch, WRITE_SIZE {g['write_bytes_per_launch_raw'] / 1e6:.4f} MB raw per launch")
        if a.algorithmic_bytes_per_launch:
            out['ntk::gemv_quant_*']['algorithmic_bytes_per_launch'] = a.algorithmic_bytes_per_launch
            print(f"gemv launches: corrected fetch / algorithmic = {g['fetch_bytes_per_launch'] / a.algorithmic_bytes_per_launch:.3f}")
    if a.json:
        try: cur = json.load(open(a.json))
        except Exception: cur = {}
        cur[a.key] = out
        json.dump(cur, open(a.json, 'w'), indent=1, sort_keys=True)

if __name__ == '__main__':
    main()
