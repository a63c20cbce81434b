#!/usr/bin/env python3
"""Prints the round's numbers table (markdown) from the evidence files under profiles/.
usage: python tools/round_table.py r04"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def line(name):
    try:
        return json.loads(open(os.path.join(P, name)).read().strip().splitlines()[-1])
    except (OSError, ValueError, IndexError):
        return None


def main():
    r = sys.argv[1] if len(sys.argv) > 1 else "r04"
    d = line("%s_bench_default_8b_q8_0_with_also.json" % r)
    drv = line("%s_bench_driver_style_n1_steps20.json" % r)
    pmc = json.load(open(os.path.join(P, "pmc_traffic.json")))
    also = {a["k"]: a for a in d["config"]["also"]}
    rows = [("8b_q8_0", "**8B Q8_0 (headline)**"), ("8b_q4_k_m", "8B Q4_K_M"), ("70b_q4_k_m", "70B Q4_K_M"), ("70b_q6_k", "70B Q6_K"),
            ("8b_q8_0_ctx3900", "8B Q8_0 behind a 3900-token prompt")]
    print("| workload | tokens/s in the default line (steps) | dedicated run (steps) | end-to-end `B_tok`·rate ÷ 8 TB/s | GEMV launches: live events | GEMV launches: kernel trace | Σ kernel time per token, profiled (clock) vs un-profiled step | PMC traffic ÷ algorithmic |")
    print("|---|---|---|---|---|---|---|---|")
    for k, label in rows:
        b = line("%s_bench_%s.json" % (r, k))
        pb = line("%s_bench_%s_profiled.json" % (r, k))
        if k == "8b_q8_0":
            v0, ms0, st0, fr0, gf0 = d["value"], d["ms_per_step"], d["steps"], d["hbm_fraction_of_8TBs_end_to_end"], d["roofline"]["frac"]
            extra = "; driver-style %d steps: %.1f" % (drv["steps"], drv["value"]) if drv else ""
        else:
            a = also[k]
            v0, ms0, st0, fr0, gf0 = a["value"], a["ms"], a["steps"], a["frac"], a.get("gemv_frac", a.get("roofline", {}).get("frac_events"))
            extra = ""
        tr = ""
        busy = ""
        try:
            txt = open(os.path.join(P, "%s_rocprofv3_kernel_trace_%s.txt" % (r, k))).read()
            mg = re.search(r"pooled: ([0-9.]+) launches/token, avg ([0-9.]+) us, .* = ([0-9.]+)% of 8 TB/s", txt)
            mt = re.search(r"kernel time per token: ([0-9.]+) us all kernels", txt)
            tr = "%.3f (%.2f µs × %d)" % (float(mg.group(3)) / 100, float(mg.group(2)), round(float(mg.group(1))))
            busy = "%.0f µs (%.0f MHz) vs %.0f / %.0f" % (float(mt.group(1)), pb["sclk_mhz"], 1e3 * ms0, 1e3 * b["ms_per_step"])
        except (OSError, AttributeError, TypeError):
            pass
        g = pmc.get(k, {}).get("ntk::gemv_quant_*")
        pr = "%.3f" % ((g["fetch_bytes_per_launch"] + g["write_bytes_per_launch_raw"]) / g["algorithmic_bytes_per_launch"]) if g else ""
        print("| %s | %s%.1f (%.3f ms; %d)%s%s | %s | %.3f | %.3f / %s | %s | %s | %s |"
              % (label, "**" if k == "8b_q8_0" else "", v0, ms0, st0, "**" if k == "8b_q8_0" else "", extra,
                 "%.1f (%d)" % (b["value"], b["steps"]) if b else "", fr0, gf0, "%.3f" % b["roofline"]["frac"] if b else "", tr, busy, pr))
    pp = d["config"].get("prompt_pass")
    print("prompt pass (default line): %s; also: %s" % (pp, {k: a.get("prompt_tok_s") for k, a in also.items() if a.get("prompt_tok_s")}))
    print("cpu_baseline: %s tokens/s, %s threads" % (d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"]))


if __name__ == "__main__":
    main()
