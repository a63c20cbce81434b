#!/usr/bin/env python3
"""Book-keeping check of the committed evidence (NOT a product test: it cannot fail on a code regression, which is why it
lives here and not under tests/): every profiles/ file DESIGN.md cites exists, and each committed bench line's own numbers are
mutually consistent (value == steps / time, fractions == bytes x rate / peak)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")
bad = 0
for name in sorted(set(re.findall(r"profiles/([A-Za-z0-9_.\-]+)", open(os.path.join(ROOT, "DESIGN.md")).read()))):
    if not os.path.exists(os.path.join(PROF, name)) and not name.endswith("_"):
        print("DESIGN.md cites a missing file: profiles/" + name)
        bad += 1
for name in sorted(os.listdir(PROF)):
    if not (name.endswith(".json") and "bench" in name):
        continue
    try:
        b = json.load(open(os.path.join(PROF, name)))
    except ValueError:
        continue
    if not isinstance(b, dict) or "value" not in b:
        continue
    v = b["n_gpus"] * 1e3 / b["ms_per_step"]
    if abs(v / b["value"] - 1) > 2e-3:
        print(name, "value vs ms_per_step:", b["value"], v)
        bad += 1
    r = b.get("roofline", {})
    if r and abs(r["achieved"] / r["peak"] - r["frac"]) > 2e-3:
        print(name, "roofline frac inconsistent")
        bad += 1
print("evidence ok" if not bad else "%d problems" % bad)
sys.exit(1 if bad else 0)
