"""Generates tests/golden/find_str.json from the REAL reference find_STR
(/root/reference/str_finder.c compiled unmodified by oracle/Makefile into
oracle/_ref/libstrfinder_ref.so).  Run in the build container only:
    python tests/golden/make_str_golden.py
The output is data (input string, expected [start,end] list), not source.
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib as O  # noqa: E402

assert O.have_ref(), "build oracle/_ref first: make -C oracle"
rng = random.Random(20261002)
vec = []
fixed = ["AGGGGAGGAGAAGAC", "AGCTCCGTATATAGGACTGAGGTAG", "GATCGGACGAGAGAGAGAGAGTCGGAC",
         "A" * 40, "AC" * 30, "ACG" * 20, "ACGT" * 15, "ACGTA" * 12, "ACGTAC" * 10,
         "ACGTACG" * 9, "ACGTACGT" * 8, "ACGTACGTA" * 7, "N" * 20 + "ACAC", "T",
         "GATTGCTAGGCTTACACACAC", "GATTGCTAGGCTTGCAAAAA"]
for s in fixed:
    vec.append([s, O.ref_find_str(s)])
while len(vec) < 260:
    L = rng.choice([5, 14, 15, 16, 40, 100, 301, 501])
    s = ""
    while len(s) < L:
        if rng.random() < 0.4:
            unit = "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 9)))
            s += unit * rng.randint(2, 10)
        else:
            s += "".join(rng.choice("ACGTN") for _ in range(rng.randint(1, 12)))
    s = s[:L]
    vec.append([s, O.ref_find_str(s)])
with open(os.path.join(HERE, "find_str.json"), "w") as f:
    json.dump(vec, f, separators=(",", ":"))
print(len(vec), "vectors")
