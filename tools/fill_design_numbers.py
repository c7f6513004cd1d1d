#!/usr/bin/env python3
"""Regenerates DESIGN.md section 8 from tools/design_section8.tmpl.md and profiles/<tag>_bench_*.json (round-end
bookkeeping: the numbers in the text come from the committed bench lines, not from memory).
Usage: python tools/fill_design_numbers.py r03f"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
def line(name):
    return json.loads(open(os.path.join(ROOT, "profiles", f"{tag}_bench_{name}.json")).read().strip().splitlines()[-1])
c20, c9, pn, hf = line("chr20_full"), line("chr20_9"), line("panel"), line("hifi_config4")
def km(d): return d["roofline"]["kernels_ms"]
def f(x, n=3): return f"{x:.{n}f}"
r = {}
k = km(c20); a = c20["around_the_step"]
r.update(C20MS=f(c20["ms_per_step"], 2), C20V=f(c20["value"], 0), C20K=f(k["k_columns2"]), C20R=f(k["k_replay"]), C20S=f(k["k_str"]),
         C20T=f(k["k_stats"]), C20O=f(k["k_overdepth"]), C20N=f(k["scan"], 2), C20F=f(k["k_fold"]), C20E=f(k["k_emit"], 2),
         C20A=f(c20["roofline"]["achieved"], 0), C20P=f(100 * c20["roofline"]["frac"], 1), C20W=f(100 * c20["roofline"]["pipeline_frac"], 1),
         C20P0=f(a["p0_ms"], 1), C20H=f(a["h2d_ms"], 0), C20D=f(a["d2h_ms"], 0), C20D1=f(a["d2h_first_ms"], 0),
         C20IP=f(a["value_incl_p0"], 0), C20IX=f(a["value_incl_pcie"], 1),
         CPU1=f(c20["cpu_baseline"]["value"], 3), CPU16=f(c20["cpu_baseline_all"]["value"], 2))
k = km(c9)
r.update(C9MS=f(c9["ms_per_step"], 1), C9V=f(c9["value"], 0), C9W=f(100 * c9["roofline"]["pipeline_frac"], 1), C9K=f(k["k_columns2"]),
         C9B=f(k["k_pblock"]), C9R=f(k["k_replay"]))
k = km(pn)
r.update(PMS=f(pn["ms_per_step"], 1), PV=f(pn["value"], 0), PW=f(100 * pn["roofline"]["pipeline_frac"], 1), PK=f(k["k_columns2"]),
         PR=f(k["k_replay"]), PB=f(k["k_pblock"]))
k = km(hf); a = hf["around_the_step"]
r.update(HMS=f(hf["ms_per_step"], 1), HV=f(hf["value"], 0), HW=f(100 * hf["roofline"]["pipeline_frac"], 1), HK=f(k["k_columns2"]),
         HR=f(k["k_replay"]), HB=f(k["k_pblock"]), HS=f(k["k_str"]), HP0=f(a["p0_ms"], 1), HIP=f(a["value_incl_p0"], 0))
w = c20["other_configs"]["config5_wgs_streamed"]
r.update(WMS=f(w["ms_per_pass"], 0), WV=f(w["value_incl_pcie"], 1), WS1=f(w["stages_ms"]["h2d_and_p0"], 0), WS2=f(w["stages_ms"]["kernels"], 0),
         WS3=f(w["stages_ms"]["d2h"], 0), WOV=f(w["overlap_wall_over_sum_of_stages"], 2))
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
# section 8 is regenerated from its template (tools/design_section8.tmpl.md) every time
a, b = s.index("## 8. Measured (MI355X;"), s.index("## 9. Host ingest / egress")
s = s[:a] + open(os.path.join(ROOT, "tools", "design_section8.tmpl.md")).read().replace("@TAG@", tag) + s[b:]
for key, val in r.items():
    s = s.replace(f"@{key}@", val)
open(p, "w").write(s)
left = [t for t in set(__import__("re").findall(r"@[A-Z0-9]+@", s))]
print("filled", len(r), "left", left)
for key in ("C20MS", "C20V", "C20K", "C20P", "HMS", "HV", "PMS", "PV", "C9MS", "WV", "C20IX"):
    print(key, r[key])
