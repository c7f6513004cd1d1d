#!/usr/bin/env python3
"""Builds tests/golden/crumble1_example.json from the ONE input -> output pair the reference publishes for
transcode(): the `samtools mpileup` of a heterozygous deletion inside a TCTC / poly-A STR before and after
`crumble -9p8 -l5 -u40 -i1.0,2` (/root/reference/crumble.1:516-600).

Run in the build container (needs /root/reference); the JSON is the committed fixture, the tests read only it.

What the fixture holds
  reads[]     every read of the 35 printed columns 1488390..1488424, reconstructed from the `before` block:
              first / last printed column, whether it starts / ends inside the block (^ / $), its mapq when the
              block shows it (^X), and per printed column the cell: base letter, '*' (deleted) or base + indel
              ("-1n", "+1A"), with the printed quality
  before/after  the two blocks verbatim (data of the man page, 35 lines each)
  expected_p0 / expected_after   per read and printed column the quality character:
              expected_p0    = what `-9 -p0 -l5 -u40 -i1.0,2` must print: the ORIGINAL quality inside the kept zone
                               [1488393, 1488421], the `after` block's character ('I' = Q40, '&' = Q5) outside it
              expected_after = the `after` block (inside the zone: P-block level 8 of the kept qualities)

Not in the man page (free parameters the tests vary): everything left of column 1488390 and right of 1488424
(flank bases, qualities, read starts and ends), the mapq of reads that start before the block, the quality of the
one inserted base (+1A: mpileup prints no quality for it)."""
import json
import os
import re
import sys

REF = "/root/reference/crumble.1"
HERE = os.path.dirname(os.path.abspath(__file__))
FIRST, LAST = 1488390, 1488424


def blocks():
    lines = open(REF).read().split("\n")
    cmd = [i for i, l in enumerate(lines) if l.startswith("samtools mpileup -Q0 -B -r 1:1488390-1488424")]
    assert len(cmd) == 2
    out = []
    for i in cmd:
        blk = lines[i + 1:i + 1 + (LAST - FIRST + 1)]
        assert all(l.startswith("1       1488") for l in blk), blk[:2]
        out.append(blk)
    return out


def parse_line(line):
    f = line.split()
    pos, n, bases, quals = int(f[1]), int(f[3]), f[4], f[5]
    cells, i = [], 0
    while i < len(bases):
        c = dict(start=False, mapq=None, end=False, indel="", base=None)
        if bases[i] == "^":
            c["start"] = True
            c["mapq"] = ord(bases[i + 1]) - 33
            i += 2
        c["base"] = bases[i]
        i += 1
        m = re.match(r"[+-](\d+)", bases[i:])
        if m:
            k = int(m.group(1))
            c["indel"] = bases[i:i + len(m.group(0)) + k]
            i += len(m.group(0)) + k
        if i < len(bases) and bases[i] == "$":
            c["end"] = True
            i += 1
        cells.append(c)
    assert len(cells) == n == len(quals), (line, len(cells), n, len(quals))
    for c, q in zip(cells, quals):
        c["qual"] = q
    return pos, cells


def reconstruct(block):
    """The pileup keeps file order: reads that ended leave the list, new reads (^) join at its end."""
    reads, active = [], []
    for line in block:
        pos, cells = parse_line(line)
        starts = [c for c in cells if c["start"]]
        assert len(cells) == len(active) + len(starts) or pos == FIRST
        if pos == FIRST:
            for c in cells:
                assert not c["start"]
                reads.append(dict(first=pos, starts_inside=False, mapq=None, cells=[]))
                active.append(len(reads) - 1)
        else:
            assert all(c["start"] for c in cells[len(active):]) and not any(c["start"] for c in cells[:len(active)])
            for c in cells[len(active):]:
                reads.append(dict(first=pos, starts_inside=True, mapq=c["mapq"], cells=[]))
                active.append(len(reads) - 1)
        nxt = []
        for r, c in zip(active, cells):
            reads[r]["cells"].append(dict(base=c["base"], indel=c["indel"], qual=c["qual"]))
            reads[r]["last"] = pos
            reads[r]["ends_inside"] = c["end"]
            if not c["end"]:
                nxt.append(r)
        active = nxt
    return reads


def main():
    before, after = blocks()
    rb, ra = reconstruct(before), reconstruct(after)
    assert len(rb) == len(ra) == 14
    # crumble changes qualities only: same reads, same cells
    for x, y in zip(rb, ra):
        assert x["first"] == y["first"] and x["last"] == y["last"] and len(x["cells"]) == len(y["cells"])
        assert [(c["base"], c["indel"]) for c in x["cells"]] == [(c["base"], c["indel"]) for c in y["cells"]]
    # a deletion announced at column c ("-1n") is the '*' of column c + 1
    for r in rb:
        for k, c in enumerate(r["cells"]):
            if c["indel"].startswith("-"):
                assert r["cells"][k + 1]["base"] == "*"
            if c["base"] == "*":
                assert r["cells"][k - 1]["indel"].startswith("-1")
                # mpileup prints the NEXT base's quality under a '*' (htslib: qpos of a deletion = next base)
                assert c["qual"] == r["cells"][k + 1]["qual"]
    zone = (1488393, 1488421)
    exp_p0, exp_after = [], []
    for x, y in zip(rb, ra):
        p0 = ""
        for k, (cb, caf) in enumerate(zip(x["cells"], y["cells"])):
            pos = x["first"] + k
            if zone[0] <= pos <= zone[1]:
                p0 += cb["qual"]
            else:
                assert caf["qual"] in "I&", (pos, caf)
                p0 += caf["qual"]
        exp_p0.append(p0)
        exp_after.append("".join(c["qual"] for c in y["cells"]))
    json.dump(dict(source="jkbonfield/crumble crumble.1 (EXAMPLES): samtools mpileup -Q0 -B -r 1:1488390-1488424 before / after "
                          "crumble -9p8 -l5 -u40 -i1.0,2",
                   first=FIRST, last=LAST, zone=list(zone), reads=rb, expected_p0=exp_p0, expected_after=exp_after,
                   before=before, after=after),
              open(os.path.join(HERE, "crumble1_example.json"), "w"), indent=1)
    print("reads", len(rb), "cells", sum(len(r["cells"]) for r in rb))


if __name__ == "__main__":
    sys.exit(main())
