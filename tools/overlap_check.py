#!/usr/bin/env python3
"""Share of a file's normalised code lines that also occur in any reference source file (developer check against
accidental copying; run in the build container only -- the reference tree is not shipped)."""
import glob, re, sys

def norm(line):
    line = re.sub(r"/\*.*?\*/", "", line)
    line = re.sub(r"//.*", "", line)
    return re.sub(r"\s+", "", line)

ref = set()
for f in glob.glob("/root/reference/src/*.[ch]"):
    for l in open(f, errors="ignore"):
        n = norm(l)
        if len(n) >= 8:
            ref.add(n)
for f in sys.argv[1:]:
    lines = [norm(l) for l in open(f, errors="ignore")]
    lines = [l for l in lines if len(l) >= 8]
    hit = sum(l in ref for l in lines)
    print(f"{f}: {hit}/{len(lines)} = {100.0 * hit / max(1, len(lines)):.1f} %")
