#!/usr/bin/env python3
"""Per-kernel resources of the last build (svt-av1_amd/build/*.res, the compiler's kernel-resource-usage remarks):
kernel, VGPRs, AGPRs, scratch bytes per lane, waves per SIMD, LDS bytes. Exits 1 if a kernel named on the command line
(substring match) uses scratch:   tools/kernel_resources.py [no-scratch-kernel ...]"""
import glob, os, re, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "svt-av1_amd", "build")
rows, cur = [], None
for path in sorted(glob.glob(os.path.join(root, "*.res"))):
    for line in open(path):
        m = re.search(r"remark: (.*)", line)
        if not m:
            continue
        t = re.sub(r"\s*\[-Rpass-analysis=[^\]]*\]\s*$", "", m.group(1).strip())
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip(), "tu": os.path.basename(path)[:-4]}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
bad = 0
for r in rows:
    name = re.sub(r"^_ZN5svtme\d+", "", r["name"])
    print("%-16s %-70s vgpr %4s agpr %3s scratch %5s occ %2s lds %6s" % (
        r["tu"], name[:70], r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("ScratchSize [bytes/lane]", "?"),
        r.get("Occupancy [waves/SIMD]", "?"), r.get("LDS Size [bytes/block]", "?")))
    if any(s in r["name"] for s in sys.argv[1:]) and r.get("ScratchSize [bytes/lane]", "0") != "0":
        bad = 1
sys.exit(bad)
