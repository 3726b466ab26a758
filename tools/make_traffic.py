#!/usr/bin/env python3
"""Turns the summed FETCH_SIZE / WRITE_SIZE counters of the separate rocprofv3 --pmc passes (tools/pmc_sum.py output)
into profiles/<round>_traffic_<workload>.json, stamped with the hash of the kernel sources they were measured on:
    python tools/make_traffic.py <workload> <pmc_fetch.csv> <pmc_write.csv> <out.json> [pictures per launch]
bench.py only reports `roofline.traffic` from a file whose stamp matches the sources it runs. A launch covers the pictures
of one batch; bench.py's per-kernel figures are per picture, so it divides by `pictures_per_launch`."""
import csv, hashlib, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_sha256():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "svt-av1_amd", "csrc")
    for name in sorted(os.listdir(d)):
        with open(os.path.join(d, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()


def kernel_name(raw):
    """'void svtme::me_search_kernel<svtme::OnePicture>' -> 'svtme::me_search_kernel'"""
    return re.sub(r"<.*$", "", raw.replace("void ", "").strip())


def main():
    wl, fetch_csv, write_csv, out = sys.argv[1:5]
    per_launch = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    kernels = {}
    for path, counter, field in ((fetch_csv, "FETCH_SIZE", "fetch_kib"), (write_csv, "WRITE_SIZE", "write_kib")):
        tot, n = {}, {}
        for r in csv.DictReader(open(path)):
            if r["counter"] != counter:
                continue
            k = kernel_name(r["kernel"])
            tot[k] = tot.get(k, 0.0) + float(r["sum"])
            n[k] = n.get(k, 0) + int(r["dispatches"])
        for k in tot:
            kernels.setdefault(k, {})[field] = round(tot[k] / max(1, n[k]), 1)
    doc = {"workload": wl, "unit": "KiB per launch (rocprofv3 FETCH_SIZE / WRITE_SIZE, separate --pmc passes, one search stream)",
           "pictures_per_launch": per_launch, "source_sha256": source_sha256(), "kernels": kernels}
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps(doc["kernels"].get("svtme::me_search_kernel")))


if __name__ == "__main__":
    main()
