#!/usr/bin/env python3
"""HBM traffic per launch of the per-step kernels from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE are
collected in separate runs: they do not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots").

    tools/pmc_traffic_kernels.py <fetch counter_collection.csv> <write counter_collection.csv> <particles> > profiles/round2_kernel_traffic.json

FETCH_SIZE / WRITE_SIZE are in KB.  gfx950 correction from the guide ("HBM"): FETCH_SIZE tallies the 128-byte
requests of wide coalesced reads at 64 bytes -> doubled; WRITE_SIZE is exact for 16-byte stores.
k_rebuild_fused runs every step but rebuilds only on some: its rebuilding launches are the ones above 10x the
median write size (the deciding launches write a few bytes)."""
import csv, json, sys

KERNELS = {"k_pair_tiles": "k_pair_tiles", "k_integrate": "k_integrate<float, 3", "k_rebuild_fused": "k_rebuild_fused", "k_bonded_work": "k_bonded_work"}


def values(path, counter, needle):
    return [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and needle in r["Kernel_Name"]]


def med(v):
    v = sorted(v)
    return v[len(v) // 2] if v else None


def main():
    fpath, wpath, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
    out = dict(particles=n, correction="traffic = 2*FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE counts 128-B requests as 64 B); KB -> bytes",
               traffic_bytes_per_launch={}, detail={})
    for key, needle in KERNELS.items():
        f, w = values(fpath, "FETCH_SIZE", needle), values(wpath, "WRITE_SIZE", needle)
        if not f or not w:
            continue
        if key == "k_rebuild_fused":     # keep the launches that rebuilt
            cut_w = 10.0 * max(med(w), 1.0)
            cut_f = 10.0 * max(med(f), 1.0)
            f, w = [x for x in f if x > cut_f], [x for x in w if x > cut_w]
            if not f or not w:
                continue
        fm, wm = med(f), med(w)
        out["traffic_bytes_per_launch"][key] = (2.0 * fm + wm) * 1024.0
        out["detail"][key] = dict(fetch_size_kb_median=fm, write_size_kb_median=wm, launches_fetch=len(f), launches_write=len(w))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
