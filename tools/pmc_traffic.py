#!/usr/bin/env python3
"""HBM traffic per launch of one kernel from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE are
collected in separate runs: they do not fit one pass, MI355X_MICROARCH.md "HBM").

    tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <kernel substring> <particles> > profiles/round1_pair_traffic.json

FETCH_SIZE / WRITE_SIZE are in KB.  gfx950 correction from the guide: FETCH_SIZE tallies the 128-byte
requests of wide coalesced reads at 64 bytes -> doubled; WRITE_SIZE is exact for 16-byte stores.
"""
import csv, json, sys


def median_kb(path, counter, needle):
    v = sorted(float(r["Counter_Value"]) for r in csv.DictReader(open(path))
               if r["Counter_Name"] == counter and needle in r["Kernel_Name"])
    return (v[len(v) // 2], len(v)) if v else (None, 0)


def main():
    fpath, wpath, needle, n = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    f, nf = median_kb(fpath, "FETCH_SIZE", needle)
    w, nw = median_kb(wpath, "WRITE_SIZE", needle)
    out = dict(kernel=needle, particles=n, fetch_size_kb_median=f, write_size_kb_median=w, launches_fetch=nf, launches_write=nw,
               correction="traffic = 2*FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE counts 128-B requests as 64 B)",
               traffic_bytes_per_launch=(2.0 * f + w) * 1024.0 if f is not None and w is not None else None)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
