#!/usr/bin/env python3
"""HBM traffic per launch of one GEMM class from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

    python tools/pmc_traffic.py <fetch_csv> <write_csv> <key> [launches_per_class]

Counters are per dispatch in KiB; FETCH_SIZE is doubled (gfx950 tallies 128-B read requests at 64 B,
MI355X_MICROARCH.md 'HBM'); WRITE_SIZE is taken as is.  A class = main kernel + split-K reduce:
all dispatches of the run whose kernel name contains 'gemm' or 'splitk' are summed and divided by
the number of main-kernel dispatches.  Result is merged into profiles/traffic.json under <key>."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def total(path, counter):
    s, mains = 0.0, 0
    seen = set()
    with open(path) as f:
        for r in csv.DictReader(f):
            n = r['Kernel_Name']
            if 'gemm' not in n and 'splitk' not in n:
                continue
            if r['Counter_Name'] != counter:
                continue
            s += float(r['Counter_Value'])
            if 'splitk' not in n and r['Dispatch_Id'] not in seen:
                seen.add(r['Dispatch_Id'])
                mains += 1
    return s, mains


def main():
    fetch_csv, write_csv, key = sys.argv[1:4]
    f, n1 = total(fetch_csv, 'FETCH_SIZE')
    w, n2 = total(write_csv, 'WRITE_SIZE')
    rd = 2.0 * f * 1024.0 / max(n1, 1)
    wr = w * 1024.0 / max(n2, 1)
    p = os.path.join(ROOT, 'profiles', 'traffic.json')
    t = json.load(open(p)) if os.path.exists(p) else {}
    t[key] = dict(hbm_bytes_per_launch=rd + wr, read_bytes=rd, write_bytes=wr, launches=[n1, n2],
                  method='rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE x2 (gfx950), KiB -> B')
    json.dump(t, open(p, 'w'), indent=1, sort_keys=True)
    print(key, t[key])


if __name__ == '__main__':
    main()
