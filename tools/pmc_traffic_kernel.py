#!/usr/bin/env python3
"""HBM-side bytes per launch of ONE kernel (matched by a substring of its name) from two rocprofv3
--pmc passes over the same command (FETCH_SIZE pass, WRITE_SIZE pass); merged into
profiles/traffic.json under <key> (bench.py reads '<kernel class>@B<batch>').

    python tools/pmc_traffic_kernel.py <fetch_csv> <write_csv> <kernel-substring> <key>

FETCH_SIZE is doubled (gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md 'HBM');
WRITE_SIZE is taken as is; both are KiB per dispatch."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_launch(path, counter, sub):
    tot, ids = 0.0, set()
    for r in csv.DictReader(open(path)):
        if sub in r['Kernel_Name'] and r['Counter_Name'] == counter:
            tot += float(r['Counter_Value'])
            ids.add(r['Dispatch_Id'])
    return tot * 1024.0 / max(len(ids), 1), len(ids)


def main():
    fetch_csv, write_csv, sub, key = sys.argv[1:5]
    rd, n1 = per_launch(fetch_csv, 'FETCH_SIZE', sub)
    wr, n2 = per_launch(write_csv, 'WRITE_SIZE', sub)
    rd *= 2.0
    p = os.path.join(ROOT, 'profiles', 'traffic.json')
    t = json.load(open(p)) if os.path.exists(p) else {}
    t[key] = dict(hbm_bytes_per_launch=rd + wr, read_bytes=rd, write_bytes=wr, launches=[n1, n2], kernel=sub,
                  method='rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py; FETCH_SIZE x2 (gfx950), KiB -> B')
    json.dump(t, open(p, 'w'), indent=1, sort_keys=True)
    print(key, t[key])


if __name__ == '__main__':
    main()
