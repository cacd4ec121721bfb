#!/usr/bin/env python3
"""Per-kernel summary of rocprofv3 --pmc passes (counter_collection CSVs).

    python tools/pmc_summary.py mfma <csv>            MFMA pipe utilisation per kernel
    python tools/pmc_summary.py traffic <fetch_csv> <write_csv>   HBM-side bytes per launch per kernel
    python tools/pmc_summary.py raw <csv>             every collected counter per launch per kernel

mfma: SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's SIMD matrix pipes (cycles); the kernel's
wall cycles are GRBM_GUI_ACTIVE / 8 (rocprofv3 sums the 8 XCDs, MI355X_MICROARCH.md 'DVFS give-back'),
so utilisation = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024 SIMDs).  The duration is End - Start of the
dispatch (serialised by the counter pass; clocks under a counter pass run lower than un-profiled).
traffic: FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B) + WRITE_SIZE, KiB -> bytes."""
import csv
import re
import sys
from collections import defaultdict


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    n = re.sub(r'\(.*', '', n)
    return n[:80]


def load(path):
    per = defaultdict(lambda: defaultdict(float))   # kernel -> counter -> sum
    cnt = defaultdict(set)
    dur = defaultdict(float)
    seen = set()
    for r in csv.DictReader(open(path)):
        k = short(r['Kernel_Name'])
        per[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[k].add(r['Dispatch_Id'])
        if r['Dispatch_Id'] not in seen:
            seen.add(r['Dispatch_Id'])
            dur[k] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    return per, {k: len(v) for k, v in cnt.items()}, dur


def main():
    mode = sys.argv[1]
    if mode == 'mfma':
        per, cnt, dur = load(sys.argv[2])
        rows = []
        for k, c in per.items():
            gui = c.get('GRBM_GUI_ACTIVE', 0.0)
            busy = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
            if gui <= 0:
                continue
            rows.append((dur[k], k, cnt[k], busy / (gui / 8.0 * 1024.0), c.get('SQ_BUSY_CYCLES', 0.0) / gui, gui / 8.0 / max(dur[k], 1) ))
        print('%-80s %6s %10s %9s %9s %8s' % ('kernel', 'calls', 'total_us', 'mfma_util', 'sq_busy', 'GHz'))
        for d, k, n, u, sb, ghz in sorted(rows, reverse=True)[:24]:
            print('%-80s %6d %10.1f %9.3f %9.3f %8.2f' % (k, n, d / 1e3, u, sb, ghz))
    elif mode == 'raw':
        per, cnt, dur = load(sys.argv[2])
        names = sorted({c for v in per.values() for c in v})
        print('%-80s %6s %10s ' % ('kernel', 'calls', 'total_us') + ' '.join('%24s' % n for n in names))
        for d, k in sorted(((dur[k], k) for k in per), reverse=True)[:24]:
            print('%-80s %6d %10.1f ' % (k, cnt[k], d / 1e3) + ' '.join('%24.0f' % (per[k].get(n, 0.0) / max(cnt[k], 1)) for n in names))
    else:
        pf, cf, _ = load(sys.argv[2])
        pw, cw, _ = load(sys.argv[3])
        print('%-80s %6s %14s %14s' % ('kernel', 'calls', 'read_B/launch', 'write_B/launch'))
        rows = []
        for k in pf:
            rd = 2.0 * 1024.0 * pf[k].get('FETCH_SIZE', 0.0) / max(cf[k], 1)
            wr = 1024.0 * pw.get(k, {}).get('WRITE_SIZE', 0.0) / max(cw.get(k, 1), 1)
            rows.append((rd + wr, k, cf[k], rd, wr))
        for _, k, n, rd, wr in sorted(rows, reverse=True)[:24]:
            print('%-80s %6d %14.0f %14.0f' % (k, n, rd, wr))


if __name__ == '__main__':
    main()
