#!/usr/bin/env python3
"""Timeline of one train iteration (from the middle of the run) in a rocprofv3 --kernel-trace CSV.

    python tools/trace_timeline.py <kernel_trace.csv> [marker-kernel-substring]

An iteration starts at each launch of the marker kernel (default: iter_stage*, the first
node queued by agx_cgan_iter_launch).  Prints per-kernel duration, the gap to the previous
kernel's end, grid and register use, then totals (busy vs gaps).
"""
import csv
import sys


def short(name):
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    if '(' in name and not name.startswith('Cijk'):
        name = name[:name.index('(')]
    return name[:70]


def main():
    path = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else 'iter_stage'
    rows = []
    if path.endswith('.db'):       # rocprofv3 default output (rocpd sqlite)
        import sqlite3
        con = sqlite3.connect(path)
        for r in con.execute('select start, end, name, grid_x, grid_y, workgroup_x, vgpr_count, accum_vgpr_count, '
                             'lds_size from kernels'):
            rows.append((r[0], r[1], r[2], r[3] // max(1, r[5]), r[4], r[5], r[6], r[7], r[8]))
    else:
        with open(path) as f:
            for r in csv.DictReader(f):
                rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'],
                             int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), int(r['Grid_Size_Y']),
                             int(r['Workgroup_Size_X']), int(r['VGPR_Count']), int(r['Accum_VGPR_Count']),
                             int(r['LDS_Block_Size'])))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if marker in r[2]]
    if len(starts) < 2:
        print('marker not found twice; showing everything')
        lo, hi = 0, len(rows)
    else:
        mid = len(starts) // 2          # an iteration from the middle of the run: inside bench.py's timed (graph) region
        lo, hi = starts[mid], starts[mid + 1]
    it = rows[lo:hi]
    busy = gaps = 0
    prev_end = it[0][0]
    agg = {}
    for s, e, n, gx, gy, wg, vg, ag, lds in it:
        gap = s - prev_end
        print('%8.2f us  gap %6.2f  grid %5dx%-3d wg %4d vgpr %3d+%-3d lds %6d  %s'
              % ((e - s) / 1e3, gap / 1e3, gx, gy, wg, vg, ag, lds, short(n)))
        busy += e - s
        gaps += max(0, gap)
        prev_end = max(prev_end, e)
        a = agg.setdefault(short(n), [0, 0])
        a[0] += 1
        a[1] += e - s
    print('---- %d kernels, busy %.1f us, gaps %.1f us, span %.1f us'
          % (len(it), busy / 1e3, gaps / 1e3, (it[-1][1] - it[0][0]) / 1e3))
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('%9.1f us  %4d x %7.2f us  %s' % (t / 1e3, c, t / c / 1e3, n))


if __name__ == '__main__':
    main()
