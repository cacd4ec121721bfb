"""Report of the per-workgroup stamps written by the -DAGX_STAMPS build of gemm_kernel (tools/stamps.py, stamps_iter.py)."""
import numpy as np


def report(st_buf, NB):
    t = st_buf.cpu().numpy().reshape(NB, 8)
    t = t[t[:, 0] > 0]
    n = len(t)
    t0 = t[:, 0].min()
    us = lambda x: x / 100.0
    ph = np.stack([t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3]], 1) / 100.0
    print('%d workgroups; kernel span (first start .. last drain) %.2f us' % (n, us(t[:, 4].max() - t0)))
    print('start skew: median %.2f  p90 %.2f  max %.2f us' % tuple(us(np.percentile(t[:, 0] - t0, q)) for q in (50, 90, 100)))
    for i, name in enumerate(['prologue (first tile staged)', 'main loop', 'epilogue issue', 'store drain']):
        print('  %-30s median %6.2f  p10 %6.2f  p90 %6.2f  max %6.2f us' % ((name,) + tuple(np.percentile(ph[:, i], q) for q in (50, 10, 90, 100))))
    print('  workgroup lifetime             median %6.2f  max %6.2f us' % (np.median(us(t[:, 4] - t[:, 0])), us((t[:, 4] - t[:, 0]).max())))
    hw, xcc = t[:, 5], t[:, 6] & 15
    cu = ((xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15))
    ids, cnt = np.unique(cu, return_counts=True)
    print('%d distinct CUs; workgroups per CU: min %d median %d max %d' % (len(ids), cnt.min(), np.median(cnt), cnt.max()))
    for c in ids[:2].tolist() + ids[-1:].tolist():
        rows = t[cu == c]; rows = rows[np.argsort(rows[:, 0])]
        print(' CU %04x:' % c)
        for r in rows:
            print('    start %6.2f | staged %6.2f | loop done %6.2f | epi issued %6.2f | drained %6.2f   tile %d' %
                  (us(r[0] - t0), us(r[1] - t0), us(r[2] - t0), us(r[3] - t0), us(r[4] - t0), r[7]))
