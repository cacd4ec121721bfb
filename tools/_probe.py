import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from adversarial_gene_expression_amd import _dev, _lib
lib = _lib.load(); dev = _dev.device()
buf = _dev.zeros(64, dtype=torch.int64)
x = _dev.zeros(4096)
N = 200
def chain(kind):
    st = _dev.stream()
    for i in range(N):
        if kind == 'set': lib.agx_set_u64(st, buf.data_ptr(), i)
        else: lib.agx_rng_normal(st, x.data_ptr(), 4096, 1, 0)
for kind in ('set', 'rng'):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        chain(kind)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            chain(kind)
    torch.cuda.synchronize()
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        tg = e0.elapsed_time(e1) * 1e3 / N
    lib.agx_prof_spin(_dev.stream(), 20000)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); chain(kind); e1.record(); torch.cuda.synchronize()
    te = e0.elapsed_time(e1) * 1e3 / N
    print('%s: graph %.2f us/node, eager back-to-back %.2f us/launch' % (kind, tg, te))
