"""agx_colstats_noalloc: the one entry point that allocates nothing (static device scratch) against
agx_colstats (caller workspace), through the C-ABI: same kernels and chunking, so mean / std must be
bit-identical; two calls in a row, the second with a smaller input; sizes beyond the static scratch
are rejected with a clear error.  No counterpart in the reference (see include/agx.h)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(lib, _lib, torch, x, N, G):
    ld = x.stride(0)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    m0, s0 = torch.full((G,), -7.0, device='cuda'), torch.full((G,), -7.0, device='cuda')
    m1, s1 = torch.full((G,), -9.0, device='cuda'), torch.full((G,), -9.0, device='cuda')
    nws = int(lib.agx_colstats_ws_doubles(N, G))
    ws = torch.full((nws,), float('nan'), dtype=torch.float64, device='cuda')
    _lib.check(lib.agx_colstats(st, p(x), ld, N, G, p(m0), p(s0), p(ws), nws), 'agx_colstats')
    _lib.check(lib.agx_colstats_noalloc(st, p(x), ld, N, G, p(m1), p(s1)), 'agx_colstats_noalloc')
    torch.cuda.synchronize()
    return m0, s0, m1, s1


def test_no_alloc_colstats_matches_workspace_variant(gpu):
    import torch
    from adversarial_gene_expression_amd import _lib
    lib = _lib.load()
    gen = torch.Generator(device='cuda').manual_seed(3)
    # first call: GTEx evaluation shape (4 row chunks, 145232 doubles of scratch); second: smaller
    for N, G, ld in ((2287, 18154, 18156), (227, 1076, 1076), (5, 3, 4)):
        x = torch.randn(N, ld, device='cuda', generator=gen) * 2.0 + 5.0
        m0, s0, m1, s1 = _both(lib, _lib, torch, x, N, G)
        assert torch.equal(m0, m1) and torch.equal(s0, s1)
        ref = x[:, :G].double()
        np.testing.assert_allclose(m1.cpu().numpy(), ref.mean(0).cpu().numpy(), rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(s1.cpu().numpy(), ref.std(0, unbiased=False).cpu().numpy(), rtol=1e-6)
    # no device memory is taken or returned by the call, and it can be recorded into a graph (the address of the
    # static scratch was resolved by agx_init): replaying the graph gives the same statistics
    N, G, ld = 227, 1076, 1076
    x = torch.randn(N, ld, device='cuda', generator=gen) * 2.0 + 5.0
    m0, s0, m1, s1 = _both(lib, _lib, torch, x, N, G)
    p = lambda t: C.c_void_p(t.data_ptr())
    mg, sg = torch.zeros(G, device='cuda'), torch.zeros(G, device='cuda')
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    _lib.check(lib.agx_colstats_noalloc(C.c_void_p(torch.cuda.current_stream().cuda_stream), p(x), ld, N, G, p(mg), p(sg)))
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0
    assert torch.equal(mg, m0) and torch.equal(sg, s0)
    mg.zero_()
    sg.zero_()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        _lib.check(lib.agx_colstats_noalloc(C.c_void_p(torch.cuda.current_stream().cuda_stream), p(x), ld, N, G, p(mg), p(sg)))
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(mg, m0) and torch.equal(sg, s0)
    # beyond the static scratch: rejected before any launch, with a message that names the limit
    x = torch.zeros(2, 8, device='cuda')
    big = 65537
    rc = lib.agx_colstats_noalloc(None, C.c_void_p(x.data_ptr()), big, 2, big, C.c_void_p(x.data_ptr()),
                                  C.c_void_p(x.data_ptr()))
    assert rc == -1 and b'static scratch' in lib.agx_last_error()
