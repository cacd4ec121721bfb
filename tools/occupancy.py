import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adversarial_gene_expression_amd import _dev, _lib
lib = _lib.load(); _dev.device()
for w, name in enumerate(['64x64 NT', '64x64 TN fused opt BK16', '128x128 NN 8-wave', '128x128 SYRK', '64x64 NN', '128x128 NN 4-wave']):
    nb, lds = C.c_int32(), C.c_int32()
    _lib.check(lib.agx_prof_occupancy(w, C.byref(nb), C.byref(lds)))
    print('%-20s resident workgroups per CU %d  (dynamic LDS %d B)' % (name, nb.value, lds.value))
