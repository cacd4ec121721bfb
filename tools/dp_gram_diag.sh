#!/bin/bash
# four-piece vs one-piece overlapped exchange at the GTEx 512-row shape: where do the weights differ?
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
O=/tmp/dpdiag; mkdir -p $O
export DP_RUNNER_ITERS=${DP_RUNNER_ITERS:-1}
AGX_DP_FORCE=1 AGX_DP_CHUNKS=0 timeout -k 10 200 python3 tests/dp_native_runner.py native gtex512 rmsprop $O/one.npz || exit 1
AGX_DP_FORCE=1 AGX_DP_CHUNKS=4 timeout -k 10 200 python3 tests/dp_native_runner.py native gtex512 rmsprop $O/four.npz || exit 1
python3 - <<'PY'
import numpy as np
O='/tmp/dpdiag/'
A=np.load(O+'four.npz'); B=np.load(O+'one.npz')
a,b=A['w'],B['w']; sizes=A['sizes']
e=np.abs(a-b)
off=0
for k,sz in enumerate(sizes):
    sub=e[off:off+sz]
    print('tensor %2d size %8d  mean %.2e  q99.9 %.2e  max %.2e  frac>1e-6 %.4f' % (k, sz, sub.mean(), np.quantile(sub,.999), sub.max(), (sub>1e-6).mean()))
    if sz == 256*18154:
        m=sub.reshape(256,18154)
        print('   W_out rows(hidden) frac>1e-6 by 64-row chunk', [float((m[i:i+64]>1e-6).mean()) for i in range(0,256,64)])
        print('   by gene eighth', [float((m[:,i:i+2270]>1e-6).mean()) for i in range(0,18154,2270)])
    off+=sz
PY
