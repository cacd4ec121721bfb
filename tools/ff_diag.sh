#!/bin/bash
# default / AGX_NO_TN_FUSE / AGX_NO_FF / AGX_NO_GRAM at the knob test's 'gram' shape: weights after two iterations
cd $GRAFT_REPO_ROOT
O=/tmp/ffdiag; mkdir -p $O
timeout -k 10 200 python3 tests/knob_runner.py $O/def.npz gram || exit 1
AGX_NO_TN_FUSE=1 timeout -k 10 200 python3 tests/knob_runner.py $O/notn.npz gram || exit 1
AGX_NO_FF=1 timeout -k 10 200 python3 tests/knob_runner.py $O/noff.npz gram || exit 1
AGX_NO_GRAM=1 timeout -k 10 200 python3 tests/knob_runner.py $O/nogram.npz gram || exit 1
python3 - <<'PY'
import numpy as np
O='/tmp/ffdiag/'
L={k:np.load(O+k+'.npz') for k in ('def','notn','noff','nogram')}
for tag in ('plain','aug'):
    for k in L: print(tag, k, L[k][tag+'_losses'])
    for a,b in (('def','notn'),('def','noff'),('notn','noff'),('noff','nogram'),('def','nogram')):
        e=np.abs(L[a][tag+'_w']-L[b][tag+'_w']); print(tag,a,b,'q50 %.2e q99 %.2e q99.9 %.2e max %.2e'%(np.quantile(e,.5),np.quantile(e,.99),np.quantile(e,.999),e.max()))
PY
