#!/bin/bash
# the factored large-batch forms under every other launch-sequence knob (knob test's 'gram' shape): must agree with the default
cd $GRAFT_REPO_ROOT
O=/tmp/ffcombo; mkdir -p $O
timeout -k 10 200 python3 tests/knob_runner.py $O/def.npz gram || exit 1
for k in AGX_NO_FUSED_OPT AGX_NO_FUSED_REST AGX_NO_TN_FUSE AGX_NO_CHAIN_INPUT AGX_NO_DRAW_FUSE AGX_NO_STAGE_DRAWS AGX_NO_GEN_TAIL AGX_NO_TAIL; do
  env $k=1 timeout -k 10 200 python3 tests/knob_runner.py $O/$k.npz gram || exit 1
done
python3 - <<'PY'
import numpy as np, glob, os
O='/tmp/ffcombo/'
d=np.load(O+'def.npz')
for f in sorted(glob.glob(O+'AGX_*.npz')):
    k=np.load(f)
    for tag in ('plain','aug'):
        e=np.abs(k[tag+'_w']-d[tag+'_w'])
        print('%-20s %-5s losses %.1e  weights q99.9 %.2e max %.2e' % (os.path.basename(f)[:-4], tag, np.abs(k[tag+'_losses']-d[tag+'_losses']).max(), np.quantile(e,.999), e.max()))
PY
