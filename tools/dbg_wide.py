import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["INTFFT_DIAG"]="1"
import numpy as np, torch
from intfftk_amd import IntFFTCore
from oracle import oracle_c as C
from tests.helpers import uniform_frames
for (L,dw,tw,d) in [(13,28,16,"FWD"),(16,28,16,"FWD"),(13,28,16,"INV"),(13,32,16,"FWD")]:
    n=1<<L
    x=uniform_frames(8 if L<16 else 1,n,dw-1,5)
    core=IntFFTCore(L,dw,tw,1,0,"NEW",d)
    y=core(torch.from_numpy(x.astype(np.int32)).cuda()).cpu().numpy()
    want=C.execute(x,C.make_params(L,dw,tw,1,0,True),{"FWD":C.FWD,"INV":C.INV}[d])
    bad=np.argwhere(y!=want)
    print(L,dw,tw,d,core.info["kernel_name"],"bad",len(bad),"of",y.size)
    if len(bad):
        print(" first", bad[:5].tolist(), [ (int(y[tuple(b)]), int(want[tuple(b)])) for b in bad[:5]])
        pos=np.unique(bad[:,1]); print(" positions bad:", len(pos), pos[:20], "frames:", np.unique(bad[:,0]))
        diff=(y-want)[tuple(bad.T)]
        print(" diff stats: min", diff.min(), "max", diff.max(), "log2|d| median", np.median(np.log2(np.abs(diff).astype(float)+1)))
