"""Per-buffer comparison of two workspace dumps of scripts/dump_unet.py (layout of unet.hip: make_layout)."""
import numpy as np, sys
r = int(sys.argv[1]); A, B = sys.argv[2], sys.argv[3]
a=np.load("gpurun_out/dump_%s_%d.npz" % (A, r)); b=np.load("gpurun_out/dump_%s_%d.npz" % (B, r))
print("====", A, "vs", B)
for k in ("pred", "gx"):
    d = np.abs(a[k] - b[k]); print(r, k, "max abs diff", d.max(), "ndiff", int((a[k] != b[k]).sum()), "of", a[k].size)
wa,wb=a["ws"].view(np.float32),b["ws"].view(np.float32)
bufs=[("cat1",6656),("cat2",6656),("ds0",1664),("ds1",1664),("up1",3328),("f",3328),("yf",3328),("sf",16)]
blocks=[(64,52,1),(64,52,0),(128,26,1),(128,26,0),(256,13,1),(256,13,0),(256,13,0),(256,13,0),(128,13,1),(128,13,0),(64,26,1),(64,26,0)]
for i,(c,L,hr) in enumerate(blocks):
    bufs += [("b%d.h1"%i,c*L),("b%d.out"%i,c*L)] + ([("b%d.res"%i,c*L)] if hr else []) + [("b%d.y1"%i,c*L),("b%d.y2"%i,c*L),("b%d.s1"%i,16),("b%d.s2"%i,16)]
o=0
for name,fl in bufs:
    n=fl*r
    x=wa[o:o+n]; y=wb[o:o+n]
    nd=(x.view(np.uint32)!=y.view(np.uint32))
    if nd.any():
        d=np.abs(x-y)
        per=nd.reshape(r,-1).sum(1)
        print("%-8s diff share %.3f max abs %.3g  (|val| mean %.3g) rows with diffs %d, first rows %s" % (name, nd.mean(), np.nanmax(d), np.abs(y).mean(), int((per>0).sum()), list(np.flatnonzero(per>0)[:6])))
    o+=(n+63)//64*64
