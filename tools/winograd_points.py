"""Which interpolation points for Winograd F(4x4,3x3) in fp32?  Cook-Toom matrices for a point set, the 7-layer scale2.0x topology on a 78x78 plane and the
single-layer standard-normal case of tests/test_gpu_parity.py::test_layer_filter_fast_kernels, against the fp64 truth and the north_star gate
(|x - oracle32| <= 1e-4 |oracle32| + 1e-5; 'gate use' = the largest ratio, 1.0 = at the gate).  numpy only (CPU):  python tools/winograd_points.py
Result (round 3): Lavin & Gray's 0, +-1, +-2 -> 9.5e-6 of the range / gate use 1.34 (FAILS); 0, +-1/2, +-3/2 -> 3.0e-6 / 0.56 with the same operation count
(symmetric point pairs) -- the points conv3x3_wino4 uses."""
import numpy as np, sys
sys.path.insert(0,'/root/repo')
from tools import gen_model
def cook_toom(points, m=4, r=3):
    n = m + r - 1
    a = [float(p) for p in points]           # n-1 finite points, plus infinity
    AT = np.zeros((m, n)); G = np.zeros((n, r))
    for k in range(n-1):
        Nk = np.prod([a[k]-a[l] for l in range(n-1) if l != k])
        for i in range(m): AT[i, k] = a[k]**i
        for j in range(r): G[k, j] = a[k]**j / Nk
    AT[m-1, n-1] = 1.0; G[n-1, r-1] = 1.0
    BT = np.zeros((n, n))
    M = np.zeros((m*r, n))
    for i in range(m):
        for j in range(r):
            M[i*r+j, :] = AT[i, :] * G[:, j]
    for p in range(n):
        rhs = np.array([1.0 if p == i + j else 0.0 for i in range(m) for j in range(r)])
        sol, res, rk, sv = np.linalg.lstsq(M, rhs, rcond=None)
        BT[:, p] = sol
    # check
    rng = np.random.default_rng(0)
    g = rng.standard_normal(r); d = rng.standard_normal(n)
    y = AT @ ((G @ g) * (BT @ d))
    want = np.array([sum(g[j]*d[i+j] for j in range(r)) for i in range(m)])
    assert np.allclose(y, want, atol=1e-9), (y, want)
    return AT, G, BT
def leaky(x): return np.where(x>0,x,x*x.dtype.type(0.1))
def direct(x,w,b,dt):
    C,H,Wd=x.shape; O=w.shape[0]
    x=x.astype(dt); w=w.astype(dt)
    out=np.zeros((O,H-2,Wd-2),dt)
    for r in range(3):
        for c in range(3):
            out+=np.einsum('oc,chw->ohw',w[:,:,r,c],x[:,r:r+H-2,c:c+Wd-2]).astype(dt)
    return leaky(out+b.astype(dt)[:,None,None])
def wino(x,w,b,mats,m=4):
    AT,G,BT = mats
    BT32=BT.astype(np.float32); AT32=AT.astype(np.float32)
    t=m+2
    C,H,Wd=x.shape; O=w.shape[0]
    oh,ow=H-2,Wd-2
    nby,nbx=(oh+m-1)//m,(ow+m-1)//m
    xp=np.zeros((C,nby*m+2,nbx*m+2),np.float32); xp[:,:H,:Wd]=x
    idx_y=(np.arange(nby)*m)[:,None]+np.arange(t)[None,:]
    idx_x=(np.arange(nbx)*m)[:,None]+np.arange(t)[None,:]
    d=xp[:,idx_y][:,:,:,idx_x].transpose(0,1,3,2,4)
    V=np.einsum('ij,cyxjk->cyxik',BT32,d).astype(np.float32)
    V=np.einsum('cyxik,lk->cyxil',V,BT32).astype(np.float32)
    U=np.einsum('ij,ocjk,lk->ocil',G,w.astype(np.float64),G).astype(np.float32)
    M=np.einsum('ocil,cyxil->oyxil',U,V).astype(np.float32)
    Y=np.einsum('ij,oyxjk->oyxik',AT32,M).astype(np.float32)
    Y=np.einsum('oyxik,lk->oyxil',Y,AT32).astype(np.float32)
    out=Y.transpose(0,1,3,2,4).reshape(O,nby*m,nbx*m)[:,:oh,:ow]
    return leaky(out+b.astype(np.float32)[:,None,None])
def gate(a,ref): return float((np.abs(a-ref)/(1e-4*np.abs(ref)+1e-5)).max())
cands = {
 'std 0,1,-1,2,-2': [0,1,-1,2,-2],
 '0,1,-1,1/2,-1/2': [0,1,-1,.5,-.5],
 '0,1,-1,1/2,-2': [0,1,-1,.5,-2],
 '0,1,-1,2,-1/2': [0,1,-1,2,-.5],
 '0,1/2,-1/2,3/2,-3/2': [0,.5,-.5,1.5,-1.5],
 '0,1,-1,1/2,-3': [0,1,-1,.5,-3],
 '0,1,-1,3/4,-4/3': [0,1,-1,.75,-4/3],
 '0,1,-1,1/4,-4': [0,1,-1,.25,-4],
 '0,1/2,-1/2,2,-2': [0,.5,-.5,2,-2],
 '0,1,-1,1/2,-1/2 (dup check)': [0,1,-1,.5,-.5],
}
layers=gen_model.synth_layers(seed=102)
rng=np.random.default_rng(7)
x0=rng.random((1,78,78)).astype(np.float32)
# single layer normal test: 32->64 on 21x37 normal (the failing parity test)

sl=gen_model.synth_layers([32,64], 200+32*7+64)
xs=np.random.default_rng(32*3+64+21).standard_normal((32,23,39)).astype(np.float32)
ref_s64=direct(xs.astype(np.float64),sl[0][2],sl[0][3],np.float64); ref_s32=direct(xs,sl[0][2],sl[0][3],np.float32)
x64=x0.astype(np.float64); x32=x0.copy()
for (ni,no,w,b) in layers:
    x64=direct(x64,w,b,np.float64); x32=direct(x32,w,b,np.float32)
rngv=np.abs(x64).max()
print('direct32: err/range %.2e ; single-layer normal: gate use %.3f'%(np.abs(x32-x64).max()/rngv, gate(ref_s32, ref_s64.astype(np.float32))))
for name,pts in cands.items():
    mats=cook_toom(pts)
    xw=x0.copy()
    for (ni,no,w,b) in layers: xw=wino(xw,w,b,mats)
    ys=wino(xs,sl[0][2],sl[0][3],mats)
    print('%-28s 7-layer err/range %.2e gate %.3f | single-layer normal: max abs err %.2e gate use %.3f | max|BT| %.1f max|AT| %.1f'%(name, np.abs(xw-x64).max()/rngv, gate(xw,x32), np.abs(ys-ref_s64).max(), gate(ys,ref_s32), np.abs(mats[2]).max(), np.abs(mats[0]).max()))
