"""Numerical check (fp64, CPU) of the parity-plane decomposition of the KFAC A factor of a 3x3 / stride 2 / padding 1
convolution: 34 plane-against-plane correlations + 5 + 5 strip correlations + one corner product reproduce F.unfold.
The HIP implementation built on it (round 4, syrk_corr2.hip) was measured and removed: LAB_NOTEBOOK.md, K1."""
import numpy as np, itertools, torch
torch.manual_seed(0)
N,C,H,W=3,4,8,6
X=torch.randn(N,C,H,W,dtype=torch.float64)
Ho,Wo=H//2,W//2
# reference factor: unfold 3x3 s2 p1
U=torch.nn.functional.unfold(X,3,padding=1,stride=2)      # (N, C*9, Ho*Wo)
A=sum(U[n]@U[n].T for n in range(N))                      # (9C,9C), row index (c, a, b)
# parity planes: row parity e: 0 = even rows (r=2u), 1 = odd rows (r=2u+1)
P={ (pe,qe): X[:,:,pe::2,qe::2].numpy() for pe in (0,1) for qe in (0,1)}   # (N,C,Ho,Wo)
# kernel offset a -> (row parity, shift s): a=0 -> (odd, -1), a=1 -> (even, 0), a=2 -> (odd, 0)
par={0:(1,-1),1:(0,0),2:(1,0)}
def shifted(Pl,s,t):
    out=np.zeros_like(Pl)
    # out[u,v] = Pl[u+s, v+t] (zero outside)
    Hh,Ww=Pl.shape[-2:]
    us=slice(max(0,-s),min(Hh,Hh-s)); vs=slice(max(0,-t),min(Ww,Ww-t))
    out[...,us,vs]=Pl[...,us.start+s:us.stop+s, vs.start+t:vs.stop+t]
    return out
def F(pl1,pl2,ds,dt):
    # sum_{n,u,v} pl1[c,u,v] * pl2[c',u+ds,v+dt]
    return np.einsum('ncuv,nduv->cd', pl1, shifted(pl2,ds,dt))
Arec=np.zeros((9*C,9*C))
for a,b,a2,b2 in itertools.product(range(3),repeat=4):
    (pe,s),(qe,t)=par[a],par[b]
    (pe2,s2),(qe2,t2)=par[a2],par[b2]
    blk=F(P[(pe,qe)],P[(pe2,qe2)],s2-s,t2-t)
    # corrections: a=a2=0 -> window rows exclude last odd row u=Ho-1; b=b2=0 -> exclude last col
    if a==0 and a2==0:
        r1=P[(1,qe)][:,:,Ho-1:Ho,:]; r2=P[(1,qe2)][:,:,Ho-1:Ho,:]
        blk-=np.einsum('ncuv,nduv->cd', r1, shifted(r2,0,t2-t))
    if b==0 and b2==0:
        c1=P[(pe,1)][:,:,:,Wo-1:Wo]; c2=P[(pe2,1)][:,:,:,Wo-1:Wo]
        blk-=np.einsum('ncuv,nduv->cd', c1, shifted(c2,s2-s,0))
    if a==0 and a2==0 and b==0 and b2==0:
        blk+=np.einsum('nc,nd->cd', P[(1,1)][:,:,Ho-1,Wo-1], P[(1,1)][:,:,Ho-1,Wo-1])
    for c in range(C):
        for d in range(C):
            Arec[c*9+a*3+b, d*9+a2*3+b2]=blk[c,d]
print("max err", np.abs(Arec-A.numpy()).max(), np.abs(A.numpy()).max())
# count distinct components up to transposition
keys=set()
for a,b,a2,b2 in itertools.product(range(3),repeat=4):
    if (a*3+b) < (a2*3+b2): continue
    (pe,s),(qe,t)=par[a],par[b]; (pe2,s2),(qe2,t2)=par[a2],par[b2]
    k=((pe,qe),(pe2,qe2),s2-s,t2-t); kt=((pe2,qe2),(pe,qe),s-s2,t-t2)
    keys.add(min(k,kt))
sym=[k for k in keys if k[0]==k[1] and k[2]==0 and k[3]==0]
print(len(keys),"distinct F components,",len(sym),"symmetric")
