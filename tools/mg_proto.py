"""numpy prototype of the multigrid V-cycle (float32), used to choose the algorithm before writing HIP."""
import numpy as np, sys, time
F32=np.float32
def rb_sweep(U,F,n=1,omega=1.0):
    H,W=U.shape
    yy,xx=np.mgrid[0:H,0:W]
    inter=np.zeros((H,W),bool); inter[1:-1,1:-1]=True
    ms=[inter&(((xx+yy)&1)==c) for c in (0,1)]
    q=F32(0.25); om=F32(omega)
    for _ in range(n):
        for m in ms:
            s=np.zeros_like(U); s[1:-1,1:-1]=(U[1:-1,:-2]+U[1:-1,2:])+(U[:-2,1:-1]+U[2:,1:-1])
            gs=q*(s-F)
            new = gs if omega==1.0 else U+om*(gs-U)
            U[m]=new[m]
    return U
def resid(U,F):
    r=np.zeros_like(U)
    r[1:-1,1:-1]=F[1:-1,1:-1]-((U[1:-1,:-2]+U[1:-1,2:])+(U[:-2,1:-1]+U[2:,1:-1])-F32(4)*U[1:-1,1:-1])
    return r
def restrict(r):
    H,W=r.shape; Hc,Wc=H//2+1,W//2+1
    # pad r so index 2J+1 is valid up to 2(Hc-1)+1
    rp=np.zeros((2*Hc+1,2*Wc+1),F32); rp[:H,:W]=r
    Fc=np.zeros((Hc,Wc),F32)
    J=np.arange(1,Hc-1)[:,None]*2; I=np.arange(1,Wc-1)[None,:]*2
    c=rp[J,I]; e=(rp[J,I-1]+rp[J,I+1])+(rp[J-1,I]+rp[J+1,I]); k=(rp[J-1,I-1]+rp[J-1,I+1])+(rp[J+1,I-1]+rp[J+1,I+1])
    Fc[1:-1,1:-1]=F32(0.25)*((F32(4)*c+F32(2)*e)+k)   # = 4 * full weighting
    return Fc
def prolong_add(U,E):
    H,W=U.shape; Hc,Wc=E.shape
    Ep=np.zeros((Hc+1,Wc+1),F32); Ep[:Hc,:Wc]=E
    y=np.arange(1,H-1)[:,None]; x=np.arange(1,W-1)[None,:]
    J=y//2; I=x//2; oy=(y&1); ox=(x&1)
    a=Ep[J,I]; b=Ep[J,I+1]; c=Ep[J+1,I]; d=Ep[J+1,I+1]
    v=np.where((oy==0)&(ox==0),a, np.where((oy==0),F32(0.5)*(a+b), np.where(ox==0,F32(0.5)*(a+c),F32(0.25)*((a+b)+(c+d)))))
    U[1:-1,1:-1]+=v.astype(F32)
    return U
def vcycle(U,F,pre=2,post=2,lvl=0,coarse_min=4):
    H,W=U.shape
    if min(H,W)-2<=coarse_min or lvl>20:
        w,h=W-2,H-2
        rho=0.5*(np.cos(np.pi/(w+1))+np.cos(np.pi/(h+1))); om=2/(1+np.sqrt(max(0,1-rho*rho)))
        n=int(max(8, 4*max(w,h))) if min(w,h)>1 else max(w,h)*2+8
        return rb_sweep(U,F,min(n,64),om)
    U=rb_sweep(U,F,pre)
    Fc=restrict(resid(U,F))
    E=np.zeros_like(Fc)
    E=vcycle(E,Fc,pre,post,lvl+1,coarse_min)
    U=prolong_add(U,E)
    U=rb_sweep(U,F,post)
    return U
if __name__=='__main__':
    from scipy import fft as sfft
    for (W,H) in [(298,192),(511,511),(512,512),(1026,770),(1000,39),(130,2048)]:
        rng=np.random.default_rng(5)
        B=rng.uniform(0,255,(H,W)).astype(F32); yy,xx=np.mgrid[0:H,0:W]
        B=(128+60*np.sin(xx/37.0)*np.cos(yy/23.)+rng.normal(0,12,(H,W))).astype(F32)
        F=np.zeros((H,W),F32); F[1:-1,1:-1]=rng.normal(0,30,(H-2,W-2)).astype(F32)
        # exact
        g=F[1:-1,1:-1].astype(np.float64).copy(); g[:,0]-=B[1:-1,0]; g[0,:]-=B[0,1:-1]; g[:,-1]-=B[1:-1,-1]; g[-1,:]-=B[-1,1:-1]
        h,w=g.shape; den=(2*np.cos(np.pi*(np.arange(w)+1)/(w+1)))[None,:]+(2*np.cos(np.pi*(np.arange(h)+1)/(h+1)))[:,None]-4
        uex=sfft.idstn(sfft.dstn(g,type=1)/den,type=1)
        U=B.copy(); f2=np.sqrt((F.astype(np.float64)**2).sum())
        t=time.time(); prev=None; line=[]
        for cyc in range(10):
            U=vcycle(U,F)
            r=np.sqrt((resid(U,F).astype(np.float64)**2).sum())/f2; err=np.abs(U[1:-1,1:-1]-uex).max()
            line.append('%.1e/%.3f'%(r,err))
        print(W,H,' '.join(line),'%.1fs'%(time.time()-t),flush=True)
