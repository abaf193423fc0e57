"""numpy prototype #2: multigrid with a non-uniform LAST interval per level (Shortley-Weller-like
coarse grids) so arbitrary ROI sizes coarsen without moving the Dirichlet boundary."""
import numpy as np, sys, time
F32=np.float32
R64=True
class Lvl:
    def __init__(s,nx,ax,ny,ay): s.nx,s.ax,s.ny,s.ay=nx,F32(ax),ny,F32(ay)
    def coef(s):
        H,W=s.ny+2,s.nx+2
        cW=np.ones(W,F32); cE=np.ones(W,F32); dX=np.full(W,2,F32)
        cW[s.nx]=F32(2)/(F32(1)+s.ax); cE[s.nx]=F32(2)/(s.ax*(F32(1)+s.ax)); dX[s.nx]=F32(2)/s.ax
        cN=np.ones(H,F32); cS=np.ones(H,F32); dY=np.full(H,2,F32)
        cN[s.ny]=F32(2)/(F32(1)+s.ay); cS[s.ny]=F32(2)/(s.ay*(F32(1)+s.ay)); dY[s.ny]=F32(2)/s.ay
        return cW,cE,dX,cN,cS,dY
def coarsen1(n,a):
    if n%2==1: return (n-1)//2,(1+a)/2
    if a>=1: return n//2,a/2
    return n//2-1,1+a/2
def apply_sum(L,U):
    cW,cE,dX,cN,cS,dY=L.coef()
    s=np.zeros_like(U)
    s[1:-1,1:-1]=(cW[None,1:-1]*U[1:-1,:-2]+cE[None,1:-1]*U[1:-1,2:])+(cN[1:-1,None]*U[:-2,1:-1]+cS[1:-1,None]*U[2:,1:-1])
    D=np.zeros_like(U); D[1:-1,1:-1]=dX[None,1:-1]+dY[1:-1,None]
    return s,D
def rb(L,U,F,n=1,omega=1.0):
    H,W=U.shape; yy,xx=np.mgrid[0:H,0:W]; inter=np.zeros((H,W),bool); inter[1:-1,1:-1]=True
    for _ in range(n):
        for c in (0,1):
            m=inter&(((xx+yy)&1)==c)
            s,D=apply_sum(L,U)
            gs=np.zeros_like(U); gs[1:-1,1:-1]=(s[1:-1,1:-1]-F[1:-1,1:-1])/D[1:-1,1:-1]
            new=gs if omega==1.0 else U+F32(omega)*(gs-U)
            U[m]=new[m]
    return U
def resid(L,U,F):
    s,D=apply_sum(L,U); r=np.zeros_like(U); r[1:-1,1:-1]=F[1:-1,1:-1]-(s[1:-1,1:-1]-D[1:-1,1:-1]*U[1:-1,1:-1]); return r
def P1d(n,a,nc):
    """(n+2) x (nc+2) 1-D prolongation matrix incl. ring rows/cols (ring = 0)."""
    P=np.zeros((n+2,nc+2),np.float64)
    for i in range(1,n+1):
        if i<=2*nc:
            if i%2==0: P[i,i//2]=1
            else:
                P[i,(i-1)//2]+=0.5; P[i,(i+1)//2]+=0.5
        else:
            D=n+a-2*nc; d=i-2*nc; P[i,nc]=1-d/D
    P[:,0]=0; P[:,nc+1]=0
    return P
def transfer(Lf,Lc):
    Px=P1d(Lf.nx,float(Lf.ax),Lc.nx); Py=P1d(Lf.ny,float(Lf.ay),Lc.ny)
    Rx=Px.T.copy(); sx=Rx.sum(1); sx[sx==0]=1; Rx/=sx[:,None]
    Ry=Py.T.copy(); sy=Ry.sum(1); sy[sy==0]=1; Ry/=sy[:,None]
    return Px.astype(F32),Py.astype(F32),Rx.astype(F32),Ry.astype(F32)
def vcycle(levels,l,U,F,pre=2,post=2):
    L=levels[l]
    if l==len(levels)-1:
        w,h=L.nx,L.ny
        rho=0.5*(np.cos(np.pi/(w+1))+np.cos(np.pi/(h+1))); om=2/(1+np.sqrt(max(0,1-rho*rho)))
        return rb(L,U,F,32,om)
    U=rb(L,U,F,pre)
    Px,Py,Rx,Ry=transfer(L,levels[l+1])
    if l==0 and R64:
        U64=U.astype(np.float64); r=np.zeros_like(U64)
        r[1:-1,1:-1]=F[1:-1,1:-1].astype(np.float64)-((U64[1:-1,:-2]+U64[1:-1,2:])+(U64[:-2,1:-1]+U64[2:,1:-1])-4*U64[1:-1,1:-1])
        r=r.astype(F32)
    else:
        r=resid(L,U,F)
    Fc=F32(4)*(Ry@r@Rx.T)
    Fc[0,:]=Fc[-1,:]=0; Fc[:,0]=Fc[:,-1]=0
    E=vcycle(levels,l+1,np.zeros_like(Fc),Fc,pre,post)
    U[1:-1,1:-1]+=(Py@E@Px.T)[1:-1,1:-1]
    return rb(L,U,F,post)
def build_levels(W,H,cmin=3):
    L=[Lvl(W-2,1.0,H-2,1.0)]
    while True:
        a=L[-1]
        if min(a.nx,a.ny)<=cmin: break
        nx,ax=coarsen1(a.nx,float(a.ax)); ny,ay=coarsen1(a.ny,float(a.ay))
        if nx<1 or ny<1: break
        L.append(Lvl(nx,ax,ny,ay))
    return L
if __name__=='__main__':
    from scipy import fft as sfft
    sizes=[(298,192),(511,511),(512,512),(1026,770),(1000,39),(130,2048),(77,53),(2048,2048)] if len(sys.argv)<2 else [tuple(map(int,a.split('x'))) for a in sys.argv[1:]]
    for (W,H) in sizes:
        rng=np.random.default_rng(5); yy,xx=np.mgrid[0:H,0:W]
        B=(128+60*np.sin(xx/37.0)*np.cos(yy/23.)+rng.normal(0,12,(H,W))).astype(F32)
        F=np.zeros((H,W),F32); F[1:-1,1:-1]=rng.normal(0,30,(H-2,W-2)).astype(F32)
        g=F[1:-1,1:-1].astype(np.float64).copy(); g[:,0]-=B[1:-1,0]; g[0,:]-=B[0,1:-1]; g[:,-1]-=B[1:-1,-1]; g[-1,:]-=B[-1,1:-1]
        h,w=g.shape; den=(2*np.cos(np.pi*(np.arange(w)+1)/(w+1)))[None,:]+(2*np.cos(np.pi*(np.arange(h)+1)/(h+1)))[:,None]-4
        uex=sfft.idstn(sfft.dstn(g,type=1)/den,type=1)
        levels=build_levels(W,H)
        U=B.copy(); f2=np.sqrt((F.astype(np.float64)**2).sum()); line=[]; t=time.time()
        for cyc in range(8):
            U=vcycle(levels,0,U,F)
            U64=U.astype(np.float64); rr=F[1:-1,1:-1].astype(np.float64)-((U64[1:-1,:-2]+U64[1:-1,2:])+(U64[:-2,1:-1]+U64[2:,1:-1])-4*U64[1:-1,1:-1]); r=np.sqrt((rr**2).sum())/f2; err=np.abs(U[1:-1,1:-1]-uex).max()
            line.append('%.1e/%.3f'%(r,err))
        print(W,H,'L=%d'%len(levels),[(l.nx,round(float(l.ax),3),l.ny,round(float(l.ay),3)) for l in levels[-2:]],' '.join(line),'%.1fs'%(time.time()-t),flush=True)
