import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import oracle_lib, test_gpu_random as T
from helpers import graph_of, rel_err
from visfs_amd import abi
olib = oracle_lib.load()

def cramer3(h):
    a,b,c,d,e,f = h
    c00=d*f-e*e; c01=c*e-b*f; c02=b*e-c*d
    det=a*c00+b*c01+c*c02
    i=1.0/det
    return np.array([[c00*i,c01*i,c02*i],[c01*i,(a*f-c*c)*i,(b*c-a*e)*i],[c02*i,(b*c-a*e)*i,(a*d-b*b)*i]])
def sym6(M): return [M[0,0],M[0,1],M[0,2],M[1,1],M[1,2],M[2,2]]
def inv6_cramer(D):
    P=D[:3,:3]; Q=D[:3,3:]; R=D[3:,3:]
    Pi=cramer3(sym6(P)); Tm=Pi@Q; Sc=R-Q.T@Tm; Si=cramer3(sym6(Sc)); U=-Tm@Si; V=Pi-U@Tm.T
    return np.block([[V,U],[U.T,Si]])
def inv6_ns(D):
    X=inv6_cramer(D); return X@(2*np.eye(6)-D@X)
def inv6_chol(D):
    C=np.linalg.cholesky(D); Ci=np.linalg.solve(C,np.eye(6)); return Ci.T@Ci

def chol6(D):
    C=np.zeros((6,6)); A=D.copy()
    for j in range(6):
        if not A[j,j]>0: raise np.linalg.LinAlgError("pivot")
        r=1.0/np.sqrt(A[j,j]); C[j:,j]=A[j:,j]*r
        for i in range(j+1,6):
            A[i:,i]-=C[i:,j]*C[i,j]
    return C
def triinv(C):
    X=np.zeros((6,6))
    for j in range(6):
        X[j,j]=1.0/C[j,j]
        for i in range(j+1,6):
            X[i,j]=-(C[i,j:i]@X[j:i,j])/C[i,i]
    return X
def block_ldl_solve(S,b,inv6,refine=0, tri=False, triinv_=False):
    n=S.shape[0]//6
    A=S.copy(); L=np.zeros_like(S); Dinv=[]
    Cs=[]
    for k in range(n):
        Dk=A[6*k:6*k+6,6*k:6*k+6]
        if triinv_:
            C=chol6(Dk); Ci=triinv(C); Cs.append(Ci)
        elif tri:
            C=np.linalg.cholesky(Dk); Cs.append(C)
        else:
            Di=inv6(Dk); Dinv.append(Di)
        for i in range(k+1,n):
            G=A[6*i:6*i+6,6*k:6*k+6].copy()
            if triinv_:
                Lik=(Ci.T@(Ci@G.T)).T
            elif tri:
                import scipy.linalg as sl
                Lik=sl.solve_triangular(C, sl.solve_triangular(C, G.T, lower=True), lower=True, trans='T').T
            else:
                Lik=G@Di
            L[6*i:6*i+6,6*k:6*k+6]=Lik
            for j in range(k+1,i+1):
                Gj=A[6*j:6*j+6,6*k:6*k+6]
                A[6*i:6*i+6,6*j:6*j+6]-=Lik@Gj.T
    def solve(rhs):
        c=rhs.copy()
        for k in range(n):
            for i in range(k+1,n): c[6*i:6*i+6]-=L[6*i:6*i+6,6*k:6*k+6]@c[6*k:6*k+6]
        for k in range(n):
            if triinv_:
                c[6*k:6*k+6]=Cs[k].T@(Cs[k]@c[6*k:6*k+6])
            elif tri:
                import scipy.linalg as sl
                c[6*k:6*k+6]=sl.cho_solve((Cs[k],True),c[6*k:6*k+6])
            else: c[6*k:6*k+6]=Dinv[k]@c[6*k:6*k+6]
        for k in range(n-1,-1,-1):
            for j in range(k): c[6*j:6*j+6]-=L[6*k:6*k+6,6*j:6*j+6].T@c[6*k:6*k+6]
        return c
    x=solve(b)
    for _ in range(refine):
        r=b-S@x
        x=x+solve(r)
    return x

for seed in [int(a) for a in sys.argv[1:]] or [559]:
    w,kw=T.random_case(seed)
    prm=abi.default_params(iterations=10, solver=0, robust_kernel_delta=kw["robust_kernel_delta"])
    wb,gb,*_=graph_of(olib.oracle_pack_window, prm, w)
    o=oracle_lib.OracleSystem(olib, prm, gb)
    o.linearize(); n6=6*o.npf
    for lam in (1e-2,1e-5,1e-8):
        o.trial(lam)
        S=o.fetch(abi.BUF_S).reshape(n6,n6).copy(); b=o.fetch(abi.BUF_BS).copy(); xo=o.fetch(abi.BUF_DX_POSE)
        xnp=np.linalg.solve(S,b)
        ev=np.linalg.eigvalsh(S)
        print(f"seed {seed} lambda {lam:.0e} cond {ev[-1]/ev[0]:.1e}: oracle {rel_err(xo,xnp):.1e}", end='')
        for name,fn,kwargs in (("cramer",inv6_cramer,{}),("cramer+ref1",inv6_cramer,{'refine':1}),("cramer+ref2",inv6_cramer,{'refine':2}),("ns",inv6_ns,{}),("ns+ref1",inv6_ns,{'refine':1}),("cholinv",inv6_chol,{}),("cholinv+ref1",inv6_chol,{'refine':1}),("tri",None,{'tri':True}),("tri+ref1",None,{'tri':True,'refine':1}),("triinv",None,{'triinv_':True}),("triinv+ref1",None,{'triinv_':True,'refine':1})):
            try: x=block_ldl_solve(S,b,fn,**kwargs)
            except np.linalg.LinAlgError: print(f" | {name} FAIL", end=''); continue
            print(f" | {name} {rel_err(x,xnp):.1e}", end='')
        print()
