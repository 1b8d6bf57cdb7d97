import sys, numpy as np
sys.path.insert(0,'/root/repo/oracle'); sys.path.insert(0,'/root/repo/tests')
import oracle_py as O
from conftest import load_pkg
pkg=load_pkg()
def run(h, n, seed, excite=1.0):
    b=pkg.make_batch(n,h,'a1',seed=seed,excite=excite); cfg=pkg.mpc_cfg('a1'); A=O.mpc_constraint_matrix(h)
    for i in range(n):
        H,g,ub=O.mpc_assemble(cfg,h,b['mpc_state'][i],b['traj'][i],b['gait'][i])
        Hd=H.astype(np.float64); gd=g.astype(np.float64); ubd=ub.astype(np.float64)
        x_ref,info=O.ref_qpoases_mpc(Hd,gd,A,np.zeros(20*h),ubd,100)
        if info['init_rc']!=0 or info['nWSR']>=100: print(i,'cap'); continue
        u,st,rc=O.mpc_solve(cfg,h,b['mpc_state'][i],b['traj'][i],b['gait'][i])
        # active set from the symmetric optimum
        Ax=A@u
        lo=np.abs(Ax)<1e-7; up=np.abs(Ax-ubd)<1e-7
        act=np.where(lo|up)[0]
        # remove dependent rows greedily
        rows=[];
        for r in act:
            test=A[rows+[r]]
            if np.linalg.matrix_rank(test,tol=1e-9)==len(rows)+1: rows.append(r)
        Aa=A[rows]; ba=np.where(up[rows],ubd[rows],0.0)
        na=len(rows); nv=12*h
        def kkt(Hm):
            K=np.block([[Hm,-Aa.T],[Aa,np.zeros((na,na))]])
            sol=np.linalg.solve(K,np.r_[-gd,ba]); return sol[:nv]
        x_asym=kkt(Hd); x_sym=kkt(0.5*(Hd+Hd.T)); x_T=kkt(Hd.T)
        s=max(1,np.abs(x_ref[:12]).max())
        print(i,'nWSR',info['nWSR'],'|sym-ref| %.2e'%(np.abs(x_sym[:12]-x_ref[:12]).max()/s),'|asym-ref| %.2e'%(np.abs(x_asym[:12]-x_ref[:12]).max()/s),'|T-ref| %.2e'%(np.abs(x_T[:12]-x_ref[:12]).max()/s), '|sym-oracle| %.1e'%(np.abs(x_sym-u).max()))
run(10,8,1001)
run(16,6,1006)
