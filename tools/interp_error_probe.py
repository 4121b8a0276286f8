"""Interpolation error of the per-pixel arrays along image columns (CPU, the oracle on 64-column strips of the two full-size
pointings): cubic through exact nodes in pairs of rows {k s, k s + 1} at spacings s = 4, 5, 8, by elevation band — the
feasibility figures behind DESIGN.md 6a "Trading precision for instructions".  python tools/interp_error_probe.py"""
import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_numpy as O
from auromat_amd.synthetic import frame_header
from auromat_amd.coordinates import transform as T
W,H=4240,2832
def strip(pointing, x0, nx):
    hdr,cam,t=frame_header(W,H,pointing)
    hdr=dict(hdr); hdr['CRPIX1']=hdr['CRPIX1']-x0; hdr['IMAGEW']=nx
    et=T.date2es(t)
    g=O.georef_frame(hdr,110,cam,O.mat_j2000_to_geo(et),None,fast=True)
    return g
def hermite_pair_err(f, s):
    """f: (rows, cols) exact samples along rows. nodes pairs at (k*s, k*s+1). cubic through 4 nodes of two adjacent groups;
    returns abs error array (NaN where not interpolated)"""
    n=f.shape[0]
    err=np.full(f.shape,np.nan)
    k=0
    while k*s+s+1 < n:
        a=k*s; nodes=np.array([a,a+1,a+s,a+s+1])
        for r in range(a+2,a+s):
            w=np.ones(4)
            for i in range(4):
                for j in range(4):
                    if i!=j: w[i]*= (r-nodes[j])/(nodes[i]-nodes[j])
            val=sum(w[i]*f[nodes[i]] for i in range(4))
            err[r]=np.abs(val-f[r])
        k+=1
    return err
for pointing in ('iss030','iss029'):
    for x0 in (0, 2000, 4100):
        g=strip(pointing,x0,64)
        el=g['elev']
        print(pointing,x0,'hit rows', int((~np.isnan(g['lat'][:,0])).sum()), 'elev range', np.nanmin(el), np.nanmax(el))
        for s in (4,5,8):
            for name in ('lat','lon','lat_c','lon_c','elev'):
                f=g[name]
                e=hermite_pair_err(f,s)
                # bucket by elevation of the nearest centre row
                elr=el if f.shape==el.shape else np.vstack([el,el[-1:]])[:, :f.shape[1]] if f.shape[1]==el.shape[1] else None
                if elr is None:
                    elr=np.vstack([el,el[-1:]]); elr=np.hstack([elr,elr[:,-1:]])
                out=[]
                for lo,hi in ((0,5),(5,10),(10,15),(15,20),(20,30),(30,50),(50,91)):
                    m=(elr>=lo)&(elr<hi)&~np.isnan(e)
                    out.append('%d-%d:%.1e'%(lo,hi,e[m].max()) if m.any() else '%d-%d:-'%(lo,hi))
                print('  s=%d %-6s'%(s,name),' '.join(out))
