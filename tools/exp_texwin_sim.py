#!/usr/bin/env python3
"""Development (round 5, docs/NOTES.md 9.3): would a DENSE texel window per mip level hold the taps of a 16x16-pixel block of config 3?
CPU only: uv / uv_da of the benchmark mesh from the oracle, then per block the share of pixels with a tap outside a 34 / 18 / 10
window set anchored at the block's smallest texture coordinate (what k_tex_grad_win did), outside windows derived from the block's
uv bounding box (S1), and outside per-level corners with a full window for EVERY level (S2), plus the number of distinct mip
levels per block.      python tools/exp_texwin_sim.py"""
import os
import numpy as np, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from nvdiffrast_amd.utils import m10k_batch
b=m10k_batch(4)
R=1024
res=[]
for n in range(4):
    ro, rdb = oracle.rasterize(b["pos"][n:n+1], b["tri"], (R,R))
    uv, uvda = oracle.interpolate(b["uv"], ro, b["tri"], rast_db=rdb, diff_attrs="all")
    uv=uv[0]; da=uvda[0]
    cov = ro[0,...,3]>0
    TW=2048
    dsdx=da[...,0]*TW; dsdy=da[...,1]*TW; dtdx=da[...,2]*TW; dtdy=da[...,3]*TW
    A=dsdx**2+dtdx**2; B=dsdy**2+dtdy**2; C=dsdx*dsdy+dtdx*dtdy
    l2b=0.5*(A+B); l2a=np.sqrt(0.25*(A-B)**2+C*C)
    with np.errstate(divide='ignore'):
        fl=0.5*np.log2(l2b+l2a)
    fl=np.clip(np.nan_to_num(fl,neginf=0),0,11)
    l0=np.floor(fl).astype(int)
    foot = cov & ~((da==0).all(-1))
    u=uv[...,0]-np.floor(uv[...,0]); v=uv[...,1]-np.floor(uv[...,1])
    lostpix=0; heavypix=0; nheavy=0; strad=0
    for by in range(0,R,16):
        for bx in range(0,R,16):
            f=foot[by:by+16,bx:bx+16]
            if not f.any(): continue
            nheavy+=1
            uu=u[by:by+16,bx:bx+16][f]; vv=v[by:by+16,bx:bx+16][f]; ll=l0[by:by+16,bx:bx+16][f]
            Lmin=ll.min(); 
            if ll.max()>Lmin: strad+=1
            umin=uu.min(); vmin=vv.min()
            heavypix+=f.sum()
            lost=np.zeros(f.sum(),bool)
            for lev_off in (0,1):
                lev=ll+lev_off
                li=lev-Lmin
                w=np.maximum(TW>>lev,1)
                ax=np.floor(umin*w-0.5); ay=np.floor(vmin*w-0.5)
                ix=np.floor(uu*w-0.5); iy=np.floor(vv*w-0.5)
                W=np.where(li==0,34,np.where(li==1,18,np.where(li==2,10,0)))
                bad=((ix+1-ax)>=W)|((iy+1-ay)>=W)
                lost|=bad
            lostpix+=lost.sum()
    print(n,"heavy blocks",nheavy,"straddle",strad,"foot px",heavypix,"lost px",lostpix, "frac %.3f"%(lostpix/max(heavypix,1)), "level hist", np.bincount(l0[foot])[:8])

print("---- scheme S1: windows from the block's uv bounding box (levels Lb.. covered), lost = taps finer than Lb; S2: per-level anchors, 34-windows for every level")
for WIN in (34, 46):
  for n in range(2):
    ro, rdb = oracle.rasterize(b["pos"][n:n+1], b["tri"], (R,R))
    uv, uvda = oracle.interpolate(b["uv"], ro, b["tri"], rast_db=rdb, diff_attrs="all")
    uv=uv[0]; da=uvda[0]
    TW=2048
    dsdx=da[...,0]*TW; dsdy=da[...,1]*TW; dtdx=da[...,2]*TW; dtdy=da[...,3]*TW
    A=dsdx**2+dtdx**2; B=dsdy**2+dtdy**2; C=dsdx*dsdy+dtdx*dtdy
    l2b=0.5*(A+B); l2a=np.sqrt(0.25*(A-B)**2+C*C)
    with np.errstate(divide='ignore'):
        fl=0.5*np.log2(l2b+l2a)
    fl=np.clip(np.nan_to_num(fl,neginf=0),0,11)
    l0=np.floor(fl).astype(int)
    cov = ro[0,...,3]>0
    foot = cov & ~((da==0).all(-1))
    u=uv[...,0]-np.floor(uv[...,0]); v=uv[...,1]-np.floor(uv[...,1])
    taps=0; lost1=0; lost2=0; nlev=[]
    for by in range(0,R,16):
        for bx in range(0,R,16):
            f=foot[by:by+16,bx:bx+16]
            if not f.any(): continue
            uu=u[by:by+16,bx:bx+16][f]; vv=v[by:by+16,bx:bx+16][f]; ll=l0[by:by+16,bx:bx+16][f]
            ext=max((uu.max()-uu.min())*TW,(vv.max()-vv.min())*TW)+2
            Lb=max(int(np.ceil(np.log2(max(ext,1)/ (WIN-2)))),0)
            taps+=2*f.sum()
            lost1+= (ll<Lb).sum() + ((ll+1)<Lb).sum()
            levs=np.unique(np.concatenate([ll,ll+1])); nlev.append(len(levs))
            for L in levs:
                sel=(ll==L)|(ll+1==L)
                w=max(TW>>L,1)
                ix=np.floor(uu[sel]*w-0.5); iy=np.floor(vv[sel]*w-0.5)
                bad=((ix+1-ix.min())>=WIN)|((iy+1-iy.min())>=WIN)
                lost2+=bad.sum()
    print("WIN",WIN,"img",n,"tap-levels",taps,"S1 lost %.3f"%(lost1/taps),"S2 lost %.4f"%(lost2/taps),"levels per block mean %.2f max %d"%(np.mean(nlev),max(nlev)))
