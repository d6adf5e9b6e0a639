#!/usr/bin/env python3
"""Sizing of a depth-tile shortcut for k_integrate, on the CPU: which share of the headline's wave-rows
(256 x-consecutive voxels of one row of one plane, 2048^3, Scene A, 640x480) could a per-tile min / max of the depth frame
decide without projecting a voxel?
  "behind": max depth over the row's pixel footprint - least g.z < -max_dist_neg  => every voxel rejected (hpp:193-196)
  "free"  : min depth over the footprint - greatest g.z > max_dist_pos, no NaN     => every voxel observed with d_new = p
The footprint is the pixel bounding box of the row's two end voxels (+- 1 px; a projective map takes the segment to a
segment), queried in T x T tiles.  `missed` = rows that ARE uniformly behind / free but that the tiles cannot prove.
usage: tile_shortcut_sizing.py [T ...]   (samples 5 poses x 24 planes x 24 rows x 8 chunks)"""
import sys

import numpy as np

sys.path.insert(0,'/root/repo')
from cpu_tsdf_amd import synth
# 2048^3 headline geometry, sampled: every 8th plane/row, all x; wave-rows = 256 voxels
res=2048; vs=2.0**-8; S=res*vs
sc=synth.scene_a(res)
pos=neg=0.03
tot=beh=free=mixed=0; obs=0; vox=0
for T in [int(t) for t in sys.argv[1:]] or [16, 8, 4]:
  tot=beh=free=mixed=0; obs=0; vox=0
  for fi in (0,3,7,12,17):
      tr=synth.turntable_pose(fi,23,S)
      dep=sc.depth(tr).astype(np.float64)
      Tm=synth.eigen_affine_inverse(tr)
      H,W=dep.shape
      # tile min/max
      th,tw=(H+T-1)//T,(W+T-1)//T
      dpad=np.full((th*T,tw*T),np.nan); dpad[:H,:W]=dep
      tiles=dpad.reshape(th,T,tw,T)
      tmax=np.nanmax(np.where(np.isnan(tiles),-np.inf,tiles),axis=(1,3))
      tmin=np.min(np.where(np.isnan(tiles),-np.inf,tiles),axis=(1,3))
      c=(np.arange(res)+0.5)*vs-S/2
      x=c
      rng=np.random.RandomState(fi)
      for z in rng.choice(res,24,replace=False):
        for y in rng.choice(res,24,replace=False):
          p=np.stack([x,np.full(res,c[y]),np.full(res,c[z]),np.ones(res)],0)
          g=Tm[:3]@p
          u=(g[0]*sc.fx/g[2]+sc.cx).astype(int); v=(g[1]*sc.fy/g[2]+sc.cy).astype(int)
          zz=dep[v,u]; raw=zz-g[2]
          act=raw>=-neg
          obs+=act.sum(); vox+=res
          for ch in range(res//256):
              sl=slice(ch*256,ch*256+256)
              a=act[sl]; r=raw[sl]
              tot+=1
              # exact classes
              ex_beh = not a.any(); ex_free = bool((r>pos).all())
              # tile-based decision
              u0,u1=min(u[sl][0],u[sl][-1])-1,max(u[sl][0],u[sl][-1])+1
              v0,v1=min(v[sl][0],v[sl][-1])-1,max(v[sl][0],v[sl][-1])+1
              tx0,tx1=max(u0,0)//T,min(u1,W-1)//T; ty0,ty1=max(v0,0)//T,min(v1,H-1)//T
              mx=tmax[ty0:ty1+1,tx0:tx1+1].max(); mn=tmin[ty0:ty1+1,tx0:tx1+1].min()
              gzmin,gzmax=g[2][sl].min(),g[2][sl].max()
              t_beh = mx-gzmin < -neg-1e-3
              t_free = mn-gzmax > pos+1e-3
              beh+=t_beh; free+=t_free
              if t_beh: assert ex_beh
              if t_free: assert ex_free
              mixed += (ex_beh and not t_beh) + (ex_free and not t_free)
  print(f"T = {T:2d} px tiles:", "wave-rows",tot,"tile-proved behind",beh/tot,"tile-proved free",free/tot,"exactly-decidable but missed",mixed/tot,"observed frac",obs/vox)
