import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import numpy as np, hevcdl_amd, ref_tools
W,H=1920,1080
yuv=ref_tools.synth_yuv(W,H,1,seed=4000)
enc=hevcdl_amd.Encoder(W,H,32,max_frames=1); lab=enc.predict_depth(yuv); recs,recon,stats=enc.compress_frames(yuv,lab); enc.close()
d=recs['depth'][0]; t=recs['tr_idx'][0]; ps=recs['part_size'][0]
# per CU stats: iterate CTUs, z-order partitions; CU of depth dd covers 256>>(2*dd) partitions
res={}
for c in range(d.shape[0]):
    z=0
    while z<256:
        dd=int(d[c,z]); n=256>>(2*dd)
        if ps[c,z]==8: z+=n; continue   # SIZE_NONE (outside picture)
        tt=t[c,z:z+n]
        key=(dd,int(ps[c,z]))
        r=res.setdefault(key,{'n':0,'split':0,'child_split':0,'children':0})
        r['n']+=1
        if tt.max()>0:
            r['split']+=1
            q=n//4
            for i in range(4):
                r['children']+=1
                if tt[i*q:(i+1)*q].max()>1: r['child_split']+=1
        z+=n
for k in sorted(res): print(k,res[k])
