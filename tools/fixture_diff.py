import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import numpy as np, hevcdl_amd, ref_tools
f=np.load('tests/golden/rd_%s.npz'%sys.argv[1])
w,h,qp=int(f['width']),int(f['height']),int(f['qp'])
enc=hevcdl_amd.Encoder(w,h,qp,max_frames=f['yuv'].shape[0])
recs,recon,stats=enc.compress_frames(f['yuv'],f['labels'])
ref=f['records']
for k in ref_tools.FIELDS:
    if not np.array_equal(recs[k],ref[k]):
        d=np.argwhere(np.asarray(recs[k]!=ref[k]).reshape(recs.shape[0],recs.shape[1],-1))
        print("DIFF",k,len(d),d[:5].tolist(), np.asarray(recs[k]).reshape(recs.shape[0],recs.shape[1],-1)[tuple(d[0])], np.asarray(ref[k]).reshape(recs.shape[0],recs.shape[1],-1)[tuple(d[0])])
