import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import numpy as np, hevcdl_amd, ref_tools
W,H,nf=int(sys.argv[1]),int(sys.argv[2]),int(sys.argv[3])
yuv=np.repeat(ref_tools.synth_yuv(W,H,1,seed=1),nf,axis=0)
enc=hevcdl_amd.Encoder(W,H,32,max_frames=nf)
enc.profile_enable(True)
t=time.time(); lab=enc.predict_depth(yuv); t1=time.time()-t
print("labels hist", np.bincount(lab.ravel(),minlength=4))
t=time.time(); recs,recon,stats=enc.compress_frames(yuv,lab); t2=time.time()-t
pr=enc.profile_get()
ct=lab.shape[0]*lab.shape[1]
print("frames",nf,"ctus",ct,"cnn wall %.3f rd wall %.3f"%(t1,t2), pr, "RD CTU/s %.1f CNN CTU/s %.1f"%(ct/(pr['rd_ms']/1e3), 2*ct/(pr['cnn_ms']/1e3)))
