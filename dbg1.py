import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import numpy as np, hevcdl_amd, ref_tools
import __graft_entry__ as g; g.build_oracle()
W,H,qp=64,64,32
kind = int(sys.argv[1]) if len(sys.argv)>1 else 2
yuv=ref_tools.synth_yuv(W,H,1,seed=1); lab=ref_tools.make_labels(W,H,1,kind,seed=2)
print("smem", hevcdl_amd.load_library().hevcdl_ctus_per_frame(W,H), flush=True)
enc=hevcdl_amd.Encoder(W,H,qp,max_frames=1)
print("created", flush=True)
recs,recon,stats=enc.compress_frames(yuv,lab)
print("done", recs['bits'], recs['dist'], recs['cost'], flush=True)
o,orecon,ostats=ref_tools.run_oracle(yuv,W,H,qp,lab)
print("oracle", o['bits'], o['dist'], o['cost'])
for k in ref_tools.FIELDS:
    if not np.array_equal(recs[k],o[k]):
        d=np.argwhere(np.asarray(recs[k]!=o[k]).reshape(-1)); print("DIFF",k,len(d), d[:8].ravel().tolist(), np.asarray(recs[k]).reshape(-1)[d[:6].ravel()], np.asarray(o[k]).reshape(-1)[d[:6].ravel()])
print("recon equal", np.array_equal(recon,orecon), "stats", stats, ostats)
