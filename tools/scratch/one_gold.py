import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/oracle')
import numpy as np, hevcdl_amd
from conftest import fixture_tiles
path=sys.argv[1]
f=np.load(path)
w,h,qp=int(f["width"]),int(f["height"]),int(f["qp"])
yuv,labels,ref=f["yuv"],f["labels"],f["records"]
tiles=fixture_tiles(f); bd=int(f["bit_depth"]) if "bit_depth" in f.files else 8
enc=hevcdl_amd.Encoder(w,h,qp,max_frames=yuv.shape[0],tiles=tiles,bit_depth=bd)
recs,recon,stats=enc.compress_frames(yuv,labels); enc.close()
bad=[k for k in ["depth","part_size","luma_dir","chroma_dir","tr_idx","cbf","tskip","bits","dist","cost","coeff_y","coeff_cb","coeff_cr"] if not np.array_equal(recs[k],ref[k])]
print("RESULT", os.path.basename(path) if False else path.split("/")[-1], w,h,qp,"labels",np.bincount(labels.ravel(),minlength=4).tolist(),"BAD",bad)
