import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import numpy as np, hevcdl_amd, ref_tools
yuv=ref_tools.synth_yuv(64,64,1,seed=1); lab=ref_tools.make_labels(64,64,1,3,seed=2)
enc=hevcdl_amd.Encoder(64,64,32,max_frames=1); enc.compress_frames(yuv,lab); enc.close()
