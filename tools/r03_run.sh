cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_cnn_gpu.py -q -x -m gpu 2>&1 | tail -3
python - <<'PY'
import sys, time
sys.path.insert(0,'.'); sys.path.insert(0,'oracle')
import numpy as np, torch, hevcdl_amd, ref_tools, cnn_oracle
W,H,F=3840,2160,600
base=ref_tools.synth_yuv(W,H,4,seed=4000)
enc=hevcdl_amd.Encoder(W,H,32,max_frames=F)
yuv=torch.from_numpy(np.ascontiguousarray(base.reshape(4,-1))).cuda().repeat(F//4,1).contiguous()
lab=torch.empty((F,enc.ctus,16),dtype=torch.uint8,device='cuda')
enc.profile_enable(True)
for i in range(3):
    enc.predict_depth_dev(yuv.data_ptr(),F,lab.data_ptr()); torch.cuda.synchronize()
pr=enc.profile_get(); print("CNN ms per 600 frames:", pr['cnn_ms']/pr['cnn_launches'])
w,h=1920,1080
y1=ref_tools.synth_yuv(w,h,1,seed=77)
e2=hevcdl_amd.Encoder(w,h,32,max_frames=1)
labels,logits=e2.predict_depth(y1,want_logits=True)
wts=cnn_oracle.load_weights(hevcdl_amd.WEIGHTS_PATH)
ol,ologits=cnn_oracle.predict_labels(wts,y1,w,h)
print("max |logit diff|", float(np.abs(logits-ologits).max()), "labels equal", float((labels==ol).mean()))
PY
