cd $GRAFT_REPO_ROOT
python tools/time_cnn.py 128 3 /tmp/la.npy
HEVCDL_LIB=/root/repo/hevc-deep-learning-pipeline_amd/lib/libhevcdl_hip_b.so python tools/time_cnn.py 128 3 /tmp/lb.npy
python - <<'PY'
import numpy as np
a=np.load('/tmp/la.npy'); b=np.load('/tmp/lb.npy'); la=np.load('/tmp/la.npy.labels.npy'); lb=np.load('/tmp/lb.npy.labels.npy')
print("max |logit diff| %.3g   labels differing %d of %d" % (np.abs(a-b).max(), int((la!=lb).sum()), la.size))
PY
