import os, sys, subprocess, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/oracle')
import test_rd_gpu as T, hevcdl_amd
name = sys.argv[1]
lib = os.path.join(hevcdl_amd.PKG_DIR, "lib", "libhevcdl_hip_trace.so")
out = "/tmp/log.npz"
r = subprocess.run([sys.executable, "-c", T._STAGE_CHILD % ('/root/repo', '/root/repo/oracle'), os.path.join(T.GOLD, name + ".npz"), out], env=dict(os.environ, HEVCDL_LIB=lib), capture_output=True, text=True)
print(r.returncode, r.stderr[-500:])
f, g = np.load(os.path.join(T.GOLD, name + ".npz")), np.load(out)
rl, rt = T._stage_sets_of_fixture(f); dl, dt = T._stage_sets_of_log(g["words"])
print("ref lines", len(rl), "dev lines", len(dl), "ref tus", len(rt), "dev tus", len(dt), "missing", len(rt - dt))
import collections
print("missing by (kind,n,comp)", collections.Counter(t[:3] for t in rt - dt).most_common())
print("present by (kind,n,comp)", collections.Counter(t[:3] for t in rt & dt).most_common())
shown = 0
for t in sorted(rt - dt):
    k, n, c, b = t; rb = np.frombuffer(b, np.int32).reshape(3, n, n)
    cands = [np.frombuffer(d[3], np.int32).reshape(3, n, n) for d in dt if d[:3] == t[:3]]
    same0 = [x for x in cands if np.array_equal(x[0], rb[0])]
    print("missing", t[:3], "candidates with the same first block:", len(same0))
    if same0:
        x = same0[0]
        for j in range(3):
            if not np.array_equal(x[j], rb[j]): print(" block", j, "ref\n", rb[j], "\n dev\n", x[j])
    else:
        print(" ref first block\n", rb[0])
    shown += 1
    if shown >= 3: break
