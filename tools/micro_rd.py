"""Leaf routines of a TU coding timed on their own (csrc/rd_kernel.hip, -DHEVCDL_MICRO): cycles per call of RDOQ / the bit counter / transform + RDOQ + bit
counter + inverse on synthetic residual blocks, with 8 waves per CU (the product's occupancy) and with one wave per CU (latency of a wave alone).
usage: tools/micro_rd.py [build] [--lib PATH] [--amp A] [--reps N]"""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hevcdl_amd
LIB = os.path.join(hevcdl_amd.PKG_DIR, "lib", "libhevcdl_hip_micro.so")


def build(out=LIB, defines=()):
    src = os.path.join(hevcdl_amd.PKG_DIR, "csrc", "rd_kernel.hip")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", "-Wno-unused-value", "-mllvm", "-amdgpu-spill-vgpr-to-agpr=0",
           "-DHEVCDL_MICRO", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(hevcdl_amd.PKG_DIR, "csrc")] + ["-D" + d for d in defines] + [src, "-o", out]
    subprocess.run(cmd, check=True)
    return out


def blocks(n, amp, count=64, seed=7):
    """Residual blocks like an intra prediction leaves them: a smooth ramp (what the prediction missed) + noise."""
    rng = np.random.default_rng(seed)
    out = np.zeros((count, 1024), np.int16)
    yy, xx = np.mgrid[0:n, 0:n] / float(n)
    for b in range(count):
        g = rng.normal(0, amp, 3)
        r = g[0] * (xx - 0.5) * 2 + g[1] * (yy - 0.5) * 2 + g[2] * 0.5 + rng.normal(0, amp * 0.6, (n, n))
        out[b, :n * n] = np.clip(np.rint(r), -255, 255).astype(np.int16).ravel()
    return out


DATA = os.path.join(ROOT, "tools", "data", "blocks_q32.npz")       # residual blocks of real TU codings (tools/capture_blocks.py)


def run(lib, n, what, active, amp, reps, groups=256, comp=0, mode=1, qp=32):
    cfg = hevcdl_amd.Config()
    hevcdl_amd.load_library().hevcdl_config_default(ctypes.byref(cfg), 64, 64, qp)
    consts = (ctypes.c_double * 12)(cfg.lambda_, cfg.sqrt_lambda, cfg.chroma_weight, cfg.lambda_chroma, *[cfg.err_scale[a][b] for a in range(2) for b in range(4)])
    sbh = (ctypes.c_longlong * 2)(cfg.sbh_rd_factor[0], cfg.sbh_rd_factor[1])
    res = blocks(n, amp) if amp > 0 else np.ascontiguousarray(np.load(DATA)["n%d" % n])
    nw = lib.hevcdl_rd_waves_per_group()
    out = np.zeros(groups * nw * 2 + 16, np.uint64)
    rc = lib.hevcdl_micro_run(consts, sbh, qp, cfg.qp_chroma, res.ctypes.data_as(ctypes.c_void_p), res.shape[0], n, comp, mode, reps, what | (active << 8), groups, out.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0, rc
    global PHASES
    PHASES = out[groups * nw * 2:groups * nw * 2 + 11].astype(np.float64) / reps
    o = out[:groups * nw * 2].reshape(-1, 2)
    live = o[:, 0] > 0
    return float(o[live, 0].mean()) / reps, float(o[live, 1].mean()) / reps, int(o[:, 1].sum() % (1 << 32))


if __name__ == "__main__":
    args = sys.argv[1:]
    lib_path = LIB
    amp, reps = 0.0, 200          # amp 0: the captured blocks
    for i, a in enumerate(args):
        if a == "--lib": lib_path = args[i + 1]
        if a == "--amp": amp = float(args[i + 1])
        if a == "--reps": reps = int(args[i + 1])
    if "build" in args:
        build(lib_path)
        if len(args) == 1: sys.exit(0)
    lib = ctypes.CDLL(lib_path)
    lib.hevcdl_micro_run.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p] + [ctypes.c_int] * 7 + [ctypes.c_void_p]
    names = {0: "rdoq", 1: "bits", 2: "fwd+rdoq+bits+inv"}
    print("lib %s  amp %.1f  reps %d   (cycles per call; sum = mean of abs_sum + bits, a checksum of the results)" % (os.path.basename(lib_path), amp, reps))
    for n in (4, 8, 16, 32):
        for what in (0, 1, 2):
            c8, s8, ck8 = run(lib, n, what, 0, amp, reps)
            c1, s1, ck1 = run(lib, n, what, 1, amp, reps)
            nw = lib.hevcdl_rd_waves_per_group()
            print("n %2d  %-18s  %d waves/CU %8.0f   1 wave/CU %8.0f   sum %9.1f  check %08x   codings per Mcycle and CU at full occupancy %.1f" % (n, names[what], nw, c8, c1, s8, ck8, 1e6 * nw / c8), flush=True)
            if "--phases" in args and what == 0:      # -DHEVCDL_MICRO_T build: cycles per call and phase of rdoq_wave (one wave per CU, workgroup 0)
                pn = ["A: rounded levels", "zero tail", "B: per-position", "B: level walk", "B: sums + group test", "B: batch end", "(B exit)", "C: last position", "signs + hiding"]
                print("      " + "   ".join("%s %.0f" % (pn[i], PHASES[i]) for i in range(9)), flush=True)
