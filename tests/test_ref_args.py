"""Container only (needs /root/reference and the reference build oracle/_ref): oracle/ref_args.py -- the reference's configuration as
command-line switches, used by bench.py's CPU baseline on the GPU box -- drives the reference encoder to exactly the run its own cfg file gives."""
import os
import shutil
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_ENC = os.path.join(ROOT, "oracle", "_ref", "TAppEncoder_ref")


@pytest.mark.skipif(not (os.path.exists("/root/reference/encoder_intra_main.cfg") and os.path.exists(REF_ENC)), reason="reference sources / build not present")
@pytest.mark.parametrize("w,h,nf,qp,bd", [(192, 128, 2, 32, 8), (136, 72, 1, 24, 10)])
def test_switches_equal_the_reference_cfg(w, h, nf, qp, bd):
    import ref_args
    import ref_tools as rt
    yuv = rt.synth_yuv(w, h, nf, 5)
    if bd == 10:
        yuv = yuv.astype(np.uint16) * 4 + 1
    lab = rt.make_labels(w, h, nf, "rand", 6)
    dump, out, bits, recon = rt.run_reference(yuv, w, h, qp, lab, extra_args=["--SEIDecodedPictureHash=1"], bit_depth=bd)
    d = tempfile.mkdtemp(prefix="hmargs_")
    try:
        os.makedirs(os.path.join(d, "rec"))
        yuv.astype(np.uint8 if bd == 8 else "<u2").tofile(os.path.join(d, "in.yuv"))
        for f in range(nf):
            os.makedirs(os.path.join(d, "pred", str(f)))
            for a in range(lab.shape[1]):
                open(os.path.join(d, "pred", str(f), "ctu%d.txt" % a), "w").write(" ".join(str(int(v)) for v in lab[f, a]) + " ")
        env = dict(os.environ, HEVCDL_DUMP=os.path.join(d, "dump.bin"))
        if bd != 8:
            env["HEVCDL_DUMP16"] = "1"
        cmd = [REF_ENC, "-i", "in.yuv", "-b", "rec/str.bin", "-o", "rec/rec.yuv", "--SEIDecodedPictureHash=1"] + ref_args.reference_args(w, h, nf, qp, bd)
        p = subprocess.run(cmd, cwd=d, env=env, capture_output=True, text=True)
        assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
        assert open(os.path.join(d, "rec", "str.bin"), "rb").read() == bits
        assert open(os.path.join(d, "rec", "rec.yuv"), "rb").read() == recon
        assert np.fromfile(env["HEVCDL_DUMP"], dtype=rt.DUMP_DTYPE if bd == 8 else rt.DUMP16_DTYPE).tobytes() == dump.tobytes()
    finally:
        shutil.rmtree(d)
