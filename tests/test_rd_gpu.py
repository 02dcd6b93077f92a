"""GPU parity of the HIP CTU-decision kernel (through the C ABI) against the reference-generated golden records
(bit-exact: CU depth, partition, intra modes, TU tree, cbf, transform-skip flags, coefficients, bits/dist/cost,
pre-loop-filter reconstruction) and against the plain-C oracle on further seeded inputs."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLD, fixture_tiles, fixture_wavefront

pytestmark = pytest.mark.gpu
FIELDS = ["depth", "part_size", "luma_dir", "chroma_dir", "tr_idx", "cbf", "tskip", "bits", "dist", "cost", "coeff_y", "coeff_cb", "coeff_cr"]


def ctu_blocks(recon_frame, w, h, addr):
    import ref_tools
    return ref_tools.ctu_recon_from_frame(recon_frame, w, h, addr)


def assert_records_equal(recs, ref, what):
    for k in FIELDS:
        if not np.array_equal(recs[k], ref[k]):
            bad = np.argwhere(np.asarray(recs[k] != ref[k]).reshape(recs.shape[0], recs.shape[1], -1).any(axis=2))
            raise AssertionError("%s: field %s differs at (frame, ctu) %s" % (what, k, bad[:5].tolist()))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "rd_*.npz"))), ids=lambda p: os.path.basename(p)[3:-4])
def test_golden_records_bit_exact(path):
    import hevcdl_amd
    f = np.load(path)
    w, h, qp = int(f["width"]), int(f["height"]), int(f["qp"])
    yuv, labels, ref = f["yuv"], f["labels"], f["records"]
    tiles = fixture_tiles(f)                                                        # rd_t* / rd_n*: reference runs with tiles enabled
    bd = int(f["bit_depth"]) if "bit_depth" in f.files else 8                       # rd_x*: InternalBitDepth 10 (uint16 samples)
    tools = int(f["tools"]) if "tools" in f.files else hevcdl_amd.TOOLS_REFERENCE   # rd_k*: reference runs with TransformSkip / SignHideFlag / StrongIntraSmoothing / FastUDIUseMPMEnabled off
    enc = hevcdl_amd.Encoder(w, h, qp, max_frames=yuv.shape[0], tiles=tiles, bit_depth=bd, tools=tools, wavefront=fixture_wavefront(f))      # rd_w*: reference runs with WaveFrontSynchro 1
    recs, recon, stats = enc.compress_frames(yuv, labels)
    launch = enc.last_rd_launch()
    enc.close()
    # the builds that carry the timed configurations are compiled for the reference cfg's tools; other tool masks run the build that reads them (csrc/rd_kernel_tools.hip)
    assert launch.split(" ")[0] == ("hevcdl_rd_frame_kernel_bd10" if bd != 8 else ("hevcdl_rd_frame_kernel_tools" if tools != hevcdl_amd.TOOLS_REFERENCE else "hevcdl_rd_frame_kernel")), launch
    assert recon.dtype == (np.uint8 if bd == 8 else np.uint16)
    assert_records_equal(recs, ref, os.path.basename(path))
    for fr in range(yuv.shape[0]):
        for a in range(labels.shape[1]):
            y, u, v = ctu_blocks(recon[fr], w, h, a)
            assert np.array_equal(y, f["rec_y"][fr, a]) and np.array_equal(u, f["rec_cb"][fr, a]) and np.array_equal(v, f["rec_cr"][fr, a]), (fr, a)


def test_workspace_reserved_up_front_changes_nothing_and_names_the_launch():
    """hevcdl_reserve_workspace (the CLI calls it right behind hevcdl_create so that a lack of device memory shows up before pictures are in flight): the largest workspace
    of the context is allocated at once; the launches behind it give the fixture's records, and hevcdl_last_rd_launch names the build and the form the library chose."""
    import hevcdl_amd
    f = np.load(os.path.join(GOLD, "rd_c192_q32_r2.npz"))
    w, h, qp = int(f["width"]), int(f["height"]), int(f["qp"])
    enc = hevcdl_amd.Encoder(w, h, qp, max_frames=2)
    assert enc.last_rd_launch() == ""
    enc.reserve_workspace()
    enc.reserve_workspace()                                # (a second call finds it there)
    recs, recon, stats = enc.compress_frames(f["yuv"], f["labels"])
    info = enc.last_rd_launch()
    enc.close()
    assert_records_equal(recs, f["records"], "reserved workspace")
    assert info.startswith("hevcdl_rd_frame_kernel form=") and "units=2" in info and "waves=8" in info, info


@pytest.mark.parametrize("flags", [2, 4], ids=["ten-wave-build", "eight-wave-build"])
@pytest.mark.parametrize("path", [p for p in sorted(glob.glob(os.path.join(GOLD, "rd_*.npz"))) if not os.path.basename(p).startswith(("rd_x", "rd_k")) and "_b10" not in os.path.basename(p)], ids=lambda p: os.path.basename(p)[3:-4])
def test_golden_records_bit_exact_on_either_build_of_the_kernel(path, flags):
    """The 8-bit decision kernel exists in two builds (8 wavefronts per workgroup with the look-ahead of the few-units form; 10 without it, csrc/rd_kernel_wide.hip) and the
    library picks by the shape of the launch -- small fixtures would only ever meet one of them.  exec_flags HEVCDL_EXEC_RD_WIDE (2) / HEVCDL_EXEC_RD_NARROW (4) force a build:
    both must reproduce every reference fixture."""
    import hevcdl_amd
    f = np.load(path)
    w, h, qp = int(f["width"]), int(f["height"]), int(f["qp"])
    yuv, labels, ref = f["yuv"], f["labels"], f["records"]
    cfg = hevcdl_amd.default_config(w, h, qp, max_frames=yuv.shape[0], tiles=fixture_tiles(f), tools=int(f["tools"]) if "tools" in f.files else hevcdl_amd.TOOLS_REFERENCE, wavefront=fixture_wavefront(f))
    cfg.exec_flags = flags
    enc = hevcdl_amd.Encoder(w, h, qp, cfg=cfg)
    recs, recon, stats = enc.compress_frames(yuv, labels)
    enc.close()
    assert_records_equal(recs, ref, os.path.basename(path))
    for fr in range(yuv.shape[0]):
        for a in range(labels.shape[1]):
            y, u, v = ctu_blocks(recon[fr], w, h, a)
            assert np.array_equal(y, f["rec_y"][fr, a]) and np.array_equal(u, f["rec_cb"][fr, a]) and np.array_equal(v, f["rec_cr"][fr, a]), (fr, a)


def test_whole_1080p_frame_matches_the_reference_golden():
    """Full-size parity against the reference itself: one 1920x1080 frame (510 CTUs), records bit for bit and the reconstruction of every CTU
    by checksum (tests/golden/full_f1080_q32.npz: a run of the reference encoder, oracle/gen_fixtures.py gen_full)."""
    import hevcdl_amd
    import ref_tools
    from test_oracle_golden import full_frame_crc
    f = np.load(os.path.join(GOLD, "full_f1080_q32.npz"))
    w, h, qp = int(f["width"]), int(f["height"]), int(f["qp"])
    yuv = ref_tools.synth_yuv(w, h, 1, int(f["seed"]))
    enc = hevcdl_amd.Encoder(w, h, qp, max_frames=1)
    recs, recon, stats = enc.compress_frames(yuv, f["labels"])
    gpu_labels, gpu_logits = enc.predict_depth(yuv, want_logits=True)
    enc.close()
    assert_records_equal(recs, f["records"], "full_f1080_q32")
    assert np.array_equal(full_frame_crc(recon[0], w, h, recs.shape[1]), f["recon_crc32"])
    # the fixture's labels are the numpy fp32 CNN's (report, do not just threshold): the device CNN -- split-f16 operands on the
    # matrix cores -- may only differ where two logits of the fp32 graph are closer than its own error (tests/test_cnn_gpu.py bounds that at 1e-3)
    diff_cells = int((gpu_labels != f["labels"]).sum()); diff_ctus = int((gpu_labels != f["labels"]).any(axis=2).sum())
    print("CNN labels, device vs fp32 numpy oracle on full_f1080_q32: %d of %d CTUs differ (%d of %d cells)" % (diff_ctus, gpu_labels.shape[1], diff_cells, gpu_labels.size))
    # no allowance by count: every CTU that carries another label must be a tie of the fp32 graph itself -- the oracle's own logits of that CTU (computed here) agree with the
    # device's within the tolerance of tests/test_cnn_gpu.py, and two classes of one of its digits lie closer together than twice that
    import cnn_oracle
    wts = cnn_oracle.load_weights(hevcdl_amd.WEIGHTS_PATH)
    ctus_rgb = cnn_oracle.yuv_to_rgb_ctus(yuv[0], w, h)
    for a in np.flatnonzero((gpu_labels[0] != f["labels"][0]).any(axis=1)):
        lg = cnn_oracle.ctu_logits(wts, ctus_rgb[a:a + 1])[0]
        assert np.abs(lg - gpu_logits[0, a]).max() < 1e-3, a
        top2 = np.sort(lg.reshape(4, 4, 4), axis=-1)
        assert (top2[..., 3] - top2[..., 2]).min() < 2e-3, "CTU %d carries another label than the fp32 oracle's without a tie in its logits" % a
    assert diff_ctus <= 2, "%d CTUs are ties of the fp32 graph: the fixture no longer tests what it was made for" % diff_ctus
    # (round 5: the one CTU that differed until then was no tie -- the fixture's labels predated the oracle's restatement of use_model.py:101-119 for a quadrant that
    #  answers "0000"; regenerated, the device's labels are the fixture's on all 510 CTUs)


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(GOLD), "..", "oracle", "_ref", "TAppEncoder_ref")), reason="reference build (oracle/_ref) not present")
def test_whole_1080p_frame_matches_a_live_reference_run():
    """The reference encoder itself, run now on this machine with the labels the device CNN gives (oracle/_ref travels with the repository):
    a fresh seed per run of the suite would be nondeterministic, so the seed is fixed but differs from the golden's."""
    import hevcdl_amd
    import ref_tools
    w, h, qp = 1920, 1080, 27
    yuv = ref_tools.synth_yuv(w, h, 1, 1234)
    enc = hevcdl_amd.Encoder(w, h, qp, max_frames=1)
    labels = enc.predict_depth(yuv)
    recs, recon, stats = enc.compress_frames(yuv, labels)
    enc.close()
    sys_path_bench = os.path.join(os.path.dirname(GOLD), "..")
    import sys
    sys.path.insert(0, sys_path_bench)
    import bench
    wall, per, dumps = bench.run_reference_pictures([yuv[0]], labels, w, h, qp, 1, dump=True)
    res = bench.parity_against_dumps(dumps, recs, [recon[0]], w, h)
    assert res["ctus"] == recs.shape[1] and res["mismatches"] == 0, res


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(GOLD), "..", "oracle", "_ref", "TAppEncoder_ref")), reason="reference build (oracle/_ref) not present")
def test_whole_2160p_frame_matches_a_live_reference_run():
    """C4's picture size, every CTU: all 34 CTU rows of one 3840x2160 frame, the last one half outside the picture (forced splits, TEncCu.cpp:574-576),
    against the reference encoder run now on this machine with the device CNN's labels -- every field of the 2040 CTU records and the reconstruction."""
    import sys
    import hevcdl_amd
    import ref_tools
    sys.path.insert(0, os.path.join(os.path.dirname(GOLD), ".."))
    import bench
    w, h, qp = 3840, 2160, 32
    yuv = ref_tools.synth_yuv(w, h, 1, 777)
    enc = hevcdl_amd.Encoder(w, h, qp, max_frames=1)
    labels = enc.predict_depth(yuv)
    recs, recon, stats = enc.compress_frames(yuv, labels)
    enc.close()
    wall, per, dumps = bench.run_reference_pictures([yuv[0]], labels, w, h, qp, 1, dump=True)
    res = bench.parity_against_dumps(dumps, recs, [recon[0]], w, h)
    assert res["ctus"] == 2040 and res["mismatches"] == 0, res


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(GOLD), "..", "oracle", "_ref", "TAppEncoder_ref")), reason="reference build (oracle/_ref) not present")
def test_c4_regime_launch_matches_live_reference_runs():
    """The regime of the timed job of bench.py inside the suite: MORE frames than CUs in one launch (320 frames of 3840x384: 2-3 masters per workgroup,
    frames travelling between workgroups, second passes left pending), labels from the device CNN -- a sample of its frames against the reference encoder
    run now on this machine (CTU records and reconstruction, bit for bit), and every frame against a second launch of the same frames in the independent
    form of the other build of the kernel."""
    import sys
    import hevcdl_amd
    import ref_tools
    sys.path.insert(0, os.path.join(os.path.dirname(GOLD), ".."))
    import bench
    w, h, qp, nf = 3840, 384, 32, 320
    base = ref_tools.synth_yuv(w, h, 8, 4242)
    rng = np.random.default_rng(17)
    yuv = np.stack([np.clip(base[i % 8].astype(np.int16) + rng.integers(-2, 3, base.shape[1]) * (i // 8 % 3), 0, 255).astype(np.uint8) for i in range(nf)])
    enc = hevcdl_amd.Encoder(w, h, qp, max_frames=nf)
    labels = enc.predict_depth(yuv)
    recs, recon, stats = enc.compress_frames(yuv, labels)
    enc.close()
    pick = [0, 1, 127, 128, 255, 256, 257, 319]             # first / last frames of the workgroups' lists, frames that start as surplus units
    wall, per, dumps = bench.run_reference_pictures([yuv[i] for i in pick], labels[pick], w, h, qp, len(pick), dump=True)
    res = bench.parity_against_dumps(dumps, recs[pick], [recon[i] for i in pick], w, h)
    assert res["ctus"] == len(pick) * recs.shape[1] and res["mismatches"] == 0, res
    cfg = hevcdl_amd.default_config(w, h, qp, max_frames=nf)
    cfg.exec_flags = 1 | 2                                   # no hand-over between workgroups, the ten-wave build
    enc = hevcdl_amd.Encoder(w, h, qp, cfg=cfg)
    recs2, recon2, stats2 = enc.compress_frames(yuv, labels)
    enc.close()
    assert_records_equal(recs, recs2, "C4-regime launch vs independent form")
    assert np.array_equal(recon, recon2) and np.array_equal(stats["est_bits"], stats2["est_bits"])


def test_wavefront_rows_on_waves_of_their_own_and_on_one_wave_give_the_oracle(oracle_built):
    """WaveFrontSynchro 1 (hevcdl_config.wavefront): a CTU row is a unit of the decision kernel -- the rows of a frame run two CTUs apart on different waves, most of them
    on different workgroups, and wait for each other through finished-CTU counts in HBM; HEVCDL_EXEC_NO_UNIT_HANDOVER selects the form that needs no co-residency (one wave
    walks a frame's rows in order).  Both forms, frames of 13 x 7 CTUs (more rows than a workgroup has waves, a ragged last row and column), against the oracle with the key
    set: records, reconstruction, estimated bits; and the two differ from the run without the key (the test is not vacuous)."""
    import hevcdl_amd
    import ref_tools
    w, h, qp, nf = 808, 424, 32, 3
    yuv = ref_tools.synth_yuv(w, h, nf, 909)
    enc = hevcdl_amd.Encoder(w, h, qp, max_frames=nf)
    labels = enc.predict_depth(yuv)
    plain, _, _ = enc.compress_frames(yuv, labels)
    enc.close()
    o_recs, o_recon, o_stats = ref_tools.run_oracle(yuv, w, h, qp, labels, wpp=True)
    for flags, form in ((0, "form=wavefront-rows"), (hevcdl_amd.EXEC_NO_UNIT_HANDOVER, "form=wavefront(one wave per frame)"), (hevcdl_amd.EXEC_RD_WIDE, "form=wavefront-rows"), (hevcdl_amd.EXEC_RD_NARROW, "form=wavefront-rows")):
        cfg = hevcdl_amd.default_config(w, h, qp, max_frames=nf, wavefront=True)
        cfg.exec_flags = flags
        enc = hevcdl_amd.Encoder(w, h, qp, cfg=cfg)
        recs, recon, stats = enc.compress_frames(yuv, labels)
        launch = enc.last_rd_launch()
        enc.close()
        assert form in launch and ("units=%d" % (nf * 7 if "rows" in form else nf)) in launch, launch
        assert_records_equal(recs, o_recs, "wavefront, exec_flags %d" % flags)
        assert np.array_equal(recon, o_recon) and np.array_equal(stats["est_bits"], o_stats["est_bits"]) and np.array_equal(stats["sse"], o_stats["sse"]), flags
    assert any(not np.array_equal(plain[k], o_recs[k]) for k in ref_tools.FIELDS)
    # what the library refuses: the key together with tiles (as the reference)
    with pytest.raises(hevcdl_amd.HevcdlError):
        hevcdl_amd.Encoder(832, 448, qp, max_frames=1, wavefront=True, tiles=(2, 1))


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(GOLD), "..", "oracle", "_ref", "TAppEncoder_ref")), reason="reference build (oracle/_ref) not present")
@pytest.mark.parametrize("size", [(1920, 1080, 27, 3), (3840, 2160, 32, 2)], ids=["1080p", "2160p"])
def test_wavefront_whole_frames_match_a_live_reference_run(size):
    """WaveFrontSynchro 1 at the sizes of C2 / C4: whole frames (17 / 34 CTU rows a frame, each a unit of the launch, the few-units form on top at these counts) against the
    reference encoder run now on this machine with --WaveFrontSynchro=1 and the device CNN's labels -- every field of every CTU record and the reconstruction."""
    import sys
    import hevcdl_amd
    import ref_tools
    sys.path.insert(0, os.path.join(os.path.dirname(GOLD), ".."))
    import bench
    w, h, qp, nf = size
    yuv = ref_tools.synth_yuv(w, h, nf, 1300 + qp)
    enc = hevcdl_amd.Encoder(w, h, qp, max_frames=nf, wavefront=True)
    labels = enc.predict_depth(yuv)
    recs, recon, stats = enc.compress_frames(yuv, labels)
    launch = enc.last_rd_launch()
    enc.close()
    assert "form=wavefront-rows" in launch and "units=%d" % (nf * ((h + 63) // 64)) in launch, launch
    wall, per, dumps = bench.run_reference_pictures(list(yuv), labels, w, h, qp, nf, dump=True, wavefront=1)
    res = bench.parity_against_dumps(dumps, recs, list(recon), w, h)
    assert res["ctus"] == nf * recs.shape[1] and res["mismatches"] == 0, res


def test_wavefront_many_units_launch_equals_the_one_wave_form():
    """The regime of a 600-frame wavefront job in miniature: more row units than the launch has wave slots (400 frames of 832x448 = 2800 rows on 256 workgroups of ten waves),
    so units queue behind each other on a slot while the rows they depend on are still walked elsewhere -- against the form in which one wave walks a frame's rows in order
    (byte for byte), which the test above pins to the oracle.  Then two frames on the same context: few units, the idle workgroups take the second passes the rows post."""
    import hevcdl_amd
    import ref_tools
    w, h, qp, nf = 832, 448, 32, 400
    base = ref_tools.synth_yuv(w, h, 6, 4343)
    rng = np.random.default_rng(23)
    yuv = np.stack([np.clip(base[i % 6].astype(np.int16) + rng.integers(-2, 3, base.shape[1]) * (i // 6 % 3), 0, 255).astype(np.uint8) for i in range(nf)])
    cfg = hevcdl_amd.default_config(w, h, qp, max_frames=nf, wavefront=True)
    cfg.exec_flags = hevcdl_amd.EXEC_NO_UNIT_HANDOVER
    enc = hevcdl_amd.Encoder(w, h, qp, cfg=cfg)
    labels = enc.predict_depth(yuv)
    ref_recs, ref_recon, ref_stats = enc.compress_frames(yuv, labels)
    enc.close()
    enc = hevcdl_amd.Encoder(w, h, qp, max_frames=nf, wavefront=True)
    recs, recon, stats = enc.compress_frames(yuv, labels)
    launch = enc.last_rd_launch()
    # a launch of fewer frames on the same context: few units, the idle workgroups take posted second passes
    recs8, recon8, stats8 = enc.compress_frames(yuv[:2], labels[:2])
    launch8 = enc.last_rd_launch()
    enc.close()
    assert "form=wavefront-rows" in launch and "units=2800" in launch and "form=wavefront-rows+few-units" in launch8 and "units=14" in launch8, (launch, launch8)
    assert_records_equal(recs, ref_recs, "wavefront rows vs one wave per frame")
    assert np.array_equal(recon, ref_recon) and np.array_equal(stats["est_bits"], ref_stats["est_bits"])
    assert_records_equal(recs8, ref_recs[:2], "wavefront rows, few units")
    assert np.array_equal(recon8, ref_recon[:2])


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(GOLD), "..", "oracle", "_ref", "TAppEncoder_ref")), reason="reference build (oracle/_ref) not present")
def test_first_label_zero_codes_one_cu_like_the_reference():
    """The reference's walk reads ONE label per CU (its top-left cell, TEncCu.cpp:496-520) and use_model.py:101-119 does emit label sets whose first label is
    0 while later quadrants are not: HM then codes one 64x64 CU.  Caller labels of that kind must reach the search as they are (the boundary policy only
    repairs what the walk would otherwise leave undecided): records and reconstruction equal the reference encoder's, run now with the same label files."""
    import sys
    import hevcdl_amd
    import ref_tools
    sys.path.insert(0, os.path.join(os.path.dirname(GOLD), ".."))
    import bench
    w, h, qp = 192, 128, 30
    yuv = ref_tools.synth_yuv(w, h, 1, 31)
    labels = np.zeros((1, 6, 16), np.uint8)
    labels[0, 0] = [0, 0, 2, 3, 0, 0, 3, 2, 1, 1, 2, 2, 1, 1, 2, 2]          # first label 0: one 64x64 CU, whatever follows
    labels[0, 1] = [1, 1, 2, 2, 1, 1, 2, 2, 1, 1, 3, 3, 1, 1, 3, 3]
    labels[0, 2] = [0, 0, 1, 1, 0, 0, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2]
    labels[0, 3] = [2, 2, 1, 1, 2, 3, 1, 1, 1, 1, 2, 2, 1, 1, 2, 2]
    labels[0, 4] = [0] * 16
    labels[0, 5] = [1, 0, 1, 3, 0, 0, 2, 0, 1, 2, 1, 0, 3, 0, 0, 0]          # valid for the walk (first labels 1, 1, 1, 1), ragged elsewhere
    enc = hevcdl_amd.Encoder(w, h, qp, max_frames=1)
    recs, recon, stats = enc.compress_frames(yuv, labels)
    enc.close()
    assert recs["depth"][0, 0].max() == 0 and recs["depth"][0, 2].max() == 0 and recs["depth"][0, 5].max() == 1
    wall, per, dumps = bench.run_reference_pictures([yuv[0]], labels, w, h, qp, 1, dump=True)
    res = bench.parity_against_dumps(dumps, recs, [recon[0]], w, h)
    assert res["ctus"] == 6 and res["mismatches"] == 0, res


def test_bench_two_ranks_on_one_gpu_give_the_single_rank_line(tmp_path):
    """bench.py's own N > 1 path (frame shards, max-over-ranks timing, gather of the per-frame rate records) under torch.distributed.run with two ranks
    sharing this GPU (HEVCDL_BENCH_BACKEND=gloo): the job's rate records must add up to the single-rank line's, and the line carries the latency floor."""
    import json
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(GOLD), "..")
    common = ["--steps", "1", "--warmup", "0", "--width", "256", "--height", "192", "--frames", "6", "--no-cpu-baseline", "--no-c2", "--no-e2e", "--saturated-frames", "0", "--weak-frames", "4"]
    env = dict(os.environ, HEVCDL_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r1 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"] + common, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r1.returncode == 0, r1.stdout[-1500:] + r1.stderr[-1500:]
    one = json.loads(r1.stdout.strip().splitlines()[-1])
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29517",
                         os.path.join(root, "bench.py"), "--gpus", "2"] + common, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r2.returncode == 0, r2.stdout[-1500:] + r2.stderr[-1500:]
    two = json.loads([l for l in r2.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1 and two["scaling"] == "strong"
    assert two["config"]["frames"] == one["config"]["frames"] == 6 and two["config"]["frames_per_gpu"] == 3
    assert two["est_bits_per_frame"] == one["est_bits_per_frame"] and one["est_bits_per_frame"] > 0
    # the collective spans both ranks (every rank adds one in an all-reduce: the driver's SCALE record can be checked by this key) and every frame's record arrived
    assert two["rccl_ranks"] == 2 and one["rccl_ranks"] == 1 and two["frames_gathered"] == 6 and two["collective_backend"] == "gloo"
    assert one["roofline"]["launch"].startswith(one["roofline"]["kernel"] + " form=") and "units=6" in one["roofline"]["launch"] and "units=3" in two["roofline"]["launch"]
    # weak scaling beside the strong headline: every rank ran a 4-frame step of its own (N > 1 only)
    assert "weak" not in one and two["weak"]["n_gpus"] == 2 and two["weak"]["frames_per_gpu"] == 4 and two["weak"]["value"] > 0 and "units=4" in two["weak"]["launch"]
    # ... and the job once more with WaveFrontSynchro 1 (extra key): the ranks' shares with CTU rows as units
    assert two["wavefront"]["n_gpus"] == 2 and two["wavefront"]["value"] > 0 and "form=wavefront-rows" in two["wavefront"]["launch"] and "units=9" in two["wavefront"]["launch"]
    assert "issue" in one["roofline"]                            # the instruction-issue bound (None unless a counter pass of this kernel source and launch shape is committed)
    for line in (one, two):
        assert line["value"] > 0 and line["latency_floor_s"] > 0 and line["strong_scaling_ceiling"]["value"] > 0 and "roofline" in line


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(GOLD), "..", "oracle", "_ref", "TAppEncoder_ref")), reason="reference build (oracle/_ref) not present")
@pytest.mark.parametrize("qp", [22, 27, 32, 37])
def test_2160p_wide_bands_match_a_live_reference_run_at_the_sweep_qps(qp):
    """The four QPs of BASELINE.json's C3 sweep at the full 3840 width: the top 3840x384 band (6 CTU rows, 360 CTUs) of three 2160p frames,
    coded as pictures of their own by the reference encoder (run now, labels of the device CNN) and by the decision kernel -- every field of
    every CTU record and the reconstruction."""
    import sys
    import hevcdl_amd
    import ref_tools
    sys.path.insert(0, os.path.join(os.path.dirname(GOLD), ".."))
    import bench
    w, h, nf = 3840, 384, 3
    full = ref_tools.synth_yuv(3840, 2160, nf, 500 + qp)
    ysz, csz = 3840 * 2160, 1920 * 1080
    band = np.concatenate([full[:, :w * h], full[:, ysz:ysz + (w // 2) * (h // 2)], full[:, ysz + csz:ysz + csz + (w // 2) * (h // 2)]], axis=1)
    enc = hevcdl_amd.Encoder(w, h, qp, max_frames=nf)
    labels = enc.predict_depth(band)
    recs, recon, stats = enc.compress_frames(band, labels)
    enc.close()
    wall, per, dumps = bench.run_reference_pictures([band[i] for i in range(nf)], labels, w, h, qp, nf, dump=True, workers=min(nf, bench.effective_cores()))
    res = bench.parity_against_dumps(dumps, recs, [recon[i] for i in range(nf)], w, h)
    assert res["ctus"] == nf * 360 and res["mismatches"] == 0, res


def test_unclamped_caller_labels_get_the_boundary_policy(oracle_built):
    """Label files of the reference's own label producer are not clamped at the picture border.  Caller-supplied
    labels therefore go through the boundary policy before the search: raw random labels on 200x136 give exactly the result of their clamped
    form (no CU left undecided, SIZE_NONE only outside the picture), and a depth above 3 is rejected."""
    import hevcdl_amd
    import ref_tools
    import cnn_oracle
    w, h, qp = 200, 136, 30
    yuv = ref_tools.synth_yuv(w, h, 1, 91)
    enc = hevcdl_amd.Encoder(w, h, qp, max_frames=1)
    raw = np.random.default_rng(5).integers(0, 4, (1, enc.ctus, 16)).astype(np.uint8)
    clamped = cnn_oracle.clamp_labels(raw.copy(), w, h)
    assert not np.array_equal(raw, clamped)
    r1, rec1, s1 = enc.compress_frames(yuv, raw)
    r2, rec2, s2 = enc.compress_frames(yuv, clamped)
    o_recs, o_recon, o_stats = ref_tools.run_oracle(yuv, w, h, qp, clamped)
    assert_records_equal(r1, r2, "raw vs clamped labels")
    assert_records_equal(r1, o_recs, "clamped labels vs oracle")
    assert np.array_equal(rec1, rec2) and np.array_equal(rec1, o_recon)
    for k in ("encode_pictures", "begin_frames"):
        a = getattr(enc, k)(yuv, raw)
        b = getattr(enc, k)(yuv, clamped)
        assert np.array_equal(a[0] if isinstance(a, tuple) else a, b[0] if isinstance(b, tuple) else b), k
    bad = raw.copy(); bad[0, 3, 7] = 4
    with pytest.raises(hevcdl_amd.HevcdlError):
        enc.compress_frames(yuv, bad)
    enc.close()


@pytest.mark.parametrize("chunk,sao", [(3, True), (16, True), (5, False)])
def test_chunked_hand_over_equals_the_whole_batch_call(chunk, sao):
    """hevcdl_encode_pictures_chunked hands the results of the same device batch over chunk by chunk (two pinned buffers, a copy stream): every
    chunk must hold, picture for picture, what hevcdl_encode_pictures writes into the caller's buffers -- ragged last chunk, chunk larger than the
    batch, CNN labels (labels == None) and no SAO included."""
    import hevcdl_amd
    import ref_tools
    w, h, qp, nf = 200, 136, 31, 11
    yuv = ref_tools.synth_yuv(w, h, nf, 17)
    enc = hevcdl_amd.Encoder(w, h, qp, max_frames=nf)
    recs, pics, params, stats = enc.encode_pictures(yuv, None, sao=sao)
    chunks = enc.encode_pictures_chunked(yuv, None, sao=sao, chunk_frames=chunk)
    enc.close()
    assert [c[0] for c in chunks] == list(range(0, nf, chunk))
    assert sum(c[1].shape[0] for c in chunks) == nf
    for first, r, p, sp, st in chunks:
        n = r.shape[0]
        assert_records_equal(r, recs[first:first + n], "chunk at %d" % first)
        assert np.array_equal(p, pics[first:first + n])
        assert (sp is None) == (not sao)
        if sao:
            assert sp.tobytes() == params[first:first + n].tobytes()
        for k in ("sse", "est_bits", "ctus"):
            assert np.array_equal(st[k], stats[k][first:first + n]), k


@pytest.mark.parametrize("nf,tiles", [(1, (1, 1)), (6, (1, 1)), (20, (1, 1)), (150, (1, 1)), (3, (2, 2))])
def test_few_units_form_equals_the_independent_form(nf, tiles):
    """A launch of at most two thirds as many units as CUs runs on ALL CUs: the workgroups without a unit take the second luma passes (with very few
    units also the chroma modes; with more units than takers only while a taker is free: the 150-frame case) the others post through HBM.  It must give, byte for byte, what the independent form gives (exec_flags
    HEVCDL_EXEC_NO_UNIT_HANDOVER: one workgroup per unit, nothing crosses workgroups) -- records, reconstruction and statistics."""
    import hevcdl_amd
    import ref_tools
    w, h, qp = 512, 320, 30
    base = ref_tools.synth_yuv(w, h, 4, 123)
    rng = np.random.default_rng(9)
    yuv = np.stack([np.clip(base[i % 4].astype(np.int16) + rng.integers(-3, 4, base.shape[1]) * (i // 4), 0, 255).astype(np.uint8) for i in range(nf)])
    out = []
    for flags in (0, 1, 1 | 4, 2):                          # few-units form (8 waves) | independent, the library's choice (10 waves) | independent, 8 waves | 10 waves forced
        cfg = hevcdl_amd.default_config(w, h, qp, max_frames=nf, tiles=tiles)
        cfg.exec_flags = flags
        enc = hevcdl_amd.Encoder(w, h, qp, cfg=cfg)
        labels = enc.predict_depth(yuv)
        out.append(enc.compress_frames(yuv, labels))
        enc.close()
    for o in out[1:]:
        assert_records_equal(out[0][0], o[0], "few-units form vs independent form")
        assert np.array_equal(out[0][1], o[1])
        for k in ("sse", "est_bits", "ctus"):
            assert np.array_equal(out[0][2][k], o[2][k]), k


@pytest.mark.parametrize("flags", [0, 2], ids=["library-chosen-build", "ten-wave-build-forced"])
def test_units_handed_over_between_workgroups_give_the_same_result(oracle_built, flags):
    """600 frames on 256 workgroups do not divide evenly: the surplus frames travel round the ring of workgroups (a frame is handed over at a CTU
    boundary: position + coder state).  The launch that migrates must give, frame by frame, what launches without migration give (<= one frame
    per workgroup), and a sample of frames must equal the oracle.  exec_flags 2 (HEVCDL_EXEC_RD_WIDE): the same with the ten-wave build of the kernel, a pair the
    library's own choice makes only between three and four units per workgroup (ten waves from three units per workgroup on, hand-over below four) -- 769 frames and more on 256 CUs."""
    import hevcdl_amd
    import ref_tools
    w, h, qp, nf = 512, 256, 33, 600                       # 32 CTUs per frame: two hand-over points per frame
    base = ref_tools.synth_yuv(w, h, 8, 77)
    rng = np.random.default_rng(3)
    yuv = np.stack([np.clip(base[i % 8].astype(np.int16) + rng.integers(-2, 3, base.shape[1]) * (1 + i % 3), 0, 255).astype(np.uint8) for i in range(nf)])
    labels = ref_tools.make_labels(w, h, nf, "rand", 5)
    cfg = hevcdl_amd.default_config(w, h, qp, max_frames=nf)
    cfg.exec_flags = flags
    enc = hevcdl_amd.Encoder(w, h, qp, cfg=cfg)
    recs, recon, stats = enc.compress_frames(yuv, labels)               # 600 units on min(600, CUs) workgroups: migrates
    assert "form=unit-handover" in enc.last_rd_launch() and ("_wide " in enc.last_rd_launch()) == (flags == 2), enc.last_rd_launch()
    parts = [enc.compress_frames(yuv[a:a + 200], labels[a:a + 200]) for a in range(0, nf, 200)]      # 200 units: one per workgroup
    enc.close()
    recs2 = np.concatenate([p[0] for p in parts]); recon2 = np.concatenate([p[1] for p in parts]); stats2 = np.concatenate([p[2] for p in parts])
    assert_records_equal(recs, recs2, "migrating launch vs plain launches")
    assert np.array_equal(recon, recon2)
    for k in ("sse", "est_bits", "ctus"):
        assert np.array_equal(stats[k], stats2[k]), k
    pick = [0, 255, 256, 511, 512, 599]
    o_recs, o_recon, o_stats = ref_tools.run_oracle(yuv[pick], w, h, qp, labels[pick])
    assert_records_equal(recs[pick], o_recs, "migrating launch vs oracle")
    assert np.array_equal(recon[pick], o_recon) and np.array_equal(stats["est_bits"][pick], o_stats["est_bits"])


def _padded_planes(yuv, w, h, dtype, margin, extra):
    """Frames [n][w*h*3/2] -> three plane views [n][rows][cols] inside padded buffers (margin samples round every plane, `extra` more per row),
    the way an encoder's picture buffers hold them (TComPicYuv: margin 80, one int16 per sample)."""
    n = yuv.shape[0]
    views = []
    off = 0
    for c in range(3):
        pw, ph = (w, h) if c == 0 else (w // 2, h // 2)
        buf = np.full((n, ph + 2 * margin, pw + 2 * margin + extra), 77, dtype)
        buf[:, margin:margin + ph, margin:margin + pw] = yuv[:, off:off + pw * ph].reshape(n, ph, pw)
        views.append(buf[:, margin:margin + ph, margin:margin + pw])
        off += pw * ph
    return views


def test_planes_with_their_own_pitch_give_the_packed_result():
    """hevcdl_compress_frames_planes / hevcdl_predict_depth_planes (planes + strides): pictures handed over as three planes
    inside padded buffers -- uint8 with an odd pitch, and HM's own layout (8-bit samples in int16, margin 80: TComPicYuv.cpp:81-104) -- must
    give exactly what the packed entry points give; so must 10-bit samples in uint16 planes; bad strides / sample sizes are refused."""
    import hevcdl_amd
    import ref_tools
    w, h, qp, nf = 200, 136, 30, 2
    yuv = ref_tools.synth_yuv(w, h, nf, seed=71)
    e = hevcdl_amd.Encoder(w, h, qp, max_frames=nf)
    labels = e.predict_depth(yuv)
    recs, recon, stats = e.compress_frames(yuv)
    for dtype, margin, extra in ((np.uint8, 3, 5), (np.int16, 80, 0)):
        y, u, v = _padded_planes(yuv, w, h, dtype, margin, extra)
        assert np.array_equal(e.predict_depth_planes(y, u, v), labels)
        r2, rec2, st2 = e.compress_frames_planes(y, u, v)
        assert r2.tobytes() == recs.tobytes() and np.array_equal(rec2, recon) and st2.tobytes() == stats.tobytes()
        r3, rec3, _ = e.compress_frames_planes(y[1], u[1], v[1], labels=labels[1:])          # one frame as 2-D views, caller's labels
        assert r3.tobytes() == recs[1:].tobytes() and np.array_equal(rec3, recon[1:])
    pl, n, keep = e._planes(*_padded_planes(yuv, w, h, np.uint8, 0, 0))
    pl.row_stride[1] = w // 2 - 1
    out = np.zeros((nf, e.ctus, 16), np.uint8)
    assert e.lib.hevcdl_predict_depth_planes(e._h, hevcdl_amd.ctypes.byref(pl), n, out.ctypes.data, None) == 1      # HEVCDL_ERR_INVALID_ARG
    e.close()
    yuv10 = (ref_tools.synth_yuv(w, h, 1, seed=72).astype(np.uint16) << 2) | 1
    e = hevcdl_amd.Encoder(w, h, qp, max_frames=1, bit_depth=10)
    recs, recon, stats = e.compress_frames(yuv10)
    r2, rec2, _ = e.compress_frames_planes(*_padded_planes(yuv10, w, h, np.uint16, 16, 2))
    assert r2.tobytes() == recs.tobytes() and np.array_equal(rec2.reshape(-1), np.asarray(recon).reshape(-1))
    with pytest.raises(hevcdl_amd.HevcdlError):
        e.compress_frames_planes(*_padded_planes(yuv10.astype(np.uint8), w, h, np.uint8, 0, 0))     # one byte per sample on a 10-bit context
    e.close()


_STAGE_CHILD = """
import ctypes, sys
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import hevcdl_amd, ref_tools
f = np.load(sys.argv[1])
w, h, qp = int(f["width"]), int(f["height"]), int(f["qp"])
yuv = ref_tools.synth_yuv(w, h, 1, int(f["seed"]))
enc = hevcdl_amd.Encoder(w, h, qp, max_frames=1)
recs, recon, stats = enc.compress_frames(yuv, f["labels"])
lib = hevcdl_amd.load_library()
lib.hevcdl_stage_trace_fetch.restype = ctypes.c_size_t
lib.hevcdl_stage_trace_fetch.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
buf = np.zeros(12 << 20, np.uint32)
used = int(lib.hevcdl_stage_trace_fetch(buf.ctypes.data, buf.size))
enc.close()
np.savez(sys.argv[2], words=buf[:min(used, buf.size)], used=used, coeff_y=recs["coeff_y"])
"""


def _stage_sets_of_fixture(f):
    kind, a, b, c, cost, off, blk = (f[k] for k in ("kind", "a", "b", "c", "cost", "blk_off", "blk"))
    lines, tus = set(), set()
    for i in range(len(kind)):
        if kind[i] == 0:
            lines.add((0, int(a[i]), int(b[i]), int(c[i]), float(cost[i])))
        elif kind[i] == 1:
            lines.add((1, int(a[i]), 0, 0, float(cost[i])))
        else:
            tus.add((int(kind[i]), int(a[i]), int(b[i]), blk[off[i]:off[i + 1]].astype(np.int32).tobytes()))
    return lines, tus


def _stage_sets_of_log(words):
    lines, tus, i = set(), set(), 0
    while i < len(words):
        kind, a, b, c = (int(v) for v in words[i:i + 4])
        if kind < 2:
            cost = float("%g" % words[i + 4:i + 6].copy().view(np.float64)[0])      # the 6 significant digits the reference prints
            lines.add((kind, a, b, c, cost))
            i += 6
        else:
            assert kind in (2, 3) and a in (4, 8, 16, 32), (i, kind, a)
            tus.add((kind, a, b, words[i + 6:i + 6 + 3 * a * a].copy().view(np.int32).tobytes()))
            i += 6 + 3 * a * a
    return lines, tus


@pytest.mark.parametrize("name", ["stage_a64_q32", "stage_b128_q27"])
def test_kernel_stages_cover_the_reference_traces(name, tmp_path):
    """F-rd-3 on the device: the stage-trace build of the library (-DHEVCDL_STAGE_TRACE, lib/libhevcdl_hip_trace.so: the decision kernel logs the
    events HM prints under DEBUG_INTRA_SEARCH_COSTS / DEBUG_TRANSFORM_AND_QUANTISE) against the reference's own traces
    (tests/golden/stage_*.npz, oracle/gen_fixtures.py gen_stage_traces).  Waves work on a CTU concurrently and speculatively and a repeated
    evaluation is memoised, so the comparison is by content, not by sequence: every distinct event of the reference -- each rough-mode line
    (mode, SATD, bits, cost), each candidate cost of the first RD loop, each TU's three blocks through transform / RDOQ and through
    dequantiser / inverse transform -- has to occur, value for value, among the kernel's."""
    import subprocess
    import sys
    import hevcdl_amd
    lib = hevcdl_amd.TRACE_LIB_PATH
    hevcdl_amd.build_ext(defines=("HEVCDL_STAGE_TRACE",), out=lib)                    # no-op when built by __graft_entry__.build()
    out = str(tmp_path / "log.npz")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _STAGE_CHILD % (root, os.path.join(root, "oracle")), os.path.join(GOLD, name + ".npz"), out],
                       env=dict(os.environ, HEVCDL_LIB=lib), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    f, g = np.load(os.path.join(GOLD, name + ".npz")), np.load(out)
    assert int(g["used"]) <= g["words"].size, "log overflow"
    ref_lines, ref_tus = _stage_sets_of_fixture(f)
    dev_lines, dev_tus = _stage_sets_of_log(g["words"])
    missing_lines, missing_tus = ref_lines - dev_lines, ref_tus - dev_tus
    assert not missing_lines, "%d of %d cost lines missing, e.g. %s" % (len(missing_lines), len(ref_lines), sorted(missing_lines)[:3])
    assert not missing_tus, "%d of %d TU events missing, e.g. %s" % (len(missing_tus), len(ref_tus), [t[:3] for t in sorted(missing_tus)[:5]])
    assert len(dev_lines - ref_lines) <= len(ref_lines) // 20        # the mode search of a CU coded while an earlier CU's second pass is pending is thrown away when that pass chooses the split
    assert len(dev_tus - ref_tus) <= len(ref_tus) // 20              # speculative codings that were thrown away (measured: 0 and 62 of 5443)
    assert len(ref_lines) > 1000 and len(ref_tus) > 1000


@pytest.mark.parametrize("w,h,qp,nf,seed", [(256, 128, 30, 3, 41), (136, 72, 24, 2, 42), (320, 192, 40, 1, 43)])
def test_matches_oracle_on_seeded_inputs(oracle_built, w, h, qp, nf, seed):
    import hevcdl_amd
    import ref_tools
    yuv = ref_tools.synth_yuv(w, h, nf, seed)
    labels = ref_tools.make_labels(w, h, nf, "rand", seed + 1)
    enc = hevcdl_amd.Encoder(w, h, qp, max_frames=nf)
    recs, recon, stats = enc.compress_frames(yuv, labels)
    enc.close()
    o_recs, o_recon, o_stats = ref_tools.run_oracle(yuv, w, h, qp, labels)
    assert_records_equal(recs, o_recs, "oracle %dx%d" % (w, h))
    assert np.array_equal(recon, o_recon)
    assert np.array_equal(stats["sse"], o_stats["sse"]) and np.array_equal(stats["est_bits"], o_stats["est_bits"])


@pytest.mark.parametrize("w,h,qp,nf,seed,tiles", [(768, 192, 32, 2, 44, (3, 2)), (520, 200, 25, 1, 45, (2, 4)), (1280, 64, 36, 1, 46, (5, 1)),
                                                  (1280, 200, 29, 1, 47, ([4, 6, 10], [1, 3]))])        # last: TileUniformSpacing 0, explicit CTU sizes
def test_tiles_match_oracle(oracle_built, w, h, qp, nf, seed, tiles):
    """Tiles (uniform spacing): one wave per (frame, tile); records, reconstruction and the per-frame sums equal the oracle's tile run."""
    import hevcdl_amd
    import ref_tools
    yuv = ref_tools.synth_yuv(w, h, nf, seed)
    labels = ref_tools.make_labels(w, h, nf, "rand", seed + 1)
    enc = hevcdl_amd.Encoder(w, h, qp, max_frames=nf, tiles=tiles)
    recs, recon, stats = enc.compress_frames(yuv, labels)
    with pytest.raises(hevcdl_amd.HevcdlError):       # the per-CTU session is the untiled raster walk
        enc.begin_frames(yuv, labels)
        enc.compress_ctu(0, 0)
    enc.close()
    o_recs, o_recon, o_stats = ref_tools.run_oracle(yuv, w, h, qp, labels, tiles=tiles)
    assert_records_equal(recs, o_recs, "oracle tiles %dx%d" % (w, h))
    assert np.array_equal(recon, o_recon)
    dbk = hevcdl_amd.Encoder(w, h, qp, max_frames=nf, tiles=tiles)
    d = dbk.deblock_frames(recon, recs); sao, fin = dbk.sao_frames(yuv, d); dbk.close()
    o_sao, o_fin = ref_tools.run_sao(yuv, d, w, h, qp, tiles=tiles)
    assert sao.tobytes() == o_sao.tobytes() and np.array_equal(fin, o_fin)      # SAO merge candidates follow the same tile layout
    assert np.array_equal(stats["sse"], o_stats["sse"]) and np.array_equal(stats["est_bits"], o_stats["est_bits"])
    u_recs, _, _ = ref_tools.run_oracle(yuv, w, h, qp, labels)
    assert any(not np.array_equal(o_recs[k], u_recs[k]) for k in FIELDS)      # the tiling does change the decisions


@pytest.mark.parametrize("w,h,qp,nf,seed,tiles,kind", [(256, 192, 24, 2, 81, (1, 1), "pattern"), (520, 136, 35, 1, 82, (2, 2), "pattern"), (192, 128, 30, 1, 83, (1, 1), "noise"),
                                                      (128, 64, 12, 1, 84, (1, 1), "noise"), (128, 128, 48, 1, 85, (1, 1), "pattern")])
def test_ten_bit_matches_oracle(oracle_built, w, h, qp, nf, seed, tiles, kind):
    """InternalBitDepth 10 (uint16 samples; C5's sample format: the 8-bit pattern * 4 + noise): records, reconstruction, sums bit-exact
    against the oracle; the CNN stage runs on the top 8 bits."""
    import hevcdl_amd
    import ref_tools
    rng = np.random.default_rng(seed)
    yuv = ref_tools.synth_yuv(w, h, nf, seed).astype(np.uint16) * 4 + rng.integers(0, 4, (nf, w * h * 3 // 2)).astype(np.uint16)
    if kind == "noise":
        yuv = rng.integers(0, 1024, yuv.shape).astype(np.uint16)
    labels = ref_tools.make_labels(w, h, nf, "rand", seed + 1)
    enc = hevcdl_amd.Encoder(w, h, qp, max_frames=nf, tiles=tiles, bit_depth=10)
    recs, recon, stats = enc.compress_frames(yuv, labels)
    cnn_labels = enc.predict_depth(yuv)
    enc.close()
    o_recs, o_recon, o_stats = ref_tools.run_oracle(yuv, w, h, qp, labels, tiles=tiles, bit_depth=10)
    assert_records_equal(recs, o_recs, "oracle 10-bit %dx%d" % (w, h))
    assert recon.dtype == np.uint16 and np.array_equal(recon, o_recon.reshape(recon.shape)) and int(recon.max()) <= 1023
    assert np.array_equal(stats["sse"], o_stats["sse"]) and np.array_equal(stats["est_bits"], o_stats["est_bits"])
    enc8 = hevcdl_amd.Encoder(w, h, qp, max_frames=nf)
    assert np.array_equal(cnn_labels, enc8.predict_depth((yuv >> 2).astype(np.uint8)))
    enc8.close()


def test_tile_ranges_assemble_to_the_whole_picture(oracle_built):
    """hevcdl_compress_tiles_dev: the tiles of a picture decided by separate launches (as separate GPUs would) into the same
    whole-frame buffers give exactly the single-launch result -- tiles never read each other."""
    import torch
    import hevcdl_amd
    import ref_tools
    w, h, qp, nf, tiles = 1024, 192, 30, 2, (4, 3)
    yuv = ref_tools.synth_yuv(w, h, nf, 47)
    labels = ref_tools.make_labels(w, h, nf, "rand", 48)
    enc = hevcdl_amd.Encoder(w, h, qp, max_frames=nf, tiles=tiles)
    recs, recon, stats = enc.compress_frames(yuv, labels)
    dev = torch.device("cuda:0")
    d_yuv = torch.from_numpy(yuv.reshape(nf, -1)).to(dev)
    d_lab = torch.from_numpy(labels).to(dev)
    d_recs = torch.zeros((nf, recs.shape[1] * recs.dtype.itemsize), dtype=torch.uint8, device=dev)
    d_recon = torch.zeros_like(d_yuv)
    sums = np.zeros(nf, hevcdl_amd.STATS_DTYPE)
    for begin, count in ((5, 7), (0, 2), (2, 3)):              # any order: no launch depends on another
        d_stats = torch.zeros((nf, hevcdl_amd.STATS_DTYPE.itemsize), dtype=torch.uint8, device=dev)
        enc.compress_tiles_dev(d_yuv.data_ptr(), nf, d_lab.data_ptr(), d_recs.data_ptr(), d_recon.data_ptr(), d_stats.data_ptr(), begin, count)
        torch.cuda.synchronize()
        part = np.frombuffer(d_stats.cpu().numpy().tobytes(), hevcdl_amd.STATS_DTYPE)
        sums["sse"] += part["sse"]; sums["est_bits"] += part["est_bits"]
    with pytest.raises(hevcdl_amd.HevcdlError):
        enc.compress_tiles_dev(d_yuv.data_ptr(), nf, d_lab.data_ptr(), d_recs.data_ptr(), d_recon.data_ptr(), None, 10, 3)
    enc.close()
    assert d_recs.cpu().numpy().tobytes() == recs.tobytes()
    assert np.array_equal(d_recon.cpu().numpy().reshape(recon.shape), recon)
    assert np.array_equal(sums["sse"], stats["sse"]) and np.array_equal(sums["est_bits"], stats["est_bits"])


@pytest.mark.parametrize("kind,qp,seed", [("noise", 22, 61), ("noise", 32, 62), ("texture", 27, 63), ("texture", 37, 64), ("edges", 22, 65), ("edges", 32, 66),
                                         ("flat", 37, 67), ("mix", 17, 68), ("mix", 45, 69)])
def test_matches_oracle_on_hard_content(oracle_built, kind, qp, seed):
    """Content that drives the rarely taken branches (escape codes, Rice adaptation, sign hiding, transform skip, large
    last positions, all-zero blocks): every field bit-exact against the oracle at every label depth."""
    import hevcdl_amd
    import ref_tools
    w, h, nf = 256, 192, 1
    rng = np.random.default_rng(seed)
    base = ref_tools.synth_yuv(w, h, nf, seed).astype(np.int64)
    if kind == "noise":
        yuv = rng.integers(0, 256, base.shape)
    elif kind == "texture":
        yuv = base + rng.normal(0, 25, base.shape).astype(np.int64)
    elif kind == "edges":
        yuv = (base // 64) * 64 + ((np.arange(base.shape[1]) // 3) % 2) * 90
    elif kind == "flat":
        yuv = np.full(base.shape, 97) + (rng.integers(0, 2, base.shape))
    else:
        yuv = np.where(rng.integers(0, 2, base.shape) > 0, base, rng.integers(0, 256, base.shape))
    yuv = np.clip(yuv, 0, 255).astype(np.uint8)
    labels = ref_tools.make_labels(w, h, nf, "rand", seed + 1)
    enc = hevcdl_amd.Encoder(w, h, qp, max_frames=nf)
    recs, recon, stats = enc.compress_frames(yuv, labels)
    enc.close()
    o_recs, o_recon, o_stats = ref_tools.run_oracle(yuv, w, h, qp, labels)
    assert_records_equal(recs, o_recs, "%s qp%d" % (kind, qp))
    assert np.array_equal(recon, o_recon)
    assert np.array_equal(stats["est_bits"], o_stats["est_bits"])


@pytest.mark.parametrize("wavefront", [False, True], ids=["default-cfg", "wavefront"])
def test_per_ctu_session_equals_batch_and_rejects_disorder(wavefront):
    """hevcdl_compress_ctu (compressCtu + encodeCtu of one CTU) called in coding order reproduces the batched path
    bit for bit; the coder state it returns can be fed back; out-of-order submission is an error, not a hang.
    With WaveFrontSynchro 1 as well (3 x 3 CTUs there): a row's first CTU starts from the contexts the session kept behind the second CTU of the row above, whatever
    state the caller hands in, and the batched launch it is compared with walks the rows on waves of their own."""
    import hevcdl_amd
    import ref_tools
    w, h, qp, nf = (192, 192, 32, 2) if wavefront else (192, 128, 32, 2)
    yuv = ref_tools.synth_yuv(w, h, nf, seed=21)
    labels = ref_tools.make_labels(w, h, nf, "rand", 22)
    e = hevcdl_amd.Encoder(w, h, qp, max_frames=nf, wavefront=wavefront)
    recs, recon, stats = e.compress_frames(yuv, labels)
    used = e.begin_frames(yuv, labels)
    assert np.array_equal(used.reshape(labels.shape), labels)
    with pytest.raises(hevcdl_amd.HevcdlError):
        e.compress_ctu(0, 1)                                  # CTU 0 not coded yet
    with pytest.raises(hevcdl_amd.HevcdlError):
        e.compress_ctu(nf, 0)                                 # frame outside the session
    for f in range(nf):
        state = None
        for a in range(e.ctus):
            # frame 0 continues from the context's own state, frame 1 feeds the returned state back explicitly
            rec, state = e.compress_ctu(f, a, state_in=(state if (f == 1 and a > 0) else None))
            for name in hevcdl_amd.REC_DTYPE.names:
                assert np.array_equal(rec[0][name], recs[f, a][name]), (f, a, name)
        assert np.array_equal(e.get_recon(f), recon[f].reshape(-1))
        assert int(state["frac"][0]) >> 15 >= 0
    # re-coding an earlier CTU is allowed only with its entry state
    with pytest.raises(hevcdl_amd.HevcdlError):
        e.compress_ctu(0, 2)
    e.close()


def walk_of_labels(labels):
    """[..., 16] labels -> the depth the reference's walk decides for every 16x16 cell: it reads ONE label per CU, the one of its top-left cell
    (TEncCu.cpp:496-520) -- a first label of 0 is one 64x64 CU whatever the other cells say, a quadrant whose first label is 1 one 32x32 CU; below
    that every cell speaks for itself."""
    walk = np.array(labels, np.uint8).reshape(-1, 16).copy()
    whole = walk[:, 0] == 0
    walk[whole] = 0
    for q in ((0, 1, 4, 5), (2, 3, 6, 7), (8, 9, 12, 13), (10, 11, 14, 15)):
        one = (~whole) & (walk[:, q[0]] == 1)
        for c in q:
            walk[one, c] = 1
    return walk.reshape(np.shape(labels))


@pytest.mark.parametrize("w,h", [(1920, 1080), (3840, 2160)])
def test_full_size_properties(w, h):
    """BASELINE.json sizes, where the oracle would take minutes: size-independent properties of the path.
    frame independence (batch == frame by frame), determinism, decided CU depth == what the reference's walk makes of the CNN labels (a CU is
    evaluated only at its labelled depth), NxN only in 8x8 CUs, cbf <-> coefficients, statistics == recomputed SSE."""
    import hevcdl_amd
    import ref_tools
    qp, nf = 32, 2
    yuv = ref_tools.synth_yuv(w, h, nf, seed=1000)
    e = hevcdl_amd.Encoder(w, h, qp, max_frames=nf)
    labels = e.predict_depth(yuv)
    recs, recon, stats = e.compress_frames(yuv, labels)
    recs1, recon1, stats1 = e.compress_frames(yuv[1:2], labels[1:2])
    assert recs1.tobytes() == recs[1:2].tobytes() and np.array_equal(recon1, recon[1:2]) and stats1.tobytes() == stats[1:2].tobytes()
    # depth map == labels: partition z -> 16x16 block of the CTU (z >> 4 in z-order, mapped to raster 4x4 grid)
    z16 = np.arange(16); zx = (z16 & 1) | ((z16 >> 1) & 2); zy = ((z16 >> 1) & 1) | ((z16 >> 2) & 2)
    blk_of_z = (zy * 4 + zx)[np.arange(256) >> 4]
    cx, cy = (w + 63) // 64, (h + 63) // 64
    inside = np.zeros((cy * cx, 256), bool)
    z = np.arange(256); px = np.zeros(256, int); py = np.zeros(256, int)
    for b in range(4):
        px |= ((z >> (2 * b)) & 1) << b; py |= ((z >> (2 * b + 1)) & 1) << b
    for a in range(cy * cx):
        inside[a] = ((a % cx) * 64 + px * 4 < w) & ((a // cx) * 64 + py * 4 < h)
    walk = walk_of_labels(labels.reshape(nf, -1, 16))
    lab_z = walk[:, :, blk_of_z]
    assert np.array_equal(recs["depth"][:, inside], lab_z[:, inside])
    assert (recs["part_size"][:, inside][recs["depth"][:, inside] < 3] == 0).all()
    # a CTU without coded coefficients carries no cbf, and vice versa
    nz = (recs["coeff_y"] != 0).any(axis=2) | (recs["coeff_cb"] != 0).any(axis=2) | (recs["coeff_cr"] != 0).any(axis=2)
    assert np.array_equal(nz, (recs["cbf"] != 0).any(axis=(2, 3)))
    ysz = w * h
    for f in range(nf):
        d = yuv[f].astype(np.int64) - recon[f].reshape(-1).astype(np.int64)
        sse = [int((d[:ysz] ** 2).sum()), int((d[ysz:ysz + ysz // 4] ** 2).sum()), int((d[ysz + ysz // 4:] ** 2).sum())]
        assert [int(v) for v in stats["sse"][f]] == sse
        assert int(stats["ctus"][f]) == cx * cy and int(stats["est_bits"][f]) > 0
    e.close()


def test_c5_eight_k_ten_bit_tiles():
    """C5 of the survey: 7680x4320, 10-bit samples (8-bit pattern * 4 + noise, uint16), tiles 4 x 2 (one wave per tile; one tile per GPU in
    the 8-GPU layout).  Size-independent properties: every CU sits at its labelled depth; tile independence -- the top-left tile is bit
    for bit the picture obtained by coding its 1920x2176 crop on its own; the filtered picture keeps to 10 bits; the access unit carries
    seven entry points whose sub-streams fill the slice data."""
    import hevcdl_amd
    import ref_tools
    import hevc_parse as hp
    w, h, qp, tiles = 7680, 4320, 32, (4, 2)
    rng = np.random.default_rng(5)
    yuv = ref_tools.synth_yuv(w, h, 1, seed=5000).astype(np.uint16) * 4 + rng.integers(0, 4, (1, w * h * 3 // 2)).astype(np.uint16)
    e = hevcdl_amd.Encoder(w, h, qp, max_frames=1, tiles=tiles, bit_depth=10)
    labels = e.predict_depth(yuv)
    recs, recon, stats = e.compress_frames(yuv, labels)
    dbk = e.deblock_frames(recon, recs)
    sao, final = e.sao_frames(yuv, dbk)
    e.close()
    cx, cy = 120, 68
    assert recs.shape == (1, cx * cy) and int(stats["ctus"][0]) == 8160 and int(recon.max()) <= 1023 and int(final.max()) <= 1023
    # depth == label inside the picture (the last CTU row is half outside: 4320 = 67.5 * 64)
    z16 = np.arange(16); zx = (z16 & 1) | ((z16 >> 1) & 2); zy = ((z16 >> 1) & 1) | ((z16 >> 2) & 2)
    blk_of_z = (zy * 4 + zx)[np.arange(256) >> 4]
    full_rows = recs["depth"][0].reshape(cy, cx, 256)[:cy - 1]
    assert np.array_equal(full_rows, walk_of_labels(labels.reshape(cy, cx, 16))[:cy - 1][:, :, blk_of_z])
    # tile 0 = CTU columns 0..29, rows 0..33 == its crop coded as a picture of its own
    tw, th = 30 * 64, 34 * 64
    ysz = w * h
    Y = yuv[0, :ysz].reshape(h, w)[:th, :tw]; U = yuv[0, ysz:ysz + ysz // 4].reshape(h // 2, w // 2)[:th // 2, :tw // 2]; V = yuv[0, ysz + ysz // 4:].reshape(h // 2, w // 2)[:th // 2, :tw // 2]
    crop = np.concatenate([Y.ravel(), U.ravel(), V.ravel()])[None]
    lab0 = labels.reshape(cy, cx, 16)[:34, :30].reshape(1, -1, 16)
    e0 = hevcdl_amd.Encoder(tw, th, qp, max_frames=1, bit_depth=10)
    recs0, recon0, _ = e0.compress_frames(crop, lab0)
    e0.close()
    assert recs.reshape(cy, cx)[:34, :30].tobytes() == recs0.reshape(34, 30).tobytes()
    assert np.array_equal(recon[0, :ysz].reshape(h, w)[:th, :tw].ravel(), recon0[0, :tw * th])
    # the access unit: main10, 8 tiles, 7 entry points
    au = hevcdl_amd.write_access_unit(w, h, qp, 0, recs[0], sao=sao[0], tiles=tiles, bit_depth=10)
    nals = hp.split_annexb(au)
    sps, pps = hp.parse_sps(nals[1][1]), hp.parse_pps(nals[2][1])
    assert (sps["width"], sps["height"], sps["bd_luma_m8"], sps["bd_chroma_m8"], sps["sao"]) == (7680, 4320, 2, 2, 1)
    assert (pps["tile_columns"], pps["tile_rows"]) == (4, 2)
    hdr, _ = hp.parse_slice_header(nals[3][1], sps, pps)
    assert len(hdr["entry_points"]) == 7 and sum(hdr["entry_points"]) < len(nals[3][1]) - hdr["data_byte_pos"]


@pytest.mark.parametrize("w,h,qp,kind,bd", [(8, 8, 32, "pattern", 8), (64, 8, 20, "noise", 8), (8, 72, 40, "noise", 8), (136, 72, 0, "noise", 8), (136, 72, 51, "pattern", 8),
                                            (72, 136, 0, "noise", 10), (200, 72, 51, "noise", 10), (64, 64, 30, "black", 8), (64, 64, 30, "white", 10)])
def test_extreme_sizes_and_qps_match_oracle(oracle_built, w, h, qp, kind, bd):
    """Smallest pictures (a single 8x8 CU inside one CTU), one-CU-wide strips, QP 0 (largest levels: escape codes, 16-bit level clipping)
    and QP 51, constant pictures at both ends of the sample range: records, reconstruction, filters bit-exact against the oracle."""
    import hevcdl_amd
    import ref_tools
    rng = np.random.default_rng(w * 1000 + h + qp)
    mx = (1 << bd) - 1
    if kind == "noise":
        yuv = rng.integers(0, mx + 1, (1, w * h * 3 // 2))
    elif kind == "black":
        yuv = np.zeros((1, w * h * 3 // 2), np.int64)
    elif kind == "white":
        yuv = np.full((1, w * h * 3 // 2), mx, np.int64)
    else:
        yuv = ref_tools.synth_yuv(w, h, 1, w + h).astype(np.int64) * (4 if bd == 10 else 1)
    yuv = yuv.astype(np.uint8 if bd == 8 else np.uint16)
    labels = ref_tools.make_labels(w, h, 1, "rand", qp + 7)
    enc = hevcdl_amd.Encoder(w, h, qp, max_frames=1, bit_depth=bd)
    recs, recon, stats = enc.compress_frames(yuv, labels)
    dbk = enc.deblock_frames(recon, recs)
    sao, final = enc.sao_frames(yuv, dbk)
    enc.close()
    o_recs, o_recon, o_stats = ref_tools.run_oracle(yuv, w, h, qp, labels, bit_depth=bd)
    assert_records_equal(recs, o_recs, "oracle %dx%d qp %d" % (w, h, qp))
    assert np.array_equal(recon, o_recon.reshape(recon.shape))
    assert np.array_equal(stats["sse"], o_stats["sse"]) and np.array_equal(stats["est_bits"], o_stats["est_bits"])
    o_dbk = ref_tools.run_deblock(o_recon.reshape(1, -1), w, h, qp, np.frombuffer(o_recs.tobytes(), dtype=ref_tools.REC_DTYPE).reshape(1, -1), bit_depth=bd)
    assert np.array_equal(dbk, o_dbk.reshape(dbk.shape))
    o_sao, o_final = ref_tools.run_sao(yuv, o_dbk, w, h, qp, bit_depth=bd)
    assert sao.tobytes() == o_sao.tobytes() and np.array_equal(final, o_final.reshape(final.shape))
    au = hevcdl_amd.write_access_unit(w, h, qp, 0, recs[0], sao=sao[0], bit_depth=bd)
    assert len(au) > 60


@pytest.mark.parametrize("seed", list(range(24)))
def test_random_configurations_match_oracle(oracle_built, seed):
    """Seeded sweep over picture size (multiples of 8, ragged CTU edges), QP 0..51, bit depth, content, label policy and -- when the picture is
    wide enough -- tile layout (uniform or explicit) with either LFCrossTileBoundaryFlag: decisions, reconstruction, deblocking and SAO
    against the oracle."""
    import hevcdl_amd
    import ref_tools
    from test_oracle_vs_reference_sweep import sweep_case      # the same cases are run against the reference encoder in the container
    w, h, qp, bd, tiles, lf, yuv, labels = sweep_case(seed)
    enc = hevcdl_amd.Encoder(w, h, qp, max_frames=1, tiles=tiles, bit_depth=bd, lf_across_tiles=lf)
    recs, recon, stats = enc.compress_frames(yuv, labels)
    dbk = enc.deblock_frames(recon, recs)
    sao, final = enc.sao_frames(yuv, dbk)
    enc.close()
    what = "%dx%d qp %d bd %d tiles %s lf %d" % (w, h, qp, bd, tiles, lf)
    o_recs, o_recon, o_stats = ref_tools.run_oracle(yuv, w, h, qp, labels, tiles=tiles, bit_depth=bd)
    assert_records_equal(recs, o_recs, what)
    assert np.array_equal(recon, o_recon.reshape(recon.shape)), what
    assert np.array_equal(stats["sse"], o_stats["sse"]) and np.array_equal(stats["est_bits"], o_stats["est_bits"]), what
    o_dbk = ref_tools.run_deblock(o_recon.reshape(1, -1), w, h, qp, np.frombuffer(o_recs.tobytes(), dtype=ref_tools.REC_DTYPE).reshape(1, -1), bit_depth=bd, tiles=tiles, lf_across_tiles=lf)
    assert np.array_equal(dbk, o_dbk.reshape(dbk.shape)), what
    o_sao, o_final = ref_tools.run_sao(yuv, o_dbk, w, h, qp, tiles=tiles, bit_depth=bd, lf_across_tiles=lf)
    assert sao.tobytes() == o_sao.tobytes() and np.array_equal(final, o_final.reshape(final.shape)), what
    au = hevcdl_amd.write_access_unit(w, h, qp, 0, recs[0], sao=sao[0], tiles=tiles, bit_depth=bd, lf_across_tiles=lf)
    assert len(au) > 60
