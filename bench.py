#!/usr/bin/env python3
"""bench.py -- all-intra CTUs/s of the hot path (on-device CNN depth predictor + depth-pruned CTU decision kernel).

Contract: `python bench.py --gpus N --steps K --warmup W`; for N>1 launched by torch.distributed.run, one rank per GPU.

Workload of the headline line = C4 of BASELINE.json as written: 600 frames of 3840x2160, QP32.  The 600 frames are the job: with N ranks
every rank takes a contiguous block of 600/N frames (sharding.shard_frames), i.e. STRONG scaling -- the total work is fixed.  A step = one
pass of the whole hot path (hevcdl_encode_frames_dev: CNN + RD search) over the rank's frames, which are already resident in HBM.  Frames
are independent, so there is no data-path collective; RCCL only gathers the per-frame rate records.
Prints ONE JSON line on rank 0 (metric of BASELINE.json + roofline + cpu_baseline); on the single-GPU run it additionally carries
  "saturated"    the same step over 2048 frames per GPU (every wave of the chip owns a frame: the throughput ceiling of the kernel),
  "c2"           C2 of BASELINE.json (10 frames of 1920x1080) next to the reference encoder on 10 host cores in the same run,
  "parity_check" the reference encoder's own CTU records / reconstruction against the GPU's: whole frames of the timed job (every CTU row, the ragged
                 bottom row of 2160p included) and the top band of 256 more,
  "e2e"          the same 600 frames through the WHOLE picture pipeline (CNN, decisions, deblocking, SAO on the device; final entropy coding on host
                 threads) -- the stage list of the CPU baseline, for a like-for-like ratio,
  "latency_floor_s" one frame alone on the GPU: a frame is a serial chain of CTUs, so no sharding of the job finishes sooner than this.
"""
import argparse
import hashlib
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
ALGO_BYTES_PER_CTU = 27408        # orig 6144 + recon 6144 + levels 12288 + record 2816 + labels 16
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_F16_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense f16 / bf16 MFMA peak (no sparsity)
# The convolutions of the label CNN per CTU (use_model.py:16-58; conv64 once per CTU, the other three per 32x32 quadrant), in multiply-accumulates:
#   conv64 64x64 positions x 16 channels x 75 taps, conv1 4 x 32x32 x 16 x 75, conv2 4 x 16x16 x 64 x 288, conv3 4 x 8x8 x 128 x 576
CNN_CONV_MACS = 64 * 64 * 16 * 75 + 4 * 32 * 32 * 16 * 75 + 4 * 16 * 16 * 64 * 288 + 4 * 8 * 8 * 128 * 576
# what the kernel executes for them (csrc/cnn_kernel.hip): conv2 / conv3 every product as three f16 MFMA products of split operands (hi*hi + hi*lo + lo*hi, f32
# accumulate); the 5x5 layers on raw split words: the 75 taps padded to five k-steps of 16, two MFMAs (K = 32 halves) a step = 320 MFMA products per output
CNN_CONV_MACS_EXECUTED = (64 * 64 * 16 + 4 * 32 * 32 * 16) * 320 + 3 * (4 * 16 * 16 * 64 * 288 + 4 * 8 * 8 * 128 * 576)
RD_KERNEL_SRC = os.path.join(ROOT, "hevc-deep-learning-pipeline_amd", "csrc", "rd_kernel.hip")


def synth_frames_torch(torch, dev, width, height, frame_ids, seed):
    """Synthetic planar 4:2:0 frames generated on the device (generator of SURVEY.md section 8d).  A frame depends only on its index
    (noise seeded per frame), so a frame's content does not depend on how the job is sharded."""
    n_frames = len(frame_ids)
    y = torch.arange(height, device=dev, dtype=torch.float32).view(height, 1)
    x = torch.arange(width, device=dev, dtype=torch.float32).view(1, width)
    out = torch.empty((n_frames, width * height * 3 // 2), dtype=torch.uint8, device=dev)
    base = 128 + 50 * torch.sin(x / 57) * torch.cos(y / 43)
    xc, yc = x[:, ::2], y[::2, :]
    U = (128 + 25 * torch.sin(xc / 61)).expand(height // 2, width // 2).clamp(0, 255).to(torch.uint8).reshape(-1)
    V = (128 + 25 * torch.cos(yc / 47)).expand(height // 2, width // 2).clamp(0, 255).to(torch.uint8).reshape(-1)
    g = torch.Generator(device=dev)
    bw, bh = min(1200, width // 3), min(600, height // 3)
    for i, k in enumerate(frame_ids):
        g.manual_seed(seed + int(k))
        Y = base + 30 * torch.sin((x + y + 3 * float(k)) / 19) + torch.randn((height, width), device=dev, generator=g) * 5
        bx = (width // 5 + 4 * int(k)) % max(1, width - bw)
        by = height // 4
        Y[by:by + bh, bx:bx + bw] = torch.floor(Y[by:by + bh, bx:bx + bw] / 24) * 24
        out[i, :width * height] = Y.clamp(0, 255).to(torch.uint8).reshape(-1)
        out[i, width * height:width * height * 5 // 4] = U
        out[i, width * height * 5 // 4:] = V
    return out


def _cpu_worker(args):
    yuv, w, h, qp, labels = args
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_tools
    t = time.time()
    ref_tools.run_oracle(yuv, w, h, qp, labels)
    return time.time() - t


REF_ENC = os.path.join(ROOT, "oracle", "_ref", "TAppEncoder_ref")
REF_STAGES = "CTU decisions (label files preloaded) + final entropy coding + deblocking + SAO + bitstream / reconstruction files, process start-up and file I/O included; label CNN excluded"
GPU_STAGES = "on-device label CNN + CTU decisions (records, levels, reconstruction before the in-loop filters), frames resident in HBM; final entropy coding, deblocking, SAO and file I/O excluded"


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _prepare_ref_run(args):
    """Working directory of one reference-encoder process: input picture, the label files it polls for (TEncCu.cpp:244-253), output dir."""
    idx, yuv, w, h, qp, labels, base = args
    d = os.path.join(base, "p%s" % idx)
    os.makedirs(os.path.join(d, "rec"))
    yuv.astype(np.uint8).tofile(os.path.join(d, "in.yuv"))
    os.makedirs(os.path.join(d, "pred", "0"))
    for a in range(labels.shape[1]):
        with open(os.path.join(d, "pred", "0", "ctu%d.txt" % a), "w") as fh:
            fh.write(" ".join(str(int(v)) for v in labels[0, a]) + " ")
    return d


def _run_ref(args):
    import subprocess
    d, cmd, dump = args
    env = dict(os.environ)
    if dump:
        env["HEVCDL_DUMP"] = os.path.join(d, "dump.bin")        # oracle/ref_hook.cpp: CTU record + reconstruction after every compressCtu
    t = time.time()
    r = subprocess.run(cmd, cwd=d, env=env, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("reference encoder failed: " + r.stdout[-500:] + r.stderr[-500:])
    return time.time() - t


def _band(fr, width, full_h, height):
    ysz, csz = width * full_h, (width // 2) * (full_h // 2)
    return np.concatenate([fr[:width * height], fr[ysz:ysz + (width // 2) * (height // 2)], fr[ysz + csz:ysz + csz + (width // 2) * (height // 2)]])


def run_reference_pictures(pictures, labels, width, height, qp, procs, dump=False, workers=None, wavefront=0):
    """P reference-encoder processes (oracle/_ref/TAppEncoder_ref; configuration = the reference's cfg as switches, oracle/ref_args.py), one
    picture each, started together; -> (wall seconds, per-process seconds, list of dump arrays or None)."""
    import shutil
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_args
    base = tempfile.mkdtemp(prefix="hevcdl_cpu_")
    dumps = None
    try:
        dirs = [_prepare_ref_run((i, pictures[i], width, height, qp, labels[i:i + 1], base)) for i in range(procs)]
        cmd = [REF_ENC, "-i", "in.yuv", "-b", "rec/str.bin", "-o", "rec/rec.yuv"] + ref_args.reference_args(width, height, 1, qp, wavefront=wavefront)
        t = time.time()
        with ThreadPoolExecutor(max_workers=workers or procs) as pool:
            per = list(pool.map(_run_ref, [(d, cmd, dump) for d in dirs]))
        wall = time.time() - t
        if dump:
            import ref_tools
            dumps = [np.fromfile(os.path.join(d, "dump.bin"), dtype=ref_tools.DUMP_DTYPE) for d in dirs]
    finally:
        shutil.rmtree(base, ignore_errors=True)
    return wall, per, dumps


def parity_against_dumps(dumps, gpu_records, gpu_recon_bands, width, height):
    """Reference CTU records + reconstruction (oracle/ref_hook.cpp dumps) against the GPU's for the same pictures: every field of the record
    and the three reconstruction blocks of every CTU, bit for bit."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_tools
    ctus = mism = 0
    first = None
    for i, dump in enumerate(dumps):
        for e in dump:
            a = int(e["addr"])
            r = gpu_records[i, a]
            ok = all(np.array_equal(e["rec"][k], r[k]) for k in ref_tools.FIELDS)
            if ok:
                ry, ru, rv = ref_tools.ctu_recon_from_frame(gpu_recon_bands[i], width, height, a)
                ok = np.array_equal(e["rec_y"], ry) and np.array_equal(e["rec_cb"], ru) and np.array_equal(e["rec_cr"], rv)
            ctus += 1
            if not ok:
                mism += 1
                first = first or [i, a]
    return {"ctus": ctus, "mismatches": mism, "first_mismatch": first,
            "checked": "every field of hevcdl_ctu_record (depth, partition, intra modes, TU tree, cbf, transform skip, bits / distortion / cost, levels) and the "
                       "reconstruction of every CTU, reference encoder (TEncCu::compressCtu through oracle/ref_hook.cpp) vs GPU"}


def effective_cores():
    """Host cores this process may use: the hardware threads, capped by the cgroup CPU quota (the GPU boxes run the job in a container
    whose cpu.max is far below the node's thread count; more processes than that only time-slice)."""
    n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline_reference(yuv_host, labels_host, width, height, qp, max_procs=None, band_rows=6, gpu_records=None, gpu_recon=None, gpu_records_full=None):
    """The reference itself on the host cores.  (1) P = usable cores: P processes, a whole frame of the workload each,
    wall clock from first start to last exit -> value.  (2) one such process alone -> one_process.  (3) parity: the top `band_rows` CTU
    rows of every sampled frame (the decisions of those rows do not depend on the rows below) are encoded with the CTU-record dump of
    oracle/ref_hook.cpp switched on and compared with the GPU's records / reconstruction of the same frames."""
    cores = effective_cores()
    p = min(cores, yuv_host.shape[0], max_procs or cores)
    nct = labels_host.shape[1]
    ww, perw, _ = run_reference_pictures([yuv_host[i] for i in range(p)], np.ascontiguousarray(labels_host[:p]), width, height, qp, p)
    out = {"value": p * nct / ww, "unit": "CTUs/s", "cores": p, "kind": "reference", "cpu_model": cpu_model(),
           "sample": "%d whole %dx%d frames of the workload, QP%d, one reference-encoder process per usable core (%d hardware threads on the node, cgroup CPU quota %d), "
                     "%.1f s wall, %.1f CTUs/s per process" % (p, width, height, qp, os.cpu_count() or 1, cores, ww, p * nct / sum(perw)),
           "stages_cpu": REF_STAGES, "stages_gpu": GPU_STAGES}
    w1, _, _ = run_reference_pictures([yuv_host[0]], np.ascontiguousarray(labels_host[:1]), width, height, qp, 1)
    out["one_process"] = {"value": nct / w1, "unit": "CTUs/s", "cores": 1, "sample": "one whole frame, one process alone on the node, %.1f s" % w1}
    parity = None
    whole = None
    if gpu_records_full is not None:      # the p whole frames once more, OUTSIDE the timed legs, with the CTU-record dump of oracle/ref_hook.cpp: every CTU of the picture
        t = time.time()
        _, _, dumps = run_reference_pictures([yuv_host[i] for i in range(p)], np.ascontiguousarray(labels_host[:p]), width, height, qp, p, dump=True)
        whole = parity_against_dumps(dumps, gpu_records_full[:p], [gpu_recon[i] for i in range(p)], width, height)
        whole["sample"] = "%d WHOLE %dx%d frames of the timed job (all %d CTU rows, the last one half outside the picture), %.1f s" % (p, width, height, (height + 63) // 64, time.time() - t)
    if gpu_records is not None:
        n = gpu_records.shape[0]
        bh = min(height, 64 * band_rows)
        nb = ((width + 63) // 64) * ((bh + 63) // 64)
        bands = [_band(yuv_host[i], width, height, bh) for i in range(n)]
        t = time.time()
        _, _, dumps = run_reference_pictures(bands, np.ascontiguousarray(labels_host[:n, :nb]), width, bh, qp, n, dump=True, workers=cores)
        parity = parity_against_dumps(dumps, gpu_records, [_band(gpu_recon[i], width, height, bh) for i in range(n)], width, bh)
        parity["sample"] = "top %dx%d band (%d CTU rows) of %d frames of the timed job, %.1f s" % (width, bh, band_rows, n, time.time() - t)
    if whole is not None:                 # one record: whole frames first, the bands of more frames beside them
        bands = parity
        parity = dict(whole)
        if bands is not None:
            parity["ctus"] += bands["ctus"]; parity["mismatches"] += bands["mismatches"]
            parity["first_mismatch"] = parity["first_mismatch"] or bands["first_mismatch"]
            parity["sample"] = whole["sample"] + " + " + bands["sample"]
            parity["whole_frames"] = {"ctus": whole["ctus"], "mismatches": whole["mismatches"]}
            parity["bands"] = {"ctus": bands["ctus"], "mismatches": bands["mismatches"]}
    return out, parity


def cpu_baseline_port(yuv_host, labels_host, width, height, qp, max_procs=None, band_rows=3):
    """Oracle (CPU port, bit-identical to the reference on the golden vectors) timed on the host cores of this node: stands in where the
    reference build (oracle/_ref) is absent.  P processes, each encoding the top `band_rows` CTU rows of a distinct frame."""
    import __graft_entry__ as g
    g.build_oracle()
    cores = effective_cores()
    p = min(cores, yuv_host.shape[0], max_procs or cores)
    bh = min(height, 64 * band_rows)
    cx = (width + 63) // 64
    bands = np.stack([_band(yuv_host[i], width, height, bh) for i in range(p)])
    lab = np.ascontiguousarray(labels_host[:p, :cx * ((bh + 63) // 64)])
    jobs = [(bands[i:i + 1], width, bh, qp, lab[i:i + 1]) for i in range(p)]
    t = time.time()
    with mp.get_context("spawn").Pool(p) as pool:
        per = pool.map(_cpu_worker, jobs)
    wall = time.time() - t
    ctus = p * lab.shape[1]
    return {"value": ctus / wall, "unit": "CTUs/s", "cores": p, "kind": "port", "cpu_model": cpu_model(),
            "sample": "top %dx%d band of %d frames of the workload, QP%d (1 band per process, labels from the GPU CNN, CNN excluded), %.1f s wall, %.1f CTUs/s per core"
                      % (width, bh, p, qp, wall, ctus / sum(per) if sum(per) > 0 else 0.0)}, None


def measured_traffic(n_ctus):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes of THIS kernel source (profiles/r*_traffic.json name the sha of
    rd_kernel.hip they were collected on); None when the source has changed since (the counters cannot be read from inside the process)."""
    import glob
    sha = hashlib.sha256(open(RD_KERNEL_SRC, "rb").read()).hexdigest()[:16]
    tj = None
    for tpath in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):       # the newest counter pass taken on this very kernel source
        cand = json.load(open(tpath))
        if cand.get("rd_kernel_sha16") == sha and cand.get("fetch_bytes_per_ctu") and cand.get("write_bytes_per_ctu"):
            tj = cand
            break
    if tj is None:
        return None, "no committed counter pass (profiles/r*_traffic.json) was taken on this build of rd_kernel.hip (%s): not reported" % sha
    return (tj["fetch_bytes_per_ctu"] + tj["write_bytes_per_ctu"]) * n_ctus, tj["source"] + "; " + tj["note"]


def measured_issue(frames, ctus_per_s):
    """The instruction-issue bound of the decision kernel on a launch of `frames` frames (profiles/r*_issue.json, tools/issue_json.py: SQ counter passes of THIS
    kernel source, keyed by the sha of rd_kernel.hip like `traffic`) with `frac` = the decision kernel's measured CTUs/s of this run against the ceiling; None
    when no committed pass was taken on this source or on this launch shape."""
    import glob
    sha = hashlib.sha256(open(RD_KERNEL_SRC, "rb").read()).hexdigest()[:16]
    for ipath in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_issue.json")), reverse=True):
        cand = json.load(open(ipath))
        sh = cand.get("shapes", {}).get(str(frames))
        if cand.get("rd_kernel_sha16") == sha and sh:
            out = {k: sh[k] for k in ("valu_per_ctu", "salu_per_ctu", "lds_per_ctu", "lanes_enabled", "ceiling_ctus_per_s")}
            out.update({"frames_per_launch": frames, "unit": "CTUs/s", "achieved": ctus_per_s, "frac": ctus_per_s / sh["ceiling_ctus_per_s"], "frac_under_counters": sh.get("frac_under_counters"),
                        "model": cand["model"], "source": os.path.relpath(ipath, ROOT) + " <- " + cand["source"],
                        "note": "valu / salu / lds = wave-instructions per CTU (SQ_INSTS_*), lanes_enabled = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU (of 64); achieved = the decision "
                                "kernel alone (HIP events of this run), the ceiling is the VALU issue time of today's instruction count on every SIMD of the chip"})
            return out
    return {"frames_per_launch": frames, "frac": None, "note": "no committed counter pass (profiles/r*_issue.json) was taken on this build of rd_kernel.hip (%s) for a %d-frame launch: not reported" % (sha, frames)}


def timed_steps(torch, enc, tensors, n_frames, steps, warmup, barrier):
    yuv, labels, records, recon, stats = tensors
    stream = torch.cuda.current_stream()

    def step():
        enc.encode_frames_dev(yuv.data_ptr(), n_frames, labels.data_ptr(), records.data_ptr(), recon.data_ptr(), stats.data_ptr(), stream.cuda_stream)
    for _ in range(warmup):
        step()
    barrier()
    enc.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    prof = enc.profile_get()
    enc.profile_enable(False)
    return elapsed, prof


E2E_STAGES = ("on-device label CNN + CTU decisions + deblocking + SAO, frames resident in HBM; CTU records, SAO parameters and final pictures copied to page-locked host "
              "memory; final entropy coding of every access unit (hevcdl_write_access_unit: parameter sets + slice) on host threads, chunk by chunk while the next "
              "chunk is copied; streams and pictures stay in memory (no file I/O, no process start-up)")


def e2e_leg(torch, hevcdl_amd, dev, enc, tensors, n_frames, width, height, qp, chunk=60, wavefront=False):
    """The job once more through the WHOLE picture pipeline -- the stages the CPU baseline's number contains (TEncGOP.cpp:1742-1935: decisions, in-loop filters,
    entropy coding), the label CNN on top: one like-for-like throughput.  The device side is one batch (a smaller batch is no faster: a frame is a serial
    chain); the host codes chunk k on its threads while chunk k + 1 is copied."""
    import ctypes
    from concurrent.futures import ThreadPoolExecutor
    yuv, labels, records, recon, stats = tensors
    lib, ctus = enc.lib, enc.ctus
    rec_b, sao_b = hevcdl_amd.REC_DTYPE.itemsize * ctus, hevcdl_amd.SAO_DTYPE.itemsize * 3 * ctus
    final = torch.empty_like(recon)
    sao = torch.zeros((max(1, n_frames), sao_b), dtype=torch.uint8, device=dev)
    chunk = max(1, min(chunk, n_frames))
    hbuf = [(torch.empty((chunk, rec_b), dtype=torch.uint8).pin_memory(), torch.empty((chunk, sao_b), dtype=torch.uint8).pin_memory(),
             torch.empty((chunk, recon.shape[1]), dtype=torch.uint8).pin_memory()) for _ in range(2)]
    threads = effective_cores()
    cfg = hevcdl_amd.StreamConfig()
    if lib.hevcdl_stream_config_default(ctypes.byref(cfg), width, height, qp) != 0:
        raise RuntimeError("stream config")
    cfg.sao_enabled = 1
    cfg.wavefront = 1 if wavefront else 0          # (records of a context with WaveFrontSynchro 1: a sub-stream per CTU row)
    cap = lib.hevcdl_access_unit_bound(width, height)
    outs = [np.empty(cap, np.uint8) for _ in range(threads)]
    free = list(range(threads))

    def code(args):
        poc, recs_np, sao_np = args
        t = free.pop()
        n = ctypes.c_size_t(0)
        st = lib.hevcdl_write_access_unit(ctypes.byref(cfg), int(poc), recs_np.ctypes.data, sao_np.ctypes.data, outs[t].ctypes.data, cap, ctypes.byref(n))
        free.append(t)
        if st != 0:
            raise RuntimeError("write_access_unit %d" % st)
        return n.value

    copy_stream = torch.cuda.Stream(device=dev)
    main = torch.cuda.current_stream()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    enc.encode_frames_dev(yuv.data_ptr(), n_frames, labels.data_ptr(), records.data_ptr(), recon.data_ptr(), stats.data_ptr(), main.cuda_stream)
    enc.deblock_frames_dev(recon.data_ptr(), n_frames, records.data_ptr(), recon.data_ptr(), main.cuda_stream)
    enc.sao_frames_dev(yuv.data_ptr(), recon.data_ptr(), n_frames, sao.data_ptr(), final.data_ptr(), main.cuda_stream)
    torch.cuda.synchronize(dev)
    t_dev = time.perf_counter() - t0
    recs2d = records.view(max(1, records.shape[0]), -1)

    def fetch(ci):
        b0 = ci * chunk
        nb = min(chunk, n_frames - b0)
        h = hbuf[ci & 1]
        with torch.cuda.stream(copy_stream):
            h[0][:nb].copy_(recs2d[b0:b0 + nb], non_blocking=True)
            h[1][:nb].copy_(sao[b0:b0 + nb], non_blocking=True)
            h[2][:nb].copy_(final[b0:b0 + nb], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(copy_stream)
        return ev, b0, nb

    total_bytes = 0
    n_chunks = (n_frames + chunk - 1) // chunk
    with ThreadPoolExecutor(max_workers=threads) as pool:
        nxt = fetch(0)
        for ci in range(n_chunks):
            ev, b0, nb = nxt
            ev.synchronize()
            h = hbuf[ci & 1]
            r_np, s_np = h[0].numpy(), h[1].numpy()
            fut = pool.map(code, [(b0 + i, r_np[i], s_np[i]) for i in range(nb)])
            if ci + 1 < n_chunks:
                nxt = fetch(ci + 1)              # into the other buffer: the host codes this chunk meanwhile
            total_bytes += sum(fut)
    dt = time.perf_counter() - t0
    del final, sao, hbuf
    return {"value": n_frames * ctus / dt, "unit": "CTUs/s", "pictures_per_s": n_frames / dt, "seconds": dt, "device_seconds": t_dev, "host_threads": threads,
            "frames": n_frames, "stream_bytes": int(total_bytes), "stages": E2E_STAGES}


def wavefront_leg(torch, hevcdl_amd, dev, local, tensors, n_frames, width, height, qp, ctus, barrier, check):
    """EXTRA keys, not the headline: the same frames with WaveFrontSynchro 1 (cfg key of the reference, TAppEncCfg.cpp:975).  The key changes the stream (a sub-stream per CTU
    row, rows start from the contexts behind the second CTU of the row above) and with it the decisions -- parity is against the reference run WITH the key -- and it is the
    one key of the kept cfg surface that breaks a frame's serial chain of CTUs: rows run two CTUs apart (34 + 2 x 33 CTU steps at 2160p instead of 2040), each row a unit of
    the launch.  Reported: the 600-frame step, a GPU's share of the 8-GPU job (75 frames), one frame alone, C2; the records of the timed step overwrite the job's (this leg
    runs last)."""
    yuv, labels, records, recon, stats = tensors
    enc = hevcdl_amd.Encoder(width, height, qp, max_frames=max(1, n_frames), device=local, wavefront=True)
    stream = torch.cuda.current_stream().cuda_stream
    out = {"cfg": "WaveFrontSynchro 1 (all other keys as the headline)", "unit": "CTUs/s"}

    def timed(n, reps):
        ts = []
        for rep in range(reps + 1):                # the first launch of a shape is a warm-up
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            enc.encode_frames_dev(yuv.data_ptr(), n, labels.data_ptr(), records.data_ptr(), recon.data_ptr(), stats.data_ptr(), stream)
            torch.cuda.synchronize(dev)
            if rep:
                ts.append(time.perf_counter() - t1)
        return sorted(ts)[len(ts) // 2], enc.last_rd_launch()
    for key, n, reps in (("latency_floor_s", 1, 3), ("share_8gpu_s", min(n_frames, 75), 3), ("share_4gpu_s", min(n_frames, 150), 2), ("share_2gpu_s", min(n_frames, 300), 2), ("job_s", n_frames, 2)):
        t, launch = timed(n, reps)
        out[key] = t
        out[key.replace("_s", "_launch")] = launch
    out["value"] = n_frames * ctus / out["job_s"]
    out["scale_projection"] = {"seconds": {"1": out["job_s"], "2": out["share_2gpu_s"], "4": out["share_4gpu_s"], "8": out["share_8gpu_s"]},
                               "value": {k: n_frames * ctus / v for k, v in (("1", out["job_s"]), ("2", out["share_2gpu_s"]), ("4", out["share_4gpu_s"]), ("8", out["share_8gpu_s"]))},
                               "note": "as the headline's scale_projection: a rank's share of the %d frames on this GPU (label CNN + decisions), median of the repeats" % n_frames}
    if check:     # parity with the key set: whole frames of the timed step against the reference encoder run with --WaveFrontSynchro=1
        nw = min(n_frames, 4)
        rec_w = np.frombuffer(records[:nw].contiguous().cpu().numpy().tobytes(), dtype=hevcdl_amd.REC_DTYPE).reshape(nw, ctus)
        t1 = time.time()
        _, _, dumps = run_reference_pictures(list(yuv[:nw].cpu().numpy()), labels[:nw].cpu().numpy(), width, height, qp, nw, dump=True, wavefront=1)
        par = parity_against_dumps(dumps, rec_w, list(recon[:nw].cpu().numpy()), width, height)
        par["sample"] = "%d WHOLE %dx%d frames of the 600-frame wavefront step against the reference run with --WaveFrontSynchro=1, %.1f s" % (nw, width, height, time.time() - t1)
        out["parity_check"] = par
    out["e2e"] = e2e_leg(torch, hevcdl_amd, dev, enc, tensors, n_frames, width, height, qp, wavefront=True)      # the whole picture pipeline with the key, as the headline's e2e
    enc.close()
    # C2 with the key: 10 frames of 1080p
    w2, h2, n2 = 1920, 1080, 10
    e3 = hevcdl_amd.Encoder(w2, h2, qp, max_frames=n2, device=local, wavefront=True)
    y3 = synth_frames_torch(torch, dev, w2, h2, list(range(n2)), seed=2000)
    t3 = alloc(torch, hevcdl_amd, dev, n2, e3.frame_bytes, e3.ctus)
    el3, pr3 = timed_steps(torch, e3, (y3,) + t3, n2, 3, 1, barrier)
    out["c2"] = {"workload": "1920x1080 8-bit 4:2:0 all-intra QP%d, 10 frames (C2 of BASELINE.json) with WaveFrontSynchro 1" % qp, "value": 3 * n2 * e3.ctus / el3, "unit": "CTUs/s",
                 "ms_per_step": 1e3 * el3 / 3, "kernel_ms": pr3["rd_ms"] / 3, "launch": e3.last_rd_launch()}
    e3.close()
    del y3, t3
    return out


def alloc(torch, hevcdl_amd, dev, n, frame_bytes, ctus):
    return (torch.zeros((n, ctus, 16), dtype=torch.uint8, device=dev), torch.zeros((n, ctus, hevcdl_amd.REC_DTYPE.itemsize), dtype=torch.uint8, device=dev),
            torch.zeros((n, frame_bytes), dtype=torch.uint8, device=dev), torch.zeros((n, hevcdl_amd.STATS_DTYPE.itemsize), dtype=torch.uint8, device=dev))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--qp", type=int, default=32)
    ap.add_argument("--frames", type=int, default=600, help="frames of the job (C4: 600), split over the ranks")
    ap.add_argument("--saturated-frames", type=int, default=2560, help="extra single-GPU measurement with this many frames in flight (0: skip); 2560 = one frame per wave of the ten-wave build on 256 CUs")
    ap.add_argument("--weak-frames", type=int, default=600, help="N > 1: every rank also runs one step over this many frames of its own (weak scaling beside the strong headline); 0: skip")
    ap.add_argument("--no-projection", action="store_true", help="skip the 300 / 150 / 75-frame launches behind scale_projection / share_8gpu_s")
    ap.add_argument("--no-label-check", action="store_true")
    ap.add_argument("--no-c2", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-latency-floor", action="store_true")
    ap.add_argument("--no-wavefront", action="store_true", help="skip the extra measurements with WaveFrontSynchro 1 (key \"wavefront\"; the headline is the default cfg)")
    ap.add_argument("--cpu-procs", type=int, default=0)
    ap.add_argument("--cpu-baseline", default="reference", choices=["reference", "port"], help="reference: oracle/_ref/TAppEncoder_ref when present; port: the plain-C oracle")
    a = ap.parse_args()

    import torch
    import hevcdl_amd
    import hevcdl_amd.sharding as sharding

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    backend = os.environ.get("HEVCDL_BENCH_BACKEND", "nccl")      # "gloo": functional test of the N > 1 path with ranks sharing one GPU
    if backend != "nccl":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cdev = dev if backend == "nccl" else torch.device("cpu")       # where the (tiny) collective tensors live
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    W, H, qp, F = a.width, a.height, a.qp, a.frames
    mine = sharding.shard_frames(F, world, rank)                   # contiguous block of the job's frames
    Fr = len(mine)
    per_rank = sharding.max_shard(F, world)
    enc = hevcdl_amd.Encoder(W, H, qp, max_frames=max(1, per_rank), device=local)
    ctus = enc.ctus
    yuv = synth_frames_torch(torch, dev, W, H, list(mine), seed=1000)
    labels, records, recon, stats = alloc(torch, hevcdl_amd, dev, max(1, Fr), enc.frame_bytes, ctus)
    elapsed, prof = timed_steps(torch, enc, (yuv, labels, records, recon, stats), Fr, a.steps, a.warmup, barrier)
    rd_launch = enc.last_rd_launch()
    # one frame alone (outside the timed steps): the serial CTU chain of a frame bounds what any sharding of the job can reach
    floor_s = None
    if Fr > 0 and not a.no_latency_floor:
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        enc.encode_frames_dev(yuv.data_ptr(), 1, labels.data_ptr(), records.data_ptr(), recon.data_ptr(), stats.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize(dev)
        floor_s = time.perf_counter() - t1
        # (frame 0's records / reconstruction / statistics are rewritten with the values the whole job gave them: frames are independent, the kernel deterministic)

    # per-frame summaries (estimated bits) gathered to rank 0: the only collective of the path
    st = np.frombuffer(stats.cpu().numpy().tobytes(), dtype=hevcdl_amd.STATS_DTYPE)["est_bits"].astype(np.int64)[:Fr]
    stt = torch.zeros(per_rank, dtype=torch.int64)
    stt[:Fr] = torch.from_numpy(st.copy())
    stt = stt.to(cdev)
    ranks_seen = 1
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        ones = torch.ones(1, dtype=torch.int64, device=cdev)          # every rank adds one: the collective really spans `world` ranks (the line carries it as rccl_ranks)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        ranks_seen = int(ones.item())
        gathered = [torch.zeros_like(stt) for _ in range(world)] if rank == 0 else None
        dist.gather(stt, gathered, dst=0)
        total_bits = int(sum(int(g.sum().item()) for g in gathered)) if rank == 0 else 0
    else:
        total_bits = int(stt.sum().item())

    # Weak scaling beside the strong one (N > 1 only; outside the timed steps): the natural shard of this path is by frame, and 600 / N frames per GPU measures one
    # frame's latency rather than the machine -- so every rank also runs the WHOLE 600-frame shape on frames of its own (1 warm-up + 1 timed step between barriers,
    # max over ranks): value = N x 600 frames' CTUs / that time.
    weak = None
    if world > 1 and a.weak_frames > 0:
        Fw = a.weak_frames
        ew = hevcdl_amd.Encoder(W, H, qp, max_frames=Fw, device=local)
        yw = synth_frames_torch(torch, dev, W, H, [rank * Fw + i for i in range(Fw)], seed=1000)
        tw = alloc(torch, hevcdl_amd, dev, Fw, ew.frame_bytes, ctus)
        elw, prw = timed_steps(torch, ew, (yw,) + tw, Fw, 1, 1, barrier)
        tmax = torch.tensor([elw], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        weak = {"scaling": "weak", "frames_per_gpu": Fw, "n_gpus": world, "seconds": float(tmax.item()), "value": world * Fw * ctus / float(tmax.item()), "unit": "CTUs/s",
                "kernel_ms_rank0": prw["rd_ms"], "launch": ew.last_rd_launch(),
                "note": "every rank runs a full %d-frame step on frames of its own (1 warm-up, 1 timed step between barriers, max over ranks); no data-path collective" % Fw}
        del yw, tw
        ew.close()
        torch.cuda.empty_cache()
    # ... and the job itself once more with WaveFrontSynchro 1 (extra key, not the headline: DESIGN.md section 5e): the rank's share of the job's frames, rows as units
    wave_n = None
    if world > 1 and not a.no_wavefront:
        ewf = hevcdl_amd.Encoder(W, H, qp, max_frames=max(1, per_rank), device=local, wavefront=True)
        elf, prf = timed_steps(torch, ewf, (yuv, labels, records, recon, stats), Fr, 1, 1, barrier)
        tmax = torch.tensor([elf], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        wave_n = {"cfg": "WaveFrontSynchro 1 (all other keys as the headline)", "scaling": "strong", "n_gpus": world, "frames": F, "seconds": float(tmax.item()), "value": F * ctus / float(tmax.item()),
                  "unit": "CTUs/s", "kernel_ms_rank0": prf["rd_ms"], "launch": ewf.last_rd_launch(), "note": "1 warm-up + 1 timed step between barriers, max over ranks"}
        ewf.close()
        torch.cuda.empty_cache()

    if rank == 0:
        total_ctus = F * ctus * a.steps
        value = total_ctus / elapsed
        rd_avg_s = (prof["rd_ms"] / max(1, prof["rd_launches"])) / 1e3
        achieved = (ALGO_BYTES_PER_CTU * Fr * ctus / rd_avg_s) / 1e9 if rd_avg_s > 0 else 0.0
        traffic, traffic_src = measured_traffic(Fr * ctus)
        is_c4 = (W, H, qp, F) == (3840, 2160, 32, 600)
        out = {
            "metric": "all-intra CTUs/s at 2160p QP32", "value": value, "unit": "CTUs/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "decisions: u8 samples / int32 transforms / f64 costs (the reference's arithmetic); label CNN: f16 hi/lo split operands x3 on MFMA, f32 accumulate (logits within 1e-3 of fp32)", "data": "synthetic",
            "config": {"workload": "%dx%d 8-bit 4:2:0 all-intra QP%d, %d frames%s, frame-sharded (contiguous blocks of %d frames per GPU), on-device CNN labels + depth-pruned CTU decisions"
                                   % (W, H, qp, F, " (C4 of BASELINE.json)" if is_c4 else "", per_rank),
                       "frames": F, "frames_per_gpu": per_rank, "ctus_per_frame": ctus, "parallelism": "frame-shard x%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": rd_launch.split(" ")[0], "launch": rd_launch, "kernel_ms": 1e3 * rd_avg_s,      # (what the library's launch_rd chose: hevcdl_last_rd_launch)
                         "cnn_kernel_ms": prof["cnn_ms"] / max(1, prof["cnn_launches"]), "algorithmic_bytes_per_ctu": ALGO_BYTES_PER_CTU,
                         "units_per_launch": "%d frames x %d CTUs (rank 0)" % (Fr, ctus)},
            "est_bits_per_frame": total_bits / max(1, F),
            "rccl_ranks": ranks_seen, "collective_backend": backend if world > 1 else "none (one rank)", "frames_gathered": F if world == 1 else int(sum(int((g != 0).sum().item()) for g in gathered)),
        }
        if rd_avg_s > 0:      # the bound that can steer this kernel: vector-ALU issue (the HBM fraction is ~1e-3 by construction)
            out["roofline"]["issue"] = measured_issue(Fr, Fr * ctus / rd_avg_s)
        cnn_s = (prof["cnn_conv_ms"] / max(1, prof["cnn_launches"])) / 1e3 if "cnn_conv_ms" in prof else 0.0
        if cnn_s > 0:    # the other stage, the other bound: the convolution kernel against the dense f16 MFMA peak
            ex = 2.0 * CNN_CONV_MACS_EXECUTED * Fr * ctus / cnn_s / 1e12
            us = 2.0 * CNN_CONV_MACS * Fr * ctus / cnn_s / 1e12
            out["roofline_cnn"] = {"bound": "mfma", "kernel": "hevcdl_cnn_ctu_kernel", "operand": "f16 hi/lo split x3, f32 acc", "executed_tflops": ex, "useful_tflops": us,
                                   "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ex / MFMA_F16_PEAK_TFLOPS, "useful_frac": us / MFMA_F16_PEAK_TFLOPS, "kernel_ms": 1e3 * cnn_s,
                                   "executed_mflop_per_ctu": 2e-6 * CNN_CONV_MACS_EXECUTED, "useful_mflop_per_ctu": 2e-6 * CNN_CONV_MACS,
                                   "head_kernel_ms": prof["cnn_ms"] / max(1, prof["cnn_launches"]) - 1e3 * cnn_s,
                                   "note": "useful_frac = the graph's own multiply-accumulates (95.2 MFLOP per CTU) against the dense f16 peak: the distance from the roof.  frac prices the MFMA products ISSUED: three per multiply-accumulate in conv2 / conv3 (f16 hi/lo split operands for f32-like accuracy; the f32 MFMA runs at 1/16 of the f16 rate), 320 per output of the 5x5 layers (75 taps as 5 k-steps of 16 raw split words, two MFMAs a step): how busy the matrix pipes are; measured with HIP events on the launch stream over the timed steps"}
        if world == 1 and is_c4 and not a.no_projection:
            # What the frame-sharded job will take on N GPUs, from this GPU alone: a rank's share of the job is a launch of 600 / N frames (frames are independent, the
            # only collective gathers 8 bytes per frame), timed here as one whole step (label CNN + decisions) each.  The driver's SCALE run can be checked against it.
            proj = {1: elapsed / a.steps}
            stream = torch.cuda.current_stream().cuda_stream
            proj_launch = {1: rd_launch}
            for n_gpu in (2, 4, 8):
                share = sharding.max_shard(F, n_gpu)
                ts = []
                for rep in range(4):          # the first launch of a shape is a warm-up (workspace growth, the cooperative form's first use); then the median of three
                    torch.cuda.synchronize(dev)
                    t1 = time.perf_counter()
                    enc.encode_frames_dev(yuv.data_ptr(), share, labels.data_ptr(), records.data_ptr(), recon.data_ptr(), stats.data_ptr(), stream)
                    torch.cuda.synchronize(dev)
                    if rep:
                        ts.append(time.perf_counter() - t1)
                proj[n_gpu] = sorted(ts)[1]
                proj_launch[n_gpu] = enc.last_rd_launch()
            out["scale_projection"] = {"seconds": {str(k): v for k, v in proj.items()}, "value": {str(k): F * ctus / v for k, v in proj.items()}, "unit": "CTUs/s",
                                       "launch": {str(k): v for k, v in proj_launch.items()},
                                       "note": "seconds of one step over a rank's share of the 600 frames (600 / 300 / 150 / 75 frames) on this GPU: one warm-up launch per share, then the median of three; "
                                               "N-GPU value = 1 224 000 CTUs / that time"}
            out["share_8gpu_s"] = proj[8]
            # the weak-scaling counterpart: %d frames PER GPU is this run's own step on every GPU at once (frames are independent, nothing is exchanged but 8 bytes
            # per frame at the end), so N GPUs process N x the job in the same time -- a projection by construction; `bench.py --gpus N` measures it (key "weak")
            out["scale_projection_weak"] = {"frames_per_gpu": F, "seconds": elapsed / a.steps, "value": {str(n): n * F * ctus / (elapsed / a.steps) for n in (1, 2, 4, 8)}, "unit": "CTUs/s",
                                            "note": "weak scaling: every GPU runs this run's own %d-frame step on frames of its own; measured by `bench.py --gpus N` as key \"weak\"" % F}
            # (the shares' records / reconstruction equal what the whole job wrote for those frames: frames are independent, the kernel deterministic -- re-run the job's step so that
            #  the legs below see the state of the timed job)
            enc.encode_frames_dev(yuv.data_ptr(), Fr, labels.data_ptr(), records.data_ptr(), recon.data_ptr(), stats.data_ptr(), stream)
            torch.cuda.synchronize(dev)
        if weak is not None:
            out["weak"] = weak
        if wave_n is not None:
            out["wavefront"] = wave_n
        if floor_s:
            out["latency_floor_s"] = floor_s
            out["strong_scaling_ceiling"] = {"value": F * ctus / floor_s, "unit": "CTUs/s",
                                             "note": "the whole job in the time one frame takes alone on a GPU (rank 0: %.3f s): more GPUs than frames-per-CU can use do not help" % floor_s}
        if world == 1:
            if not a.no_cpu_baseline:       # the CPU baseline is timed on rank 0 of the single-GPU run only
                nb = min(Fr, 256)                      # frames sampled for the parity check
                cx = (W + 63) // 64
                rows = 6
                if os.path.exists(REF_ENC) and a.cpu_baseline != "port":
                    rec_h = np.frombuffer(records[:nb, :cx * rows].contiguous().cpu().numpy().tobytes(), dtype=hevcdl_amd.REC_DTYPE).reshape(nb, cx * rows)
                    nw = min(Fr, a.cpu_procs or effective_cores())          # the frames the timed CPU leg encodes whole
                    rec_w = np.frombuffer(records[:nw].contiguous().cpu().numpy().tobytes(), dtype=hevcdl_amd.REC_DTYPE).reshape(nw, ctus)
                    cb, parity = cpu_baseline_reference(yuv[:nb].cpu().numpy(), labels[:nb].cpu().numpy(), W, H, qp, a.cpu_procs or None, rows,
                                                        gpu_records=rec_h, gpu_recon=recon[:nb].cpu().numpy(), gpu_records_full=rec_w)
                else:     # the reference build travels with the repository (oracle/_ref); without it the plain-C port stands in
                    cb, parity = cpu_baseline_port(yuv[:nb].cpu().numpy(), labels[:nb].cpu().numpy(), W, H, qp, a.cpu_procs or None)
                out["cpu_baseline"] = cb
                if parity is not None:
                    out["parity_check"] = parity
            if not a.no_label_check:      # the labels of the split-f16 MFMA kernel against the same graph in fp32 (checker: oracle/cnn_torch.py), outside the timed region
                sys.path.insert(0, os.path.join(ROOT, "oracle"))
                import cnn_oracle
                import cnn_torch
                nl = min(Fr, 16)
                t1 = time.perf_counter()
                chk = cnn_torch.label_check(torch, cnn_oracle.load_weights(hevcdl_amd.WEIGHTS_PATH), yuv[:nl].cpu().numpy(), W, H, dev, labels[:nl].cpu().numpy())
                chk["sample"] = "the first %d frames of the timed job, %.1f s" % (nl, time.perf_counter() - t1)
                out["cnn_label_check"] = chk
            if not a.no_e2e:
                out["e2e"] = e2e_leg(torch, hevcdl_amd, dev, enc, (yuv, labels, records, recon, stats), Fr, W, H, qp)
                if "cpu_baseline" in out and out["cpu_baseline"].get("kind") == "reference":
                    out["e2e"]["over_cpu_baseline"] = out["e2e"]["value"] / out["cpu_baseline"]["value"]
                    out["e2e"]["stages_cpu"] = REF_STAGES
            if not a.no_wavefront and is_c4:
                try:      # (extra keys: whatever happens in this leg, the headline line is printed)
                    out["wavefront"] = wavefront_leg(torch, hevcdl_amd, dev, local, (yuv, labels, records, recon, stats), Fr, W, H, qp, ctus, barrier,
                                                     check=(not a.no_cpu_baseline and os.path.exists(REF_ENC) and a.cpu_baseline != "port"))
                except Exception as exc:
                    out["wavefront"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
            del yuv, labels, records, recon, stats
            enc.close()
            torch.cuda.empty_cache()
            if a.saturated_frames > 0:      # every wave of the chip owns a frame: what the kernel does when the job is large enough
                S = a.saturated_frames
                e2 = hevcdl_amd.Encoder(W, H, qp, max_frames=S, device=local)
                y2 = synth_frames_torch(torch, dev, W, H, list(range(min(S, 64))), seed=1000)
                y2 = y2.repeat((S + y2.shape[0] - 1) // y2.shape[0], 1)[:S].contiguous()
                t2 = alloc(torch, hevcdl_amd, dev, S, e2.frame_bytes, ctus)
                el2, pr2 = timed_steps(torch, e2, (y2,) + t2, S, 1, 0, barrier)
                out["saturated"] = {"frames_per_gpu": S, "value": S * ctus / el2, "unit": "CTUs/s", "ms_per_step": 1e3 * el2, "kernel_ms": pr2["rd_ms"], "cnn_kernel_ms": pr2["cnn_ms"],
                                    "per_cu": S * ctus / el2 / torch.cuda.get_device_properties(dev).multi_processor_count,
                                    "issue": measured_issue(S, S * ctus / (pr2["rd_ms"] / 1e3)) if pr2["rd_ms"] > 0 else None,
                                    "note": "one step, %d distinct frames repeated; a frame per wave of the ten-wave build of the decision kernel (csrc/rd_kernel_wide.hip); not the "
                                            "headline: the job of BASELINE.json has 600 frames" % min(S, 64)}
                del y2, t2
                e2.close()
                torch.cuda.empty_cache()
            if not a.no_c2:                 # C2 of BASELINE.json: 10 frames of 1080p on one GPU, the reference on 10 host cores beside it
                w2, h2, n2 = 1920, 1080, 10
                e3 = hevcdl_amd.Encoder(w2, h2, qp, max_frames=n2, device=local)
                y3 = synth_frames_torch(torch, dev, w2, h2, list(range(n2)), seed=2000)
                t3 = alloc(torch, hevcdl_amd, dev, n2, e3.frame_bytes, e3.ctus)
                el3, pr3 = timed_steps(torch, e3, (y3,) + t3, n2, 3, 1, barrier)
                c2 = {"workload": "1920x1080 8-bit 4:2:0 all-intra QP%d, 10 frames (C2 of BASELINE.json)" % qp, "value": 3 * n2 * e3.ctus / el3, "unit": "CTUs/s",
                      "ms_per_step": 1e3 * el3 / 3, "kernel_ms": pr3["rd_ms"] / 3, "cnn_kernel_ms": pr3["cnn_ms"] / 3}
                if not a.no_cpu_baseline and os.path.exists(REF_ENC) and a.cpu_baseline != "port":
                    ww, perw, _ = run_reference_pictures(list(y3.cpu().numpy()), t3[0].cpu().numpy(), w2, h2, qp, n2)
                    c2["cpu_reference"] = {"value": n2 * e3.ctus / ww, "unit": "CTUs/s", "cores": n2, "sample": "the same 10 frames, one reference-encoder process per frame, %.1f s wall" % ww,
                                           "stages_cpu": REF_STAGES, "stages_gpu": GPU_STAGES}
                    c2["gpu_over_cpu"] = c2["value"] / c2["cpu_reference"]["value"]
                out["c2"] = c2
                if "c2" in out.get("wavefront", {}) and "cpu_reference" in c2:      # (the reference's own time does not depend on the key: one process per frame either way)
                    out["wavefront"]["c2"]["gpu_over_cpu_reference_default_cfg"] = out["wavefront"]["c2"]["value"] / c2["cpu_reference"]["value"]
                e3.close()
        print(json.dumps(out), flush=True)
    if world > 1 or rank != 0:
        enc.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
