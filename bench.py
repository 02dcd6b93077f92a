#!/usr/bin/env python3
"""bench.py -- all-intra CTUs/s of the hot path (on-device CNN depth predictor + depth-pruned CTU decision kernel).

Contract: `python bench.py --gpus N --steps K --warmup W`; for N>1 launched by torch.distributed.run, one rank per GPU.
A step = one pass of the whole hot path (hevcdl_encode_frames_dev: CNN + RD search) over one batch of synthetic frames
that is already resident in HBM.  Frames are independent, so ranks shard by frame with no data-path collective; RCCL
only gathers the per-frame rate/SSE records (weak scaling: per-GPU work is fixed).
Prints ONE JSON line on rank 0 (metric of BASELINE.json + roofline + cpu_baseline).
"""
import argparse
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


def synth_frames_torch(torch, dev, width, height, n_frames, seed):
    """Synthetic planar 4:2:0 frames generated on the device (generator of SURVEY.md section 8d)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    y = torch.arange(height, device=dev, dtype=torch.float32).view(1, height, 1)
    x = torch.arange(width, device=dev, dtype=torch.float32).view(1, 1, width)
    f = torch.arange(n_frames, device=dev, dtype=torch.float32).view(n_frames, 1, 1)
    out = torch.empty((n_frames, width * height * 3 // 2), dtype=torch.uint8, device=dev)
    chunk = 8
    for s in range(0, n_frames, chunk):
        e = min(n_frames, s + chunk)
        ff = f[s:e]
        Y = 128 + 50 * torch.sin(x / 57) * torch.cos(y / 43) + 30 * torch.sin((x + y + 3 * ff) / 19)
        Y = Y + torch.randn((e - s, height, width), device=dev, generator=g) * 5
        bw, bh = min(1200, width // 3), min(600, height // 3)
        for k in range(s, e):
            bx = (width // 5 + 4 * k) % max(1, width - bw)
            by = height // 4
            blk = Y[k - s, by:by + bh, bx:bx + bw]
            Y[k - s, by:by + bh, bx:bx + bw] = torch.floor(blk / 24) * 24
        Y = Y.clamp(0, 255).to(torch.uint8)
        xc, yc = x[:, :, ::2], y[:, ::2, :]
        U = (128 + 25 * torch.sin(xc / 61)).expand(e - s, height // 2, width // 2).clamp(0, 255).to(torch.uint8)
        V = (128 + 25 * torch.cos(yc / 47)).expand(e - s, height // 2, width // 2).clamp(0, 255).to(torch.uint8)
        out[s:e] = torch.cat([Y.reshape(e - s, -1), U.reshape(e - s, -1), V.reshape(e - s, -1)], dim=1)
    return out


def _cpu_worker(args):
    yuv, w, h, qp, labels = args
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_tools
    t = time.time()
    ref_tools.run_oracle(yuv, w, h, qp, labels)
    return time.time() - t


REF_ENC = os.path.join(ROOT, "oracle", "_ref", "TAppEncoder_ref")


def _prepare_ref_run(args):
    """Working directory of one reference-encoder process: input band, the label files it polls for (TEncCu.cpp:244-253), output dir."""
    idx, yuv, w, h, qp, labels, base = args
    d = os.path.join(base, "p%d" % idx)
    os.makedirs(os.path.join(d, "rec"))
    yuv.astype(np.uint8).tofile(os.path.join(d, "in.yuv"))
    os.makedirs(os.path.join(d, "pred", "0"))
    for a in range(labels.shape[1]):
        with open(os.path.join(d, "pred", "0", "ctu%d.txt" % a), "w") as fh:
            fh.write(" ".join(str(int(v)) for v in labels[0, a]) + " ")
    return d


def _run_ref(args):
    import subprocess
    d, cmd = args
    t = time.time()
    r = subprocess.run(cmd, cwd=d, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("reference encoder failed: " + r.stdout[-500:] + r.stderr[-500:])
    return time.time() - t


def cpu_baseline_reference(yuv_host, labels_host, width, height, qp, max_procs=None, band_rows=6):
    """The reference itself (oracle/_ref/TAppEncoder_ref, built from /root/reference by oracle/build_ref.sh; configuration = the
    reference's cfg as switches, oracle/ref_args.py) on the host cores: P processes, each encoding the top `band_rows` CTU rows of a
    distinct frame with its labels already on disk, wall clock from first start to last exit.  Encoder only (in-loop filters and bitstream
    included, as the reference runs them); the label CNN is excluded."""
    import shutil
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_args
    cores = os.cpu_count() or 1
    p = min(cores, yuv_host.shape[0], max_procs or cores)
    full_h = height
    height = min(height, 64 * band_rows)
    cx = (width + 63) // 64
    ysz, csz = width * full_h, (width // 2) * (full_h // 2)

    def band(fr):
        return np.concatenate([fr[:width * height], fr[ysz:ysz + (width // 2) * (height // 2)], fr[ysz + csz:ysz + csz + (width // 2) * (height // 2)]])
    labels_host = np.ascontiguousarray(labels_host[:p, :cx * ((height + 63) // 64)])
    base = tempfile.mkdtemp(prefix="hevcdl_cpu_")
    try:
        dirs = [_prepare_ref_run((i, band(yuv_host[i]), width, height, qp, labels_host[i:i + 1], base)) for i in range(p)]
        cmd = [REF_ENC, "-i", "in.yuv", "-b", "rec/str.bin", "-o", "rec/rec.yuv"] + ref_args.reference_args(width, height, 1, qp)
        t = time.time()
        with ThreadPoolExecutor(max_workers=p) as pool:
            per = list(pool.map(_run_ref, [(d, cmd) for d in dirs]))
        wall = time.time() - t
    finally:
        shutil.rmtree(base, ignore_errors=True)
    ctus = p * labels_host.shape[1]
    return {"value": ctus / wall, "unit": "CTUs/s", "cores": p, "kind": "reference",
            "sample": "top %dx%d band of %d frames of the workload, QP%d (1 reference-encoder process per band, label files from the GPU CNN, CNN excluded; deblocking, SAO and bitstream included), %.1f s wall, %.1f CTUs/s per core"
                      % (width, height, p, qp, wall, ctus / sum(per) if sum(per) > 0 else 0.0)}


def cpu_baseline(yuv_host, labels_host, width, height, qp, max_procs=None, band_rows=3):
    """Oracle (CPU port, bit-identical to the reference on the golden vectors) timed on the host cores of this node:
    P processes, each encoding the top `band_rows` CTU rows of a distinct frame of the same workload (bounded sample:
    a full 2160p frame per process would take minutes), wall clock from first start to last exit."""
    import __graft_entry__ as g
    g.build_oracle()
    cores = os.cpu_count() or 1
    p = min(cores, yuv_host.shape[0], max_procs or cores)
    full_h = height
    height = min(height, 64 * band_rows)
    cx = (width + 63) // 64
    ysz, csz = width * full_h, (width // 2) * (full_h // 2)

    def band(fr):
        return np.concatenate([fr[:width * height], fr[ysz:ysz + (width // 2) * (height // 2)], fr[ysz + csz:ysz + csz + (width // 2) * (height // 2)]])
    yuv_host = np.stack([band(yuv_host[i]) for i in range(p)])
    labels_host = np.ascontiguousarray(labels_host[:p, :cx * ((height + 63) // 64)])
    jobs = [(yuv_host[i:i + 1], width, height, qp, labels_host[i:i + 1]) for i in range(p)]
    t = time.time()
    with mp.get_context("spawn").Pool(p) as pool:
        per = pool.map(_cpu_worker, jobs)
    wall = time.time() - t
    ctus = p * labels_host.shape[1]
    return {"value": ctus / wall, "unit": "CTUs/s", "cores": p, "kind": "port",
            "sample": "top %dx%d band of %d frames of the workload, QP%d (1 band per process, labels from the GPU CNN, CNN excluded), %.1f s wall, %.1f CTUs/s per core"
                      % (width, height, p, qp, wall, ctus / sum(per) if sum(per) > 0 else 0.0)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--qp", type=int, default=32)
    ap.add_argument("--frames", type=int, default=2048, help="frames per GPU per step (weak scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-procs", type=int, default=0)
    ap.add_argument("--cpu-baseline", default="reference", choices=["reference", "port"], help="reference: oracle/_ref/TAppEncoder_ref when present; port: the plain-C oracle")
    a = ap.parse_args()

    import torch
    import hevcdl_amd

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

    W, H, qp, F = a.width, a.height, a.qp, a.frames
    enc = hevcdl_amd.Encoder(W, H, qp, max_frames=F, device=local)
    ctus = enc.ctus
    yuv = synth_frames_torch(torch, dev, W, H, F, seed=1000 + rank)
    labels = torch.zeros((F, ctus, 16), dtype=torch.uint8, device=dev)
    records = torch.zeros((F, ctus, hevcdl_amd.REC_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    recon = torch.zeros_like(yuv)
    stats = torch.zeros((F, hevcdl_amd.STATS_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev)

    def step():
        enc.encode_frames_dev(yuv.data_ptr(), F, labels.data_ptr(), records.data_ptr(), recon.data_ptr(), stats.data_ptr(), stream.cuda_stream)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(a.warmup):
        step()
    barrier()
    enc.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    prof = enc.profile_get()
    enc.profile_enable(False)

    # per-frame summaries (bits, SSE) gathered to rank 0: the only collective of the path
    st = torch.from_numpy(np.frombuffer(stats.cpu().numpy().tobytes(), dtype=hevcdl_amd.STATS_DTYPE)["est_bits"].astype(np.int64)).to(cdev)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        gathered = [torch.zeros_like(st) for _ in range(world)] if rank == 0 else None
        dist.gather(st, gathered, dst=0)
        total_bits = int(sum(int(g.sum().item()) for g in gathered)) if rank == 0 else 0
    else:
        total_bits = int(st.sum().item())

    if rank == 0:
        total_ctus = world * F * ctus * a.steps
        value = total_ctus / elapsed
        rd_avg_s = (prof["rd_ms"] / max(1, prof["rd_launches"])) / 1e3
        achieved = (ALGO_BYTES_PER_CTU * F * ctus / rd_avg_s) / 1e9 if rd_avg_s > 0 else 0.0
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "r01k_traffic.json")
        if os.path.exists(tpath):   # PMC counters cannot be collected from inside the process: per-CTU bytes of the committed rocprofv3 passes
            tj = json.load(open(tpath))
            traffic = (tj["fetch_bytes_per_ctu"] + tj["write_bytes_per_ctu"]) * F * ctus
            traffic_src = tj["source"] + "; " + tj["note"]
        out = {
            "metric": "all-intra CTUs/s at 2160p QP32", "value": value, "unit": "CTUs/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/int32/f64", "data": "synthetic",
            "config": {"workload": "%dx%d 8-bit 4:2:0 all-intra QP%d, %d frames per GPU per step (frame-sharded; C4 of BASELINE.json is 75/GPU at 8 GPUs), on-device CNN labels + depth-pruned CTU decisions" % (W, H, qp, F),
                       "frames_per_gpu": F, "ctus_per_frame": ctus, "parallelism": "frame-shard x%d" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "kernel": "hevcdl_rd_frame_kernel", "kernel_ms": 1e3 * rd_avg_s,
                         "cnn_kernel_ms": prof["cnn_ms"] / max(1, prof["cnn_launches"]), "algorithmic_bytes_per_ctu": ALGO_BYTES_PER_CTU},
            "est_bits_per_frame": total_bits / max(1, world * F),
        }
        if not a.no_cpu_baseline and world == 1:       # the CPU baseline is timed on rank 0 of the single-GPU run only
            nb = min(F, os.cpu_count() or 1)
            # the reference build travels with the repository (oracle/_ref); without it the plain-C port stands in
            base_fn = cpu_baseline_reference if os.path.exists(REF_ENC) and a.cpu_baseline != "port" else cpu_baseline
            out["cpu_baseline"] = base_fn(yuv[:nb].cpu().numpy(), labels[:nb].cpu().numpy(), W, H, qp, a.cpu_procs or None)
        print(json.dumps(out), flush=True)
    enc.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
