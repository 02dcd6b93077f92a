"""Whole-sequence encode over one or more GPUs (one process per GPU): the host loop of the reference's TAppEncTop::encode /
TEncGOP::compressGOP for the all-intra configuration, on top of the C ABI.

Per batch of pictures: CNN labels (or label files) -> CTU decisions -> deblocking -> SAO -> access units (+ MD5 SEI) and the
reference's log line.  Across ranks the pictures shard by contiguous frame ranges (`sharding.shard_frames`): every rank writes
the access units of its own pictures; rank 0 concatenates the parts in POC order (all-intra pictures are independent: IDR at POC 0,
CRA afterwards, parameter sets re-sent with every picture, so the shards ARE the single-process stream) and prints the summary from
the gathered per-frame rows (`sharding.gather_frame_summaries`) -- the only collective, 40 bytes per picture.
"""
import os
import time

import numpy as np

from . import Encoder, write_access_unit, picture_hash_sei
from . import metrics, sharding


_POOL = None


def _pool():
    global _POOL
    if _POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 1))
    return _POOL


def read_frames(path, width, height, first, count, bit_depth=8):
    """Planar 4:2:0 frames [count, w*h*3/2] (uint8, or uint16 little endian for bit_depth 10) starting at frame `first`."""
    dt = np.uint8 if bit_depth == 8 else np.dtype("<u2")
    n = width * height * 3 // 2
    with open(path, "rb") as f:
        f.seek(first * n * (1 if bit_depth == 8 else 2))
        a = np.fromfile(f, dt, n * count)
    if a.size != n * count:
        raise ValueError("short read of %s: wanted %d frames from frame %d" % (path, count, first))
    return a.reshape(count, n)


def encode_sequence(input_path, width, height, qp, n_frames, bitstream_path=None, recon_path=None, frame_skip=0, batch=256, tiles=(1, 1),
                    lf_across_tiles=True, bit_depth=8, level_idc=186, frame_rate=30.0, hash_sei=False, labels_fn=None, device=None, log=print, tools=0x7f, wavefront=False):
    """Encode frames [frame_skip, frame_skip + n_frames) of a planar YUV file.  Works stand-alone and under torch.distributed
    (initialised by the caller): rank r takes a contiguous share of the frames.  Returns, on rank 0, the summary (metrics.Summary)
    and the list of per-picture rows [poc, bits, sseY, sseU, sseV]; other ranks return (None, None).
    labels_fn(first_poc, count) -> uint8 [count, ctus, 16] replaces the on-device CNN (the reference's label files)."""
    import torch
    import torch.distributed as dist
    multi = dist.is_available() and dist.is_initialized()
    rank, world = (dist.get_rank(), dist.get_world_size()) if multi else (0, 1)
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    mine = sharding.shard_frames(n_frames, world, rank)
    per_rank = sharding.max_shard(n_frames, world)
    rows = np.full((per_rank + 1, 5), -1, np.int64)
    part_bits = (bitstream_path + ".part%d" % rank) if bitstream_path else None
    part_rec = (recon_path + ".part%d" % rank) if recon_path else None
    fb, fr = open(part_bits, "wb") if part_bits else None, open(part_rec, "wb") if part_rec else None
    if len(mine):
        enc = Encoder(width, height, qp, max_frames=min(batch, len(mine)), device=device, tiles=tiles, bit_depth=bit_depth, lf_across_tiles=lf_across_tiles, tools=tools, wavefront=wavefront)      # tools: HEVCDL_TOOL_*; wavefront: WaveFrontSynchro mask (the cfg's tool switches)
        ysz = width * height
        for b0 in range(mine.start, mine.stop, batch):
            nb = min(batch, mine.stop - b0)
            yuv = read_frames(input_path, width, height, frame_skip + b0, nb, bit_depth)
            t0 = time.time()
            labels = labels_fn(b0, nb) if labels_fn else None
            recs, final, sao, _ = enc.encode_pictures(yuv, labels)          # CNN -> decisions -> deblocking -> SAO, pictures stay in HBM
            et = (time.time() - t0) / nb
            def one_picture(i):          # host work of a picture (arithmetic coder, hash, SSE): independent -> thread pool (ctypes drops the GIL)
                au = write_access_unit(width, height, qp, b0 + i, recs[i], level_idc=level_idc, sao=sao[i], tiles=tiles, bit_depth=bit_depth, lf_across_tiles=lf_across_tiles, tools=tools, wavefront=wavefront)
                sei = picture_hash_sei(width, height, final[i], bit_depth) if hash_sei else b""
                d = (yuv[i].astype(np.int64) - final[i].astype(np.int64)) ** 2
                return au, sei, [int(d[:ysz].sum()), int(d[ysz:ysz + ysz // 4].sum()), int(d[ysz + ysz // 4:].sum())]
            for i, (au, sei, sse) in enumerate(_pool().map(one_picture, range(nb))):
                poc = b0 + i
                if fb:
                    fb.write(au + sei)
                rows[poc - mine.start] = [poc, len(au) * 8] + sse
            if fr:
                final.tofile(fr)
            log("rank %d: pictures %d..%d encoded (%.2f s per picture)" % (rank, b0, b0 + nb - 1, et))
        enc.close()
    for f in (fb, fr):
        if f:
            f.close()
    if multi:
        t = torch.from_numpy(rows)
        allrows = sharding.gather_frame_summaries(t.to(torch.device("cuda", device)) if dist.get_backend() == "nccl" else t)     # the rank's own device, not torch's current one
        dist.barrier()
    else:
        allrows = rows[rows[:, 0] >= 0]
    if rank != 0:
        return None, None
    for path in (bitstream_path, recon_path):                      # parts in rank order = POC order (contiguous shards)
        if path:
            with open(path, "wb") as out:
                for r in range(world):
                    with open(path + ".part%d" % r, "rb") as part:
                        while True:
                            chunk = part.read(1 << 24)
                            if not chunk:
                                break
                            out.write(chunk)
                    os.remove(path + ".part%d" % r)
    summ = metrics.Summary(width, height, frame_rate, bit_depth)
    for poc, bits, sy, su, sv in allrows:
        p = summ.add(int(bits), (int(sy), int(su), int(sv)))
        log(metrics.frame_line(int(poc), qp, int(bits), p))
    log("\n\nSUMMARY --------------------------------------------------------")
    log(summ.text())
    return summ, allrows


def encode_sequence_tile_sharded(input_path, width, height, qp, n_frames, tiles, bitstream_path=None, recon_path=None, frame_skip=0, batch=None, lf_across_tiles=True,
                                 bit_depth=8, level_idc=186, frame_rate=30.0, hash_sei=False, device=None, log=print, tools=0x7f):
    """The within-picture partition: every rank decides its share of the TILES of every picture of a batch
    (`hevcdl_compress_tiles_dev`), one all-to-all moves the tile payloads to the picture's owner (`sharding.exchange_tiles_to_owners`:
    picture i of the batch -> rank i mod world), the owner runs the in-loop filters on the assembled picture and writes its access unit.
    Needs torch.distributed initialised; batch (default: world size) is a multiple of the world size.  Output == encode_sequence()."""
    import torch
    import torch.distributed as dist
    from . import REC_DTYPE, SAO_DTYPE
    rank, world = dist.get_rank(), dist.get_world_size()
    via_host = dist.get_backend() != "nccl"                         # gloo moves host tensors
    batch = batch or world
    if batch % world:
        raise ValueError("batch must be a multiple of the world size")
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    dev = torch.device("cuda", device)
    from . import tile_layout
    tl = tile_layout(tiles, width, height)                          # (columns, rows) or explicit CTU sizes
    n_tiles = tl[0] * tl[1]
    t_begin, t_count = sharding.shard_tiles(n_tiles, world, rank)
    enc = Encoder(width, height, qp, max_frames=batch, device=device, tiles=tiles, bit_depth=bit_depth, lf_across_tiles=lf_across_tiles, tools=tools)
    ctus, fsamp, ysz = enc.ctus, width * height * 3 // 2, width * height
    bps = 1 if bit_depth == 8 else 2
    part_bits = (bitstream_path + ".part%d" % rank) if bitstream_path else None
    part_rec = (recon_path + ".part%d" % rank) if recon_path else None
    fb, fr = open(part_bits, "wb") if part_bits else None, open(part_rec, "wb") if part_rec else None
    rows = []
    for b0 in range(0, n_frames, batch):
        nb = min(batch, n_frames - b0)
        yuv = read_frames(input_path, width, height, frame_skip + b0, nb, bit_depth)
        if nb < batch:                                              # last batch: repeat the last picture, the copies are dropped below
            yuv = np.concatenate([yuv, np.repeat(yuv[-1:], batch - nb, axis=0)])
        d_yuv = torch.from_numpy(yuv.view(np.uint8).reshape(batch, fsamp * bps)).to(dev)
        d_lab = torch.zeros((batch, ctus, 16), dtype=torch.uint8, device=dev)
        d_recs = torch.zeros((batch, ctus * REC_DTYPE.itemsize), dtype=torch.uint8, device=dev)
        d_recon = torch.zeros_like(d_yuv)
        enc._check(enc.lib.hevcdl_predict_depth_dev(enc._h, d_yuv.data_ptr(), batch, d_lab.data_ptr(), None, None))
        enc.compress_tiles_dev(d_yuv.data_ptr(), batch, d_lab.data_ptr(), d_recs.data_ptr(), d_recon.data_ptr(), None, t_begin, t_count)
        torch.cuda.synchronize(dev)
        if via_host:
            owned, o_recon, o_recs = sharding.exchange_tiles_to_owners(d_recon.cpu(), d_recs.cpu(), width, height, tiles, bps=bps)
            o_recon, o_recs = o_recon.to(dev), o_recs.to(dev)
        else:
            owned, o_recon, o_recs = sharding.exchange_tiles_to_owners(d_recon, d_recs, width, height, tiles, bps=bps)
        no = len(owned)
        o_org = d_yuv[torch.tensor(owned, device=dev)].contiguous()
        o_dbk = torch.zeros_like(o_recon)
        o_fin = torch.zeros_like(o_recon)
        o_sao = torch.zeros((no, ctus * 3 * SAO_DTYPE.itemsize), dtype=torch.uint8, device=dev)
        enc.deblock_frames_dev(o_recon.data_ptr(), no, o_recs.data_ptr(), o_dbk.data_ptr())
        enc.sao_frames_dev(o_org.data_ptr(), o_dbk.data_ptr(), no, o_sao.data_ptr(), o_fin.data_ptr())
        torch.cuda.synchronize(dev)
        recs = np.frombuffer(o_recs.cpu().numpy().tobytes(), REC_DTYPE).reshape(no, ctus)
        sao = np.frombuffer(o_sao.cpu().numpy().tobytes(), SAO_DTYPE).reshape(no, ctus, 3)
        final = np.frombuffer(o_fin.cpu().numpy().tobytes(), np.uint8 if bit_depth == 8 else "<u2").reshape(no, fsamp)
        for j, i in enumerate(owned):
            poc = b0 + i
            if i >= nb:
                continue
            au = write_access_unit(width, height, qp, poc, recs[j], level_idc=level_idc, sao=sao[j], tiles=tiles, bit_depth=bit_depth, lf_across_tiles=lf_across_tiles, tools=tools)
            blob = au + (picture_hash_sei(width, height, final[j], bit_depth) if hash_sei else b"")
            if fb:
                fb.write(blob)
            if fr:
                final[j].tofile(fr)
            d = (yuv[i].astype(np.int64) - final[j].astype(np.int64)) ** 2
            rows.append([poc, len(au) * 8, int(d[:ysz].sum()), int(d[ysz:ysz + ysz // 4].sum()), int(d[ysz + ysz // 4:].sum()), len(blob)])
        log("pictures %d..%d: tiles [%d, %d) of %d decided on rank %d" % (b0, b0 + nb - 1, t_begin, t_begin + t_count, n_tiles, rank))
    enc.close()
    for f in (fb, fr):
        if f:
            f.close()
    cap = (n_frames + world - 1) // world + batch
    t = torch.full((cap, 6), -1, dtype=torch.int64)
    if rows:
        t[:len(rows)] = torch.tensor(rows, dtype=torch.int64)
    allrows = sharding.gather_frame_summaries(t.to(dev) if not via_host else t)
    dist.barrier()
    if rank != 0:
        return None, None
    # merge the per-rank parts in POC order: every part holds its pictures in increasing POC
    owner_of = {int(r[0]): ((int(r[0]) % batch) % world) for r in allrows}
    size_of = {int(r[0]): int(r[5]) for r in allrows}
    for path, sizes in ((bitstream_path, size_of), (recon_path, None)):
        if not path:
            continue
        parts = [open(path + ".part%d" % r, "rb") for r in range(world)]
        with open(path, "wb") as out:
            for poc in range(n_frames):
                out.write(parts[owner_of[poc]].read(sizes[poc] if sizes else fsamp * bps))
        for r, fpart in enumerate(parts):
            fpart.close(); os.remove(path + ".part%d" % r)
    summ = metrics.Summary(width, height, frame_rate, bit_depth)
    for poc, bits, sy, su, sv, _ in allrows:
        p = summ.add(int(bits), (int(sy), int(su), int(sv)))
        log(metrics.frame_line(int(poc), qp, int(bits), p))
    log("\n\nSUMMARY --------------------------------------------------------")
    log(summ.text())
    return summ, allrows
