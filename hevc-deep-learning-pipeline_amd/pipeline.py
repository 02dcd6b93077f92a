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


def encode_sequence(input_path, width, height, qp, n_frames, bitstream_path=None, recon_path=None, frame_skip=0, batch=16, tiles=(1, 1),
                    bit_depth=8, level_idc=186, frame_rate=30.0, hash_sei=False, labels_fn=None, device=None, log=print):
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
    per_rank = len(sharding.shard_frames(n_frames, world, 0))
    rows = np.full((per_rank + 1, 5), -1, np.int64)
    part_bits = (bitstream_path + ".part%d" % rank) if bitstream_path else None
    part_rec = (recon_path + ".part%d" % rank) if recon_path else None
    fb, fr = open(part_bits, "wb") if part_bits else None, open(part_rec, "wb") if part_rec else None
    if len(mine):
        enc = Encoder(width, height, qp, max_frames=min(batch, len(mine)), device=device, tiles=tiles, bit_depth=bit_depth)
        ysz = width * height
        for b0 in range(mine.start, mine.stop, batch):
            nb = min(batch, mine.stop - b0)
            yuv = read_frames(input_path, width, height, frame_skip + b0, nb, bit_depth)
            t0 = time.time()
            labels = labels_fn(b0, nb) if labels_fn else None
            recs, recon, _ = enc.compress_frames(yuv, labels)
            dbk = enc.deblock_frames(recon, recs)
            sao, final = enc.sao_frames(yuv, dbk)
            et = (time.time() - t0) / nb
            for i in range(nb):
                poc = b0 + i
                au = write_access_unit(width, height, qp, poc, recs[i], level_idc=level_idc, sao=sao[i], tiles=tiles, bit_depth=bit_depth)
                if fb:
                    fb.write(au)
                    if hash_sei:
                        fb.write(picture_hash_sei(width, height, final[i], bit_depth))
                d = (yuv[i].astype(np.int64) - final[i].astype(np.int64)) ** 2
                rows[poc - mine.start] = [poc, len(au) * 8, int(d[:ysz].sum()), int(d[ysz:ysz + ysz // 4].sum()), int(d[ysz + ysz // 4:].sum())]
            if fr:
                final.tofile(fr)
            log("rank %d: pictures %d..%d encoded (%.2f s per picture)" % (rank, b0, b0 + nb - 1, et))
        enc.close()
    for f in (fb, fr):
        if f:
            f.close()
    if multi:
        t = torch.from_numpy(rows)
        allrows = sharding.gather_frame_summaries(t.cuda() if dist.get_backend() == "nccl" else t)
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
