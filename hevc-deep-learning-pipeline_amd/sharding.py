"""Work partitioning across GPUs (one process per GPU, torch.distributed over RCCL; gloo in the CPU tests).

Two partitions exist on this path:

* frames -- the default.  Every picture of the all-intra configuration is independent (IRAP I-slices, contexts re-initialised per
  slice, parameter sets re-sent), so rank r takes a contiguous block of frames and nothing crosses ranks on the data path; only the
  48-byte per-frame summaries are gathered (`gather_frame_summaries`).
* tiles of one picture -- when the cfg enables tiles (TileUniformSpacing 1, NumTileColumnsMinus1, NumTileRowsMinus1) and there are
  fewer pictures in flight than GPUs.  Tiles are independent units of the decision path (own coder state, no prediction across
  tile borders), so rank r decides tiles [r * T / R, (r + 1) * T / R) of every picture with `hevcdl_compress_tiles_dev`.  The
  in-loop filters are NOT tile-local: deblocking and SAO cross tile borders (LFCrossTileBoundaryFlag 1) and the SAO decision is one
  serial chain over the CTUs of the whole picture (TEncSampleAdaptiveOffset.cpp:836).  So there is one real exchange step: every
  rank sends the records and reconstruction rectangles of its tiles to the picture's OWNER (frame f -> rank f mod R), which then
  filters the assembled picture and writes its access unit.  That exchange is a single `all_to_all_single` of equal, padded
  splits per batch of pictures (`exchange_tiles_to_owners`): over xGMI every pair of GPUs has its own link, so the all-to-all runs
  on all links at once; a picture's payload is ~1.5 B per luma sample of reconstruction + 15 120 B per CTU of records.

Everything here is plain tensor plumbing (uint8 views); it runs unchanged on CPU tensors with gloo, which is how tests/ covers it.
"""
import numpy as np
import torch
import torch.distributed as dist

REC_BYTES = 15120                      # sizeof(hevcdl_ctu_record)


def shard_frames(n_frames, world, rank):
    """Contiguous frame ranges (better for file reads than a stride): rank r takes [r * n / world, (r + 1) * n / world) -- sizes differ by at most one and no
    rank is empty while world <= n_frames (blocks of ceil(n / world) left trailing ranks without a frame: 5 frames on 4 ranks, 9 on 8)."""
    return range(rank * n_frames // world, (rank + 1) * n_frames // world)


def max_shard(n_frames, world):
    """Frames of the largest block shard_frames deals: what per-rank buffers and gather rows are sized for."""
    return (n_frames + world - 1) // world


def shard_tiles(n_tiles, world, rank):
    """Tiles [begin, begin + count) of every picture that `rank` decides (raster order of the tile grid)."""
    begin, end = (rank * n_tiles) // world, ((rank + 1) * n_tiles) // world
    return begin, end - begin


def tile_grid(width, height, tiles):
    """Tiles ((columns, rows) uniformly spaced, or explicit sizes: hevcdl_amd.tile_layout) -> list of (cx0, cy0, cx1, cy1) in CTUs,
    raster order of tiles."""
    from . import tile_layout
    c, r, _, cb, rb = tile_layout(tiles, width, height)
    return [(cb[tc], rb[tr], cb[tc + 1], rb[tr + 1]) for tr in range(r) for tc in range(c)]


def _planes(frames, width, height, bps=1):
    """[F, w*h*3/2*bps] uint8 -> byte views (Y [F,h,w*bps], U [F,h/2,w/2*bps], V [F,h/2,w/2*bps]); bps = bytes per sample."""
    f = frames.shape[0]
    ysz, csz = width * height * bps, width * height // 4 * bps
    return (frames[:, :ysz].view(f, height, width * bps), frames[:, ysz:ysz + csz].view(f, height // 2, width // 2 * bps),
            frames[:, ysz + csz:].view(f, height // 2, width // 2 * bps))


def tile_payload_bytes(width, height, rect, bps=1):
    cx0, cy0, cx1, cy1 = rect
    w, h = min(cx1 * 64, width) - cx0 * 64, min(cy1 * 64, height) - cy0 * 64
    return w * h * 3 // 2 * bps + (cx1 - cx0) * (cy1 - cy0) * REC_BYTES


def pack_tile(recon, records, width, height, rect, out, bps=1):
    """Reconstruction rectangle (Y, U, V) + CTU records of one tile of every frame -> out [F, >= payload] uint8.  recon is the byte
    view of the pictures (bps bytes per sample: 2 for 10-bit)."""
    cx0, cy0, cx1, cy1 = rect
    ctus_x = (width + 63) // 64
    x0, y0, x1, y1 = cx0 * 64, cy0 * 64, min(cx1 * 64, width), min(cy1 * 64, height)
    f, o = recon.shape[0], 0
    for p, sh in zip(_planes(recon, width, height, bps), (0, 1, 1)):
        blk = p[:, y0 >> sh:y1 >> sh, (x0 >> sh) * bps:(x1 >> sh) * bps].reshape(f, -1)
        out[:, o:o + blk.shape[1]] = blk
        o += blk.shape[1]
    recs = records.view(f, -1, REC_BYTES).view(f, (height + 63) // 64, ctus_x, REC_BYTES)[:, cy0:cy1, cx0:cx1].reshape(f, -1)
    out[:, o:o + recs.shape[1]] = recs
    return o + recs.shape[1]


def unpack_tile(buf, recon, records, width, height, rect, bps=1):
    """Inverse of pack_tile: buf [F, >= payload] -> the tile's rectangle of recon [F, w*h*3/2] and records [F, ctus*REC_BYTES]."""
    cx0, cy0, cx1, cy1 = rect
    ctus_x = (width + 63) // 64
    x0, y0, x1, y1 = cx0 * 64, cy0 * 64, min(cx1 * 64, width), min(cy1 * 64, height)
    f, o = recon.shape[0], 0
    for p, sh in zip(_planes(recon, width, height, bps), (0, 1, 1)):
        hh, ww = (y1 - y0) >> sh, ((x1 - x0) >> sh) * bps
        p[:, y0 >> sh:y1 >> sh, (x0 >> sh) * bps:(x1 >> sh) * bps] = buf[:, o:o + hh * ww].view(f, hh, ww)
        o += hh * ww
    n = (cx1 - cx0) * (cy1 - cy0) * REC_BYTES
    records.view(f, (height + 63) // 64, ctus_x, REC_BYTES)[:, cy0:cy1, cx0:cx1] = buf[:, o:o + n].view(f, cy1 - cy0, cx1 - cx0, REC_BYTES)
    return o + n


def exchange_tiles_to_owners(recon, records, width, height, tiles, group=None, bps=1):
    """Tile-sharded decisions -> whole pictures at their owners.

    recon [F, w*h*3/2*bps] / records [F, ctus*REC_BYTES] (uint8 byte views, this rank's tiles filled in, same F on every rank, F a
    multiple of the world size; bps = bytes per sample).  Frame f is owned by rank f % world.  Returns (owned frame indices, recon [F/world, ...], records [F/world, ...])
    with every tile of the owned pictures in place.  One all_to_all_single of equal padded splits."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    f = recon.shape[0]
    if f % world:
        raise ValueError("the number of pictures in a batch must be a multiple of the world size")
    grid = tile_grid(width, height, tiles)
    mine = shard_tiles(len(grid), world, rank)
    # split s of the send buffer = this rank's tiles of the pictures owned by rank s, padded to the largest per-rank payload
    per_rank = [sum(tile_payload_bytes(width, height, grid[t], bps) for t in range(*_span(shard_tiles(len(grid), world, r)))) for r in range(world)]
    pad, fo = max(per_rank), f // world
    send = torch.zeros((world, fo, pad), dtype=torch.uint8, device=recon.device)
    for s in range(world):
        idx = torch.arange(s, f, world, device=recon.device)
        rsub, csub = recon[idx], records[idx]
        o = 0
        for t in range(*_span(mine)):
            o += pack_tile(rsub, csub, width, height, grid[t], send[s, :, o:], bps)
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv.view(-1), send.view(-1), group=group)
    owned = list(range(rank, f, world))
    out_recon = recon[torch.tensor(owned, device=recon.device)].clone()
    out_records = records[torch.tensor(owned, device=recon.device)].clone()
    for r in range(world):
        o = 0
        for t in range(*_span(shard_tiles(len(grid), world, r))):
            o += unpack_tile(recv[r, :, o:], out_recon, out_records, width, height, grid[t], bps)
    return owned, out_recon, out_records


def _span(begin_count):
    return begin_count[0], begin_count[0] + begin_count[1]


def gather_frame_summaries(summaries, group=None):
    """Per-frame summary rows (int64 [n, k]: poc, bits, sse...; equal n on every rank, pad with poc -1) -> rank 0 gets them sorted by
    POC (others None).  The only collective of the frame-sharded path; latency-bound (48 B per picture)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    out = [torch.zeros_like(summaries) for _ in range(world)] if rank == 0 else None
    if summaries.is_cuda:             # RCCL has no gather primitive: all_gather of a few hundred bytes
        out = [torch.zeros_like(summaries) for _ in range(world)]
        dist.all_gather(out, summaries, group=group)
    else:
        dist.gather(summaries, out, dst=0, group=group)
    if rank != 0:
        return None
    allrows = torch.cat(out).cpu().numpy()
    allrows = allrows[allrows[:, 0] >= 0]
    return allrows[np.argsort(allrows[:, 0], kind="stable")]
