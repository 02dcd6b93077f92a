"""CPU, world_size 2 over gloo: the N>1 path of the bench (frame sharding with no data-path collective, gather of the
per-frame summary records to rank 0) -- exercised with the oracle standing in for the per-rank work."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as tmp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


sys.path.insert(0, ROOT)
from hevcdl_amd.sharding import max_shard, shard_frames        # the product's own dealing, not a copy of it  # noqa: E402


def _worker(rank, world, port, n_frames, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_tools
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w, h, qp = 64, 64, 32
    yuv = ref_tools.synth_yuv(w, h, n_frames, seed=77)
    labels = ref_tools.make_labels(w, h, n_frames, "rand", 78)
    mine = list(shard_frames(n_frames, world, rank))
    recs, recon, stats = ref_tools.run_oracle(yuv[mine], w, h, qp, labels[mine])
    rec = torch.full((max_shard(n_frames, world), 5), -1, dtype=torch.int64)     # poc, bits, sseY, sseU, sseV
    for i, f in enumerate(mine):
        rec[i] = torch.tensor([f, int(stats["est_bits"][i])] + [int(v) for v in stats["sse"][i]])
    out = [torch.zeros_like(rec) for _ in range(world)] if rank == 0 else None
    dist.gather(rec, out, dst=0)
    if rank == 0:
        allrec = torch.cat(out)
        allrec = allrec[allrec[:, 0] >= 0].numpy()              # padding rows of the shorter blocks dropped
        _, _, full = ref_tools.run_oracle(yuv, w, h, qp, labels)
        q.put((allrec.tolist(), [[i, int(full["est_bits"][i])] + [int(v) for v in full["sse"][i]] for i in range(n_frames)]))
    dist.barrier()
    dist.destroy_process_group()


def test_frame_sharding_gather_world2(oracle_built):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = tmp.get_context("spawn")
    q = ctx.Queue()
    n_frames = 5                             # uneven: blocks of 2 and 3 frames
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, want = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == want                       # frame-sharded == single-process, in POC order


def test_shard_ranges_cover_every_frame_once():
    for n in (1, 5, 7, 9, 75, 600):
        for world in (1, 2, 4, 8):
            blocks = [shard_frames(n, world, r) for r in range(world)]
            assert [f for b in blocks for f in b] == list(range(n))
            assert max(len(b) for b in blocks) == max_shard(n, world)
            if world <= n:                   # no rank without a frame (5 frames on 4 ranks, 9 on 8: blocks of ceil(n / world) left the last rank empty)
                assert min(len(b) for b in blocks) >= 1 and max(len(b) for b in blocks) - min(len(b) for b in blocks) <= 1


def _tile_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_tools
    import hevcdl_amd.sharding as sh
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w, h, qp, nf, tiles = 520, 136, 32, 2, (2, 2)            # ragged right / bottom edge, uneven tile columns (4 + 5 CTUs)
    yuv = ref_tools.synth_yuv(w, h, nf, seed=91)
    labels = ref_tools.make_labels(w, h, nf, "rand", 92)
    recs, recon, _ = ref_tools.run_oracle(yuv, w, h, qp, labels, tiles=tiles)          # stands in for the per-rank GPU work ...
    full_recon = torch.from_numpy(recon.reshape(nf, -1).copy())
    full_recs = torch.from_numpy(np.frombuffer(recs.tobytes(), np.uint8).reshape(nf, -1).copy())
    grid = sh.tile_grid(w, h, tiles)
    begin, count = sh.shard_tiles(len(grid), world, rank)
    part_recon, part_recs = torch.full_like(full_recon, 0xAA), torch.full_like(full_recs, 0xAA)   # ... of which only this rank's tiles exist here
    scratch = torch.zeros((nf, max(sh.tile_payload_bytes(w, h, g) for g in grid)), dtype=torch.uint8)
    for t in range(begin, begin + count):
        n = sh.pack_tile(full_recon, full_recs, w, h, grid[t], scratch)
        assert n == sh.tile_payload_bytes(w, h, grid[t])
        sh.unpack_tile(scratch, part_recon, part_recs, w, h, grid[t])
    owned, got_recon, got_recs = sh.exchange_tiles_to_owners(part_recon, part_recs, w, h, tiles)
    ok = owned == list(range(rank, nf, world)) and torch.equal(got_recon, full_recon[owned]) and torch.equal(got_recs, full_recs[owned])
    rows = torch.tensor([[f, 100 + f] for f in owned] + [[-1, 0]], dtype=torch.int64)
    summ = sh.gather_frame_summaries(rows)
    if rank == 0:
        ok = ok and summ.tolist() == [[f, 100 + f] for f in range(nf)]
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_tile_sharding_exchange_world2(oracle_built):
    """Tiles of one picture decided on different ranks, then assembled at the picture's owner for the in-loop filters: the
    all-to-all of padded tile payloads (reconstruction rectangles + CTU records) reproduces the single-process picture."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = tmp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_tile_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


def test_tile_shards_cover_every_tile_once():
    import hevcdl_amd.sharding as sh
    for n in (1, 2, 6, 8, 20):
        for world in (1, 2, 4, 8):
            seen = [t for r in range(world) for t in range(sh.shard_tiles(n, world, r)[0], sum(sh.shard_tiles(n, world, r)))]
            assert seen == list(range(n))
    assert sh.tile_grid(7680, 4320, (4, 2))[5] == (30, 34, 60, 68) and len(sh.tile_grid(520, 136, (2, 2))) == 4
