"""N > 1 host logic on CPU: band ownership, packed tiles, one gather, reassembly (gloo, world_size 2 and 3)."""
import os
import socket

import numpy as np
import pytest

import raytracer_amd as rt


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, h, w, band_rows, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = np.random.default_rng(5).integers(0, 256, (h, w, 3), dtype=np.uint8)   # same "frame" on every rank
    nb_max = (((h + band_rows - 1) // band_rows) + world - 1) // world
    local = np.zeros((nb_max * band_rows, w, 3), np.uint8)
    rows = rt.band_rows_of(h, band_rows, rank, world)
    local[:len(rows)] = full[rows]            # what rt_render_bands leaves in the packed buffer
    t = torch.from_numpy(local)
    gathered = [torch.zeros_like(t) for _ in range(world)] if rank == 0 else None
    dist.gather(t, gathered, dst=0)
    if rank == 0:
        img = rt.assemble_bands([g.numpy() for g in gathered], h, band_rows)
        q.put(bool(np.array_equal(img, full)))
    dist.barrier()
    dist.destroy_process_group()


def _render_worker(rank, world, port, name, w, h, band_rows, q):
    """Each rank RENDERS its own bands (the CPU checker stands in for the GPU here: the same rows, band by band, into the
    packed buffer rt_render_bands fills) and rank 0 rebuilds the frame from one gather."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch
    import torch.distributed as dist
    from oracle_api import OracleScene
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = OracleScene("scenes/%s.json" % name, w, h, 3)
    nb_max = (((h + band_rows - 1) // band_rows) + world - 1) // world
    local = np.zeros((nb_max * band_rows, w, 3), np.uint8)
    ids = np.full((nb_max * band_rows, w), -7, np.int32)
    bands = (h + band_rows - 1) // band_rows
    k = 0
    for b in range(rank, bands, world):          # band k of this rank starts at row k * band_rows of the packed buffer
        r0, r1 = b * band_rows, min((b + 1) * band_rows, h)
        rgb_b, ids_b, _, _ = o.render(r0, r1)
        local[k * band_rows:k * band_rows + (r1 - r0)] = rgb_b[r0:r1]
        ids[k * band_rows:k * band_rows + (r1 - r0)] = ids_b[r0:r1]
        k += 1
    t, ti = torch.from_numpy(local), torch.from_numpy(ids)
    g = [torch.zeros_like(t) for _ in range(world)] if rank == 0 else None
    gi = [torch.zeros_like(ti) for _ in range(world)] if rank == 0 else None
    dist.gather(t, g, dst=0)
    dist.gather(ti, gi, dst=0)
    if rank == 0:
        want, want_ids, _, _ = o.render()
        img = rt.assemble_bands([x.numpy() for x in g], h, band_rows)
        img_ids = rt.assemble_bands([x.numpy() for x in gi], h, band_rows)
        q.put(bool(np.array_equal(img, want)) and bool(np.array_equal(img_ids, want_ids)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,name,w,h", [(2, "lambertBPHSpheres", 48, 40), (3, "meshMedBunny", 40, 50), (2, "sphere", 9, 16)])
def test_rendered_bands_of_every_rank_rebuild_the_frame(world, name, w, h):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_render_worker, args=(r, world, port, name, w, h, 16, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok = q.get(timeout=200)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok


@pytest.mark.parametrize("world,h", [(2, 2160 // 8), (3, 150), (2, 16), (3, 7)])
def test_gather_of_packed_bands_rebuilds_the_frame(world, h):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, h, 64, 16, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok


def test_band_ownership_covers_every_row_once():
    for h in (1, 15, 16, 17, 150, 2160):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                seen += rt.band_rows_of(h, 16, r, world)
            assert sorted(seen) == list(range(h))
