"""Multi-process sharding of one plane into row bands (the N>1 path): world_size-2 gloo test on CPU.

Each rank takes its band from w2xc.shard_rows / shard_view -- the same split the HIP engine uses for
its in-process multi-device path and bench.py --workload plane uses per rank -- converts the band's
view with the CPU oracle standing in for the GPU (test infrastructure), and rank 0 gathers.  The
stitched result must be bit-identical to the unsharded conversion: bands are independent, no
collective on the data path (only the gather of finished rows)."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, rand_plane


def _worker(rank, world, port, tmpdir, h, w):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as graft
    from tools import gen_model
    from oracle import oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w2xc = graft.load_package()
    layers = gen_model.synth_layers([1, 8, 8, 8, 1], 21)
    n = len(layers)
    plane = rand_plane(h, w, 5)
    ra, rb = w2xc.shard_rows(h, world, rank)
    y0, y1 = w2xc.shard_view(h, ra, rb, n)
    view = np.ascontiguousarray(plane[y0:y1])
    band = orc.Oracle(layers).convert(view, block_splitting=False)[ra - y0:rb - y0]
    sizes = [w2xc.shard_rows(h, world, r) for r in range(world)]
    rows_max = max(b - a for a, b in sizes)     # gather wants equal shapes: pad the shorter band
    padded = np.zeros((rows_max, w), np.float32)
    padded[:rb - ra] = band
    mine = torch.from_numpy(padded)
    if rank == 0:
        parts = [torch.empty((rows_max, w), dtype=torch.float32) for _ in sizes]
        dist.gather(mine, parts, dst=0)
        np.save(os.path.join(tmpdir, "stitched.npy"), torch.cat([p[:b - a] for p, (a, b) in zip(parts, sizes)]).numpy())
    else:
        dist.gather(mine, None, dst=0)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("h,w", [(40, 30), (9, 17)])
def test_two_rank_row_band_shard_matches_unsharded(oracle_built, tmp_path, h, w):
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    from tools import gen_model
    from oracle import oracle as orc
    port = 29600 + (os.getpid() % 300) + h
    mp.spawn(_worker, args=(2, port, str(tmp_path), h, w), nprocs=2, join=True)
    got = np.load(str(tmp_path / "stitched.npy"))
    want = orc.Oracle(gen_model.synth_layers([1, 8, 8, 8, 1], 21)).convert(rand_plane(h, w, 5), block_splitting=False)
    assert np.array_equal(got, want)


def test_shard_geometry(w2xc):
    for h in (1, 7, 100, 2160, 16384):
        for n in (1, 2, 3, 8):
            parts = [w2xc.shard_rows(h, n, p) for p in range(n)]
            assert parts[0][0] == 0 and parts[-1][1] == h
            assert all(parts[i][1] == parts[i + 1][0] for i in range(n - 1))
            for a, b in parts:
                y0, y1 = w2xc.shard_view(h, a, b, 7)
                assert 0 <= y0 <= a and b <= y1 <= h and (y0 == 0 or a - y0 == 7) and (y1 == h or y1 - b == 7)
