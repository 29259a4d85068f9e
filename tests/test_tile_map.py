"""The tile -> rank map of the multi-GPU path (iile_tile_owner, include/iile_scene.h) on CPU."""
import importlib.util
import os
import subprocess
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _mg():
    spec = importlib.util.spec_from_file_location("iile_multigpu", os.path.join(REPO, "pbrt-v3-iile_amd", "multigpu.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    return mg


def test_python_map_is_the_header_map(oracle):
    """multigpu.tile_owner restates the static inline of iile_scene.h; the oracle library exports the header's own."""
    mg = _mg()
    for n in (1, 2, 3, 4, 5, 8, 16):
        for tx in range(0, 130, 7):
            for ty in range(0, 70, 5):
                assert mg.tile_owner(tx, ty, n) == oracle.tile_owner(tx, ty, n)


def test_map_properties_on_the_1080p_grid(oracle):
    """1080p has 120 x 68 tiles. For every rank count: a partition; equal shares (to one tile per tile row); every run
    of n consecutive tiles along a row or a column holds all n ranks — no rank owns a stripe of the image."""
    ntx, nty = 120, 68
    for n in (2, 3, 4, 8):
        owner = np.array([[oracle.tile_owner(tx, ty, n) for tx in range(ntx)] for ty in range(nty)])
        counts = np.bincount(owner.ravel(), minlength=n)
        assert counts.sum() == ntx * nty and counts.min() >= 0
        assert counts.max() - counts.min() <= nty
        for ty in range(nty):
            for tx in range(ntx - n + 1):
                assert len(set(owner[ty, tx:tx + n])) == n
        for tx in range(ntx):
            for ty in range(nty - n + 1):
                assert len(set(owner[ty:ty + n, tx])) == n
        # the linear interleave this replaced gave whole columns to one rank whenever n divides 120
        assert not any((owner[:, tx] == owner[0, tx]).all() for tx in range(ntx))


WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {repo!r}); sys.path.insert(0, os.path.join({repo!r}, "tests"))
import oracle_binding as ob
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
orc = ob.Oracle()
ntx, nty = 120, 68
mine = torch.tensor([[1 if orc.tile_owner(tx, ty, world) == rank else 0 for tx in range(ntx)] for ty in range(nty)], dtype=torch.int32)
total = mine.clone()
dist.all_reduce(total)
counts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
dist.all_gather(counts, mine.sum().reshape(1).to(torch.int64))
if rank == 0:
    np.save({out!r}, np.array([int(total.min()), int(total.max())] + [int(c) for c in counts]))
dist.barrier()
dist.destroy_process_group()
'''


def test_two_ranks_partition_the_tile_grid(tmp_path, oracle):
    """World size 2 over gloo: each rank evaluates its own share of the 1080p tile grid; together every tile exactly once."""
    out = str(tmp_path / "map.npy")
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(repo=REPO, out=out))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29613", str(script)]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:]
    r = np.load(out)
    assert r[0] == 1 and r[1] == 1
    assert r[2] + r[3] == 120 * 68 and abs(int(r[2]) - int(r[3])) <= 68
