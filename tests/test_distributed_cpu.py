"""N > 1 path on CPU: two gloo ranks render interleaved tile shards (with the CPU oracle
standing in for the GPU kernels) and merge them with the package's single sum-reduction."""
import os
import subprocess
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {repo!r}); sys.path.insert(0, os.path.join({repo!r}, "tests"))
import importlib.util
import __graft_entry__ as ge
import oracle_binding as ob
b = ge._load_binding()
spec = importlib.util.spec_from_file_location("iile_multigpu", os.path.join({repo!r}, "pbrt-v3-iile_amd", "multigpu.py"))
mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
dist.init_process_group(backend="gloo")
scene = b.HostScene(xres=96, yres=80, spp=2)
orc = ob.Oracle()
h, w = scene.film_shape
film = torch.zeros((h, w, 4), dtype=torch.float32)
def render(rank, world):
    part, _ = orc.render(scene, threads=2, tile_rank=rank, tile_nranks=world)
    film.copy_(torch.from_numpy(part))
mg.render_sharded(render, film, dist)
if dist.get_rank() == 0:
    np.save({out!r}, film.numpy())
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_tile_sharding_matches_single_process(tmp_path, binding, oracle):
    out = str(tmp_path / "merged.npy")
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(repo=REPO, out=out))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29611", str(script)]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:]
    merged = np.load(out)
    scene = binding.HostScene(xres=96, yres=80, spp=2)
    full, _ = oracle.render(scene, threads=2)
    assert np.array_equal(merged[..., 3], full[..., 3])
    assert np.allclose(merged, full, rtol=1e-6, atol=0)
    # tiles are disjoint: away from tile borders the merge is exact
    assert np.array_equal(merged[1:15, 1:15].view(np.uint32), full[1:15, 1:15].view(np.uint32))


IISPT_WORKER = r'''
import os, sys, types, importlib
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {repo!r})
frame_mod = importlib.import_module("pbrt-v3-iile_amd.iispt_frame")
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
w, h, n_tasks, n_direct = 96, 80, 21, 5
tasks = list(frame_mod.schedule((0, 0, w, h), n_tasks, 4.0))
film = torch.zeros((h, w, 4), dtype=torch.float64)
direct = torch.zeros((h, w, 4), dtype=torch.float64)
yy, xx = np.mgrid[0:h, 0:w]
for i, (x0, y0, x1, y1, ts) in enumerate(tasks):          # the rule of IisptFrame.run_batched: task i belongs to rank i mod N
    if i % world != rank:
        continue
    v = np.float32(np.sin(0.37 * i + 0.011 * xx[y0:y1, x0:x1] + 0.007 * yy[y0:y1, x0:x1]) ** 2)
    film[y0:y1, x0:x1, :3] += torch.from_numpy(np.stack([v, v * np.float32(0.5), v * np.float32(0.25)], -1).astype(np.float64))
    film[y0:y1, x0:x1, 3] += 0.5
for p in range((n_direct * rank) // world, (n_direct * (rank + 1)) // world):   # ... and of run_direct: a contiguous block of passes
    v = np.float32(np.cos(1.3 * p + 0.02 * xx + 0.03 * yy) ** 2 * 10.0 ** (p - 2))
    direct[..., :3] += torch.from_numpy(np.stack([v, v, v], -1).astype(np.float64))
    direct[..., 3] += 1.0
me = types.SimpleNamespace(film=film, film_direct=direct, stats={{}})
frame_mod.IisptFrame.reduce_monitors(me, dist=dist)
if rank == 0:
    np.save({out!r}, np.stack([film.numpy(), direct.numpy()]))
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_iispt_monitors_add_up(tmp_path):
    """The IISPT frame's sharding rule (iispt_frame.py: task i -> rank i mod N, direct passes in contiguous blocks) and its one
    all-reduce per film monitor, on two gloo ranks with stand-in task and pass values: the reduced monitors equal the single
    process's sums bit for bit (doubles holding float values: the order of additions does not show)."""
    import importlib
    sys.path.insert(0, REPO)
    frame_mod = importlib.import_module("pbrt-v3-iile_amd.iispt_frame")
    out = str(tmp_path / "monitors.npy")
    script = tmp_path / "iispt_worker.py"
    script.write_text(IISPT_WORKER.format(repo=REPO, out=out))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29613", str(script)]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:]
    got = np.load(out)
    w, h, n_tasks, n_direct = 96, 80, 21, 5
    film, direct = np.zeros((h, w, 4)), np.zeros((h, w, 4))
    yy, xx = np.mgrid[0:h, 0:w]
    for i, (x0, y0, x1, y1, ts) in enumerate(frame_mod.schedule((0, 0, w, h), n_tasks, 4.0)):
        v = np.float32(np.sin(0.37 * i + 0.011 * xx[y0:y1, x0:x1] + 0.007 * yy[y0:y1, x0:x1]) ** 2)
        film[y0:y1, x0:x1, :3] += np.stack([v, v * np.float32(0.5), v * np.float32(0.25)], -1).astype(np.float64)
        film[y0:y1, x0:x1, 3] += 0.5
    for p_ in range(n_direct):
        v = np.float32(np.cos(1.3 * p_ + 0.02 * xx + 0.03 * yy) ** 2 * 10.0 ** (p_ - 2))
        direct[..., :3] += np.stack([v, v, v], -1).astype(np.float64)
        direct[..., 3] += 1.0
    assert (film[..., 3] >= 1.0).all() and (direct[..., 3] == n_direct).all()
    assert np.array_equal(got[0].view(np.uint64), film.view(np.uint64)) and np.array_equal(got[1].view(np.uint64), direct.view(np.uint64))
