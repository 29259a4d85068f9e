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
