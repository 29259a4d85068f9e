"""Multi-GPU decomposition of SamplerIntegrator::Render — Python binding of libiile_dist.so.

The reference fans its 16x16 tiles out over threads (src/core/parallel.cpp:247-299) and
merges FilmTiles under a mutex (src/core/film.cpp:135-148). Here rank r of n renders the
tiles iile_tile_owner (include/iile_scene.h) assigns to it into a full-resolution {X,Y,Z,w}
film that is zero elsewhere, and ONE sum-reduction to rank 0 merges them: RCCL's ncclReduce
through the C ABI (include/iile_dist.h, `iile_dist_film_reduce`) — the same entry point the C++
host (csrc/host/gpu_integrator.h, `iile_pbrt --gpurank r/n`) calls. Tiles are disjoint, so the sum
only ever adds a value to zeros — except for the samples whose film position is a whole number,
which also land in a neighbouring pixel that may belong to another rank (the reference's 1-pixel
FilmTile halo). No collective runs inside the render.

torch.distributed is only the launcher's rendezvous here (it carries the 128-byte RCCL id from
rank 0 to the others, and the CPU test's gloo ranks): the data path is the C ABI.
"""
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))


def _binding():
    name = "iile_binding"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(_HERE, "binding.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def tile_owner(tx, ty, nranks):
    """iile_tile_owner of include/iile_scene.h (diagonal interleave of the 16x16 tiles)."""
    return 0 if nranks <= 1 else (int(tx) + int(ty)) % int(nranks)


def create_comm(dist, device):
    """One film-merge communicator per process: rank 0 makes the RCCL id, torch.distributed (already initialised by the
    launcher, any backend) hands it to the other ranks, every rank joins through iile_dist_create."""
    import torch
    b = _binding()
    rank, world = dist.get_rank(), dist.get_world_size()
    ident = [b.Dist.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(ident, src=0, device=torch.device(device) if dist.get_backend() == "nccl" else None)
    return b.Dist(ident[0], rank, world)


def render_sharded(render_fn, film, dist=None, comm=None, stream=None):
    """render_fn(tile_rank, tile_nranks) must leave this rank's contribution in `film` (a torch tensor). After the
    call rank 0 holds the merged film. With `comm` (a binding.Dist; film on the GPU) the merge is
    iile_dist_film_reduce on `stream`; without one (CPU tensors, the gloo test) torch.distributed's reduce stands in."""
    if comm is not None and comm.size > 1:
        render_fn(comm.rank, comm.size)
        comm.film_reduce(film.data_ptr(), film.numel() // 4, 0, stream)
        return film
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        render_fn(0, 1)
        return film
    if film.is_cuda:
        raise RuntimeError("render_sharded: GPU films are merged through libiile_dist.so — pass comm=create_comm(dist, device)")
    rank, world = dist.get_rank(), dist.get_world_size()
    render_fn(rank, world)
    dist.reduce(film, dst=0, op=dist.ReduceOp.SUM)
    return film
