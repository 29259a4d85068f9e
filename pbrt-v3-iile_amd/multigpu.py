"""Multi-GPU decomposition of SamplerIntegrator::Render.

The reference fans its 16x16 tiles out over threads (src/core/parallel.cpp:247-299) and
merges FilmTiles under a mutex (src/core/film.cpp:135-148). Here rank r of n renders the
tiles with index % n == r into a full-resolution {X,Y,Z,w} film that is zero elsewhere,
and ONE sum-reduction to rank 0 (RCCL over xGMI when the tensors live on GPUs) merges
them. Tiles are disjoint, so the sum only ever adds a value to zeros — except for the
k = 0 samples with a zero fractional film offset, which also land in a neighbouring
pixel that may belong to another rank (the reference's 1-pixel FilmTile halo).
No collective runs inside the render.
"""


def render_sharded(render_fn, film, dist=None):
    """render_fn(tile_rank, tile_nranks) must leave this rank's contribution in `film`
    (a torch tensor, any device). Returns film; after the call rank 0 holds the merged film."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        render_fn(0, 1)
        return film
    rank, world = dist.get_rank(), dist.get_world_size()
    render_fn(rank, world)
    dist.reduce(film, dst=0, op=dist.ReduceOp.SUM)
    return film
