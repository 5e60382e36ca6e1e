"""k-point sharding for one-process-per-GPU runs (pure index arithmetic).

Every k-point, plaquette and Berry string is independent, so the path shards with
no data-path collective:

* `split_list`  -- contiguous chunks of a flat k list (solve_all).
* `split_rows`  -- slabs along mesh axis 0 for solve_on_grid + berry_flux.  A slab
  owns `rows` plaquette rows and stores `rows + 1` mesh rows: its last row is the
  first row of the next slab (or the periodic image for the last slab) and is
  recomputed locally -- the kernels are deterministic, so the copy is bit-identical
  and nothing is exchanged.
* `split_strings` -- Berry strings along `dir` are sharded along another axis so
  that every string stays local.

The only communication is the final gather of eigenvalues / phases / partial flux
sums (`tbk_comm_allgather_f64`, RCCL over xGMI).
"""


def split_list(n_items, world_size, rank):
    """[begin, end) of rank's contiguous chunk; the first n%world chunks get one more."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    base, extra = divmod(int(n_items), world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def split_rows(mesh0, world_size, rank):
    """Slab of a mesh whose axis 0 has `mesh0` points (mesh0-1 plaquette rows).

    Returns (row0, nrows_stored): the slab stores global mesh rows
    [row0, row0 + nrows_stored) and owns the nrows_stored-1 plaquette rows between them."""
    begin, end = split_list(int(mesh0) - 1, world_size, rank)
    if end - begin < 1:
        raise ValueError("more ranks than plaquette rows")
    return begin, end - begin + 1


def split_strings(mesh, dir, world_size, rank):
    """Axis (!= dir) and [begin,end) range of it that this rank's strings cover."""
    axes = [d for d in range(len(mesh)) if d != dir]
    if not axes:
        raise ValueError("a 1-D array has a single string; it does not shard")
    axis = max(axes, key=lambda d: mesh[d])
    begin, end = split_list(int(mesh[axis]), world_size, rank)
    return axis, begin, end
