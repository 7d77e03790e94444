"""Multi-GPU execution: one process per GPU, units (spectrum, chain) sharded with NO data-path collective.

The path shards naturally: chains are independent given the data and spectra are independent fits.
Communication is limited to (1) one broadcast of the problem description from rank 0 (matrices + spectra, a few
MB) and (2) one gather of the results at the end -- RCCL over xGMI when the process group backend is "nccl"
(GPU tensors), gloo on CPU tensors in the tests.  Draws are a function of (seed, chain id) only, so the result is
independent of how units are distributed over ranks.
"""
import numpy as np


def shard_bounds(n_items, world, rank):
    """Contiguous block partition: ranks [0, n_items % world) get one extra item."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def make_units(n_spectra, chains):
    """unit u -> (spectrum, chain id): spectrum-major so that a block partition keeps whole spectra on one GPU."""
    spec = np.repeat(np.arange(n_spectra, dtype=np.int32), chains)
    chain = np.tile(np.arange(chains, dtype=np.int32), n_spectra)
    return spec, chain


def _dist():
    import torch.distributed as dist
    return dist


def _device_for_group(group=None):
    import torch
    dist = _dist()
    backend = dist.get_backend(group)
    return torch.device('cuda', torch.cuda.current_device()) if backend == 'nccl' else torch.device('cpu')


def broadcast_arrays(arrays, src=0, group=None):
    """Broadcast a dict of numpy arrays / scalars from `src` to every rank (one metadata object + one flat fp64
    buffer, i.e. a single large collective instead of one per array)."""
    import torch
    dist = _dist()
    rank = dist.get_rank(group)
    meta = [None]
    if rank == src:
        meta[0] = [(k, np.asarray(v).shape, str(np.asarray(v).dtype)) for k, v in arrays.items()]
    dist.broadcast_object_list(meta, src=src, group=group)
    dev = _device_for_group(group)
    total = int(sum(int(np.prod(shape)) if len(shape) else 1 for _, shape, _ in meta[0]))
    if rank == src:
        flat = np.concatenate([np.asarray(arrays[k], dtype=np.float64).ravel() for k, _, _ in meta[0]]) if total else np.zeros(0)
        buf = torch.from_numpy(flat).to(dev)
    else:
        buf = torch.empty(total, dtype=torch.float64, device=dev)
    if total:
        dist.broadcast(buf, src=src, group=group)
    flat = buf.cpu().numpy()
    out, o = {}, 0
    for k, shape, dtype in meta[0]:
        n = int(np.prod(shape)) if len(shape) else 1
        a = flat[o:o + n].reshape(shape)
        out[k] = a.astype(dtype) if dtype != 'float64' else a.copy()
        o += n
    return out


def gather_rows(local, counts, group=None):
    """All-gather row blocks of unequal length (counts[r] rows on rank r): returns the concatenation on every rank."""
    import torch
    dist = _dist()
    world = dist.get_world_size(group)
    dev = _device_for_group(group)
    local = np.ascontiguousarray(local, dtype=np.float64)
    tail = local.shape[1:]
    width = int(np.prod(tail)) if len(tail) else 1
    maxc = int(max(counts))
    pad = np.zeros((maxc, width))
    pad[:local.shape[0]] = local.reshape(local.shape[0], width)
    t = torch.from_numpy(pad).to(dev)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t, group=group)
    parts = [outs[r].cpu().numpy()[:counts[r]] for r in range(world)]
    return np.concatenate(parts, axis=0).reshape((int(sum(counts)),) + tuple(tail))


def _gpu_sample_fn(problem_kwargs, spec, chain_ids, warmup, n_draws, seed, control):
    """Default per-rank worker: build the problem in this rank's HBM and run the device-resident NUTS."""
    import ctypes as C
    from . import _lib
    from .engine import sample_units
    from .model import Problem
    kw = dict(problem_kwargs)
    blocks = kw.pop('blocks'); Z = kw.pop('Z'); freq = kw.pop('freq')
    prob = Problem(blocks, Z, freq, **kw)
    ctrl = _lib.NutsControl()
    prob._lib.bdrt_nuts_defaults(C.byref(ctrl))
    for k, v in (control or {}).items():
        setattr(ctrl, k, v)
    draws, lp, diag = sample_units(prob, len(spec), warmup, n_draws, seed, ctrl, spec=spec, chain_ids=chain_ids)
    stats = np.array([[d['n_leapfrog'], d['n_divergent'], d['n_max_treedepth'], d['stepsize'], d['mean_accept']] for d in diag])
    return draws, lp, stats


def _pack_problem(problem_kwargs):
    flat = {}
    kw = dict(problem_kwargs)
    blocks = kw.pop('blocks')
    flat['n_blocks'] = np.array(len(blocks))
    for b, blk in enumerate(blocks):
        for k in ('A', 'L0', 'L1', 'L2'):
            flat['b%d_%s' % (b, k)] = np.asarray(blk[k], dtype=np.float64)
        flat['b%d_flags' % b] = np.array([float(bool(blk.get('parallel', False))), float(bool(blk.get('nonneg', False))),
                                         float(blk.get('x_scale', 1.0))])
    for k, v in kw.items():
        if v is not None:
            flat['kw_' + k] = np.asarray(v, dtype=np.float64)
    return flat


def _unpack_problem(flat):
    nb = int(flat['n_blocks'])
    blocks = []
    for b in range(nb):
        fl = flat['b%d_flags' % b]
        blocks.append(dict(A=flat['b%d_A' % b], L0=flat['b%d_L0' % b], L1=flat['b%d_L1' % b], L2=flat['b%d_L2' % b],
                           parallel=bool(fl[0]), nonneg=bool(fl[1]), x_scale=float(fl[2])))
    kw = {'blocks': blocks}
    for k, v in flat.items():
        if k.startswith('kw_'):
            name = k[3:]
            kw[name] = v if v.ndim else (int(v) if name in ('outlier_mode',) else (bool(v) if name == 'use_x_sum' else float(v)))
    return kw


def sample_sharded(problem_kwargs, n_spectra, chains, warmup, n_draws, seed=1234, control=None, group=None,
                   sample_fn=None):
    """Sample `chains` chains for each of `n_spectra` spectra on all ranks of the process group.

    problem_kwargs (needed on rank 0 only; other ranks may pass None): dict(blocks=[...], Z=[n_spectra x 2nf], freq=...,
    **scalars) as for model.Problem.  Returns on every rank (draws [n_units, n_draws, D], lp [n_units, n_draws],
    stats [n_units, 5]) in unit order (spectrum-major), identical for any world size."""
    dist = _dist()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    flat = broadcast_arrays(_pack_problem(problem_kwargs) if rank == 0 else None, src=0, group=group)
    kw = _unpack_problem(flat)
    spec, chain = make_units(n_spectra, chains)
    bounds = [shard_bounds(n_spectra, world, r) for r in range(world)]        # whole spectra per rank
    s0, s1 = bounds[rank]
    sel = (spec >= s0) & (spec < s1)
    fn = sample_fn or _gpu_sample_fn
    if s1 > s0:
        local_kw = dict(kw)
        local_kw['Z'] = np.atleast_2d(kw['Z'])[s0:s1]                          # only this rank's spectra go to HBM
        draws, lp, stats = fn(local_kw, spec[sel] - s0, chain[sel], warmup, n_draws, seed, control)
    else:
        D = None
        draws = lp = stats = None
    # agree on D for ranks without work
    dd = [None]
    if rank == 0:
        dd[0] = int(draws.shape[2])
    dist.broadcast_object_list(dd, src=0, group=group)
    D = dd[0]
    if draws is None:
        draws, lp, stats = np.zeros((0, n_draws, D)), np.zeros((0, n_draws)), np.zeros((0, 5))
    counts = [(b[1] - b[0]) * chains for b in bounds]
    return (gather_rows(draws, counts, group), gather_rows(lp, counts, group), gather_rows(stats, counts, group))
