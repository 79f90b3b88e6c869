"""Sequence sharding across the GPUs of a node (one process per GPU, RCCL over xGMI).

Video sequences are independent units of work (the reference deletes its processor between
sequences, eval_interactive_davis.py:79-83), so the data path has NO collective: every rank runs
its own clips on its own replica of the weights.  The only exchange is a latency-bound gather of a
few hundred bytes of per-clip records at the end (``gather_records``); xGMI bandwidth is irrelevant.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Initialise torch.distributed from the torchrun environment (RANK / WORLD_SIZE / MASTER_*).
    backend defaults to 'nccl' (= RCCL on ROCm) when a GPU is visible, else 'gloo'.  Returns
    (rank, world_size, local_rank); a single-process run returns (0, 1, 0) without initialising."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # MIVOS_DIST_BACKEND=gloo: plumbing tests of the multi-rank path on a box with fewer GPUs than ranks
            backend = os.environ.get("MIVOS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            n_dev = torch.cuda.device_count()
            if backend == "nccl":
                # RCCL needs one device per rank: two ranks on one GPU end in a duplicate-device error or a hang
                if local >= n_dev:
                    raise RuntimeError(f"LOCAL_RANK {local} but only {n_dev} GPU(s) visible: the nccl (RCCL) backend needs one GPU per "
                                       f"rank (MIVOS_DIST_BACKEND=gloo runs the multi-rank plumbing on fewer GPUs)")
                torch.cuda.set_device(local)
            else:
                torch.cuda.set_device(local % n_dev)         # gloo plumbing runs: ranks may share a GPU
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def clip_cost(n_frames, n_objects):
    """Relative cost of a clip: every propagated frame costs one object-independent part (query
    encoder + decoder skip branches + launch floor) plus a per-object part (memorize + decode), SURVEY.md §8(d).
    The constant is the measured one: a least-squares fit of per-clip seconds on an MI355X (bench.py --config 4, round 3:
    2.10 ms + 0.66 ms per object and frame, residuals <= 12 %; the FLOP count alone would say 1.6)."""
    return n_frames * (1.0 + 0.32 * n_objects)


def assign_sequences(costs, world_size):
    """Longest-processing-time-first greedy partition.  costs: list of floats.  Returns a list of
    world_size lists of clip indices; deterministic, identical on every rank."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0.0] * world_size
    parts = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda j: (loads[j], j))
        parts[r].append(i)
        loads[r] += costs[i]
    return parts


def gather_records(records):
    """All ranks' per-clip records (small picklable objects) on every rank, ordered by rank."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return list(records)
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, list(records))
    return [r for part in out for r in part]


def max_over_ranks(value, device=None):
    """max of a python float over all ranks (the timed-region contract of bench.py)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def mean_over_ranks(value):
    """Mean of a python float over the ranks (logging: the reference's Integrator reduces every logged scalar, log_integrator.py:66-73)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t)
    return float(t.item()) / dist.get_world_size()


def collective_ranks(device=None):
    """How many ranks the collective backend actually reaches: an all-reduce of ones (on `device` under nccl = RCCL over xGMI,
    on the host under gloo).  Returned as dict(dist_backend, rccl_ranks | gloo_ranks) for the benchmark line, so that a reader
    sees that RCCL saw N GPUs; a single-process run returns dist_backend None."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return dict(dist_backend=None, rccl_ranks=None)
    backend = dist.get_backend()
    t = torch.ones(1, dtype=torch.float32, device=device if backend == "nccl" else "cpu")
    dist.all_reduce(t)
    n = int(round(float(t.item())))
    if n != dist.get_world_size():
        raise RuntimeError(f"all-reduce of ones over {dist.get_world_size()} ranks returned {n}")
    return dict(dist_backend=backend, rccl_ranks=n if backend == "nccl" else None, **({} if backend == "nccl" else {f"{backend}_ranks": n}))


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def average_gradients(flat):
    """Data-parallel training (model/fusion_model.py wraps FusionNet in DistributedDataParallel): the ONE collective of a training
    step - sum the flat gradient vector over the ranks (RCCL all-reduce over xGMI; 160 KB for FusionNet: latency bound) and
    divide by the world size, in place.  Under gloo (CPU plumbing tests, or GPU tensors without RCCL) the reduction runs on a
    host copy."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return flat
    if flat.is_cuda and dist.get_backend() != "nccl":
        host = flat.cpu()
        dist.all_reduce(host)
        flat.copy_(host)
    else:
        dist.all_reduce(flat)
    flat /= dist.get_world_size()
    return flat


def broadcast_parameters(flat, src=0):
    """Every rank starts from rank `src`'s parameters (what DistributedDataParallel does at construction)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return flat
    if flat.is_cuda and dist.get_backend() != "nccl":
        host = flat.cpu()
        dist.broadcast(host, src)
        flat.copy_(host)
    else:
        dist.broadcast(flat, src)
    return flat
