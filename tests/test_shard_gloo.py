"""world_size-2 gloo test of the multi-GPU plumbing (sequence sharding + record gather), on CPU."""
import os
import subprocess
import sys
import textwrap

from mivos_amd.shard import assign_sequences, clip_cost

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_assignment_is_a_balanced_partition():
    costs = [clip_cost(t, k) for t, k in [(100, 1), (20, 5), (70, 3), (35, 2), (180, 1), (25, 4), (60, 2)]]
    parts = assign_sequences(costs, 3)
    assert sorted(i for p in parts for i in p) == list(range(len(costs)))
    loads = [sum(costs[i] for i in p) for p in parts]
    assert max(loads) <= 1.34 * (sum(costs) / 3)            # LPT bound 4/3 - 1/(3m)
    assert assign_sequences(costs, 3) == parts              # deterministic
    assert assign_sequences(costs[:1], 4) == [[0], [], [], []]


def test_two_rank_gloo_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        from mivos_amd import shard
        rank, world, local = shard.init_distributed(backend="gloo")
        costs = [shard.clip_cost(t, k) for t, k in [(30, 1), (20, 5), (70, 3), (35, 2), (10, 1)]]
        mine = shard.assign_sequences(costs, world)[rank]
        recs = [dict(clip=i, rank=rank, frames=int(costs[i])) for i in mine]
        shard.barrier()
        allrecs = shard.gather_records(recs)
        mx = shard.max_over_ranks(1.0 + rank)
        if rank == 0:
            assert sorted(r["clip"] for r in allrecs) == list(range(5)), allrecs
            assert {{r["rank"] for r in allrecs}} == {{0, 1}}
            assert mx == 2.0
            print("GATHER_OK", len(allrecs))
    """))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29617")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29617", str(script)],
                         capture_output=True, text=True, env=env, timeout=240)
    assert "GATHER_OK 5" in out.stdout, out.stdout + out.stderr


def test_two_rank_gloo_training_collectives(tmp_path):
    """The collectives of the data-parallel FusionNet training step (mivos_amd/model/fusion_model.py; reference: DistributedDataParallel in
    model/fusion_model.py:23-25): rank 0's parameters are broadcast, and the flat 39 905-element gradient is all-reduced and
    averaged; then both ranks apply the same update (here: the CPU training oracle's Adam) and stay bit-identical."""
    script = tmp_path / "worker.py"
    script.write_text(textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        import torch
        from mivos_amd import shard
        rank, world, local = shard.init_distributed(backend="gloo")
        n = 39905                                                      # FusionNet's parameter count
        g = torch.Generator().manual_seed(100 + rank)
        flat = torch.randn(n, generator=g)                             # different initialisation per rank ...
        shard.broadcast_parameters(flat, 0)                            # ... until rank 0's is broadcast
        ref0 = torch.randn(n, generator=torch.Generator().manual_seed(100))
        assert torch.equal(flat, ref0)
        grad = torch.randn(n, generator=torch.Generator().manual_seed(200 + rank))
        mean = (torch.randn(n, generator=torch.Generator().manual_seed(200)) + torch.randn(n, generator=torch.Generator().manual_seed(201))) / 2
        shard.average_gradients(grad)
        assert torch.allclose(grad, mean, rtol=0, atol=1e-7)
        p = flat.clone().requires_grad_(True)
        opt = torch.optim.Adam([p], lr=1e-4, weight_decay=1e-7)
        p.grad = grad.clone()
        opt.step()
        rec = shard.gather_records([dict(rank=rank, checksum=float(p.detach().double().sum()), first=float(p[0]))])
        if rank == 0:
            assert len(rec) == 2 and rec[0]["checksum"] == rec[1]["checksum"] and rec[0]["first"] == rec[1]["first"], rec
            print("TRAIN_COLLECTIVES_OK")
    """))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29619")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29619", str(script)],
                         capture_output=True, text=True, env=env, timeout=240)
    assert "TRAIN_COLLECTIVES_OK" in out.stdout, out.stdout + out.stderr
