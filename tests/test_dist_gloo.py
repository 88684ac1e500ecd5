"""world_size-2 gloo test of the multi-GPU host logic: shard -> ONE all-gather of the packed banks ->
this rank's similarity row block.  The HIP similarity kernel needs a GPU, so the checker matmul is
injected; what is under test is sharding, packing, gather order and row-block placement."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cacophony_amd import dist as cdist


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        per = 6
        g = torch.Generator().manual_seed(123)
        a_all = torch.nn.functional.normalize(torch.randn(world * per, 32, generator=g), dim=1)
        t_all = torch.nn.functional.normalize(torch.randn(world * per, 32, generator=g), dim=1)
        lo, hi = cdist.shard_range(world * per, rank, world)
        assert hi - lo == per
        ga, gt = cdist.gather_embedding_banks(a_all[lo:hi], t_all[lo:hi])
        assert torch.equal(ga, a_all) and torch.equal(gt, t_all)
        # the packed form the towers fill directly: one collective, banks come back as strided views in rank order
        bank = torch.stack([a_all[lo:hi], t_all[lo:hi]], 1).contiguous()
        allb = cdist.gather_packed(bank, check_sizes=True)
        assert allb.shape == (world * per, 2, 32) and torch.equal(allb[:, 0], a_all) and torch.equal(allb[:, 1], t_all)
        # unequal shards (what shard_range hands out when the global count does not divide) are refused, not hung on
        n_bad = per + (1 if rank == 0 else 0)
        try:
            cdist.gather_embedding_banks(torch.zeros(n_bad, 32), torch.zeros(n_bad, 32))
            raised = False
        except ValueError:
            raised = True
        assert raised
        block = cdist.sharded_similarity(a_all[lo:hi], t_all[lo:hi], scale=2.0,
                                         similarity_fn=lambda a, t, s: s * a @ t.T)
        ref = 2.0 * a_all @ t_all.T
        assert block.shape == (per, world * per)
        assert torch.allclose(block, ref[lo:hi], atol=1e-6)
        np.save(os.path.join(out_dir, f"block{rank}.npy"), block.numpy())
    finally:
        dist.destroy_process_group()


def test_two_rank_gather_and_row_blocks(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    blocks = [np.load(tmp_path / f"block{r}.npy") for r in range(world)]
    full = np.concatenate(blocks, 0)
    assert full.shape == (12, 12)


def _worker8(rank, world, port, n_global):
    """World size 8 (the target node): shard_range's spans -> packed banks -> ONE gather.  2048 rows divide (256 each, the
    configs[3] shape); 2049 do not - rank 0 gets 257 rows - and check_sizes must refuse that on every rank."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(7)
        a_all = torch.randn(n_global, 16, generator=g)
        t_all = torch.randn(n_global, 16, generator=g)
        lo, hi = cdist.shard_range(n_global, rank, world)
        bank = torch.stack([a_all[lo:hi], t_all[lo:hi]], 1).contiguous()
        if n_global % world:
            assert hi - lo == n_global // world + (1 if rank < n_global % world else 0)
            try:
                cdist.gather_packed(bank, check_sizes=True)
                raised = False
            except ValueError as e:
                raised = "pad the shards to one size" in str(e)
            assert raised, f"rank {rank}: {hi - lo} rows of {n_global} over {world} ranks were not refused"
        else:
            allb = cdist.gather_packed(bank, check_sizes=True)
            assert allb.shape == (n_global, 2, 16) and torch.equal(allb[:, 0], a_all) and torch.equal(allb[:, 1], t_all)
            block = cdist.sharded_similarity(a_all[lo:hi], t_all[lo:hi], similarity_fn=lambda a, t, s: s * a @ t.T)
            assert block.shape == (n_global // world, n_global)
            assert torch.allclose(block, (a_all @ t_all.T)[lo:hi], atol=1e-5)
    finally:
        dist.destroy_process_group()


def test_eight_rank_gather_2048_rows():
    mp.spawn(_worker8, args=(8, _free_port(), 2048), nprocs=8, join=True)


def test_eight_rank_remainder_shards_are_refused():
    """2049 rows over 8 ranks: shard_range's remainder branch gives rank 0 one row more; every rank must raise."""
    mp.spawn(_worker8, args=(8, _free_port(), 2049), nprocs=8, join=True)


def test_shard_range_remainder_goes_to_the_first_ranks():
    spans = [cdist.shard_range(2049, r, 8) for r in range(8)]
    assert spans[0] == (0, 257) and spans[1] == (257, 513) and spans[-1] == (1793, 2049)
    assert [hi - lo for lo, hi in spans] == [257] + [256] * 7
    spans = [cdist.shard_range(2055, r, 8) for r in range(8)]
    assert [hi - lo for lo, hi in spans] == [257] * 7 + [256] and spans[-1][1] == 2055


def test_shard_range_covers_everything():
    for n, w in ((2048, 8), (10, 3), (7, 8), (256, 1)):
        spans = [cdist.shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))


def test_single_process_is_identity():
    a, t = torch.randn(4, 8), torch.randn(4, 8)
    ga, gt = cdist.gather_embedding_banks(a, t)
    assert torch.equal(ga, a) and torch.equal(gt, t)
