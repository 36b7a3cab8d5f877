"""N > 1 path on CPU: two gloo ranks shard an ensemble, each produces per-year
sufficient statistics for its block, one all-reduce merges them; the result must
equal the single-process statistics.  (On the GPU box the same functions run with
backend "nccl" = RCCL; the member block of a rank comes from shard_range.)"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, SCENARIO


def _worker(rank, world, port, n_total, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hector_amd import ensemble
    from hector_amd.distributed import shard_range, allreduce_stats, stats_numpy
    import oracle_binding
    off, cnt = shard_range(n_total, rank, world)
    S, q10 = ensemble.ecs_q10(cnt, offset=off)
    orc = oracle_binding.Oracle(SCENARIO)
    co2, tg, err = orc.run_ecs_q10(S, q10, 1850)
    k = 1850 - 1745 + 1
    st = torch.from_numpy(stats_numpy(tg[:, :k].T.copy()))
    allreduce_stats(st, dist)
    if rank == 0:
        q.put(st.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_stat_reduction_equals_single_process(oracle):
    from hector_amd import ensemble
    from hector_amd.distributed import stats_numpy, finalize
    n_total, world = 11, 2  # ragged on purpose
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs: p.start()
    got = q.get(timeout=120)
    for p in procs: p.join(timeout=60)
    S, q10 = ensemble.ecs_q10(n_total)
    co2, tg, err = oracle.run_ecs_q10(S, q10, 1850)
    k = 1850 - 1745 + 1
    ref = stats_numpy(tg[:, :k].T.copy())
    assert np.array_equal(got[:, 0], ref[:, 0]) and np.array_equal(got[:, 3:], ref[:, 3:])
    assert np.allclose(got[:, 1:3], ref[:, 1:3], rtol=1e-13, atol=1e-13)
    mean, std, mn, mx = finalize(got)
    assert np.all(mn <= mean + 1e-12) and np.all(mean <= mx + 1e-12)
