"""World-size-2 test of the multi-GPU path on CPU (gloo): env sharding + RankGather to rank 0.
The engines are oracle-backed (tests/oracle_backend.py); on the GPU box the same code runs over RCCL."""
import copy
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, what, mode, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from common import load_cfg, RES
    from oracle_backend import OracleVecEnv
    from tinycarlo_amd.distributed import RankGather, shard_range, shard_seed
    cfg, path = load_cfg("simple_layout")
    cfg = copy.deepcopy(cfg)
    cfg["camera"]["resolution"] = [64, 64]
    cfg["sim"]["observation_space_format"] = "classes"
    cfg["map"]["json_path"] = os.path.join(os.path.dirname(path), cfg["map"]["json_path"])
    lo, hi = shard_range(total, rank, world)
    n = hi - lo
    env = OracleVecEnv(cfg, num_envs=n)
    env.reset(seed=shard_seed(100, rank, total // world))
    rng = np.random.default_rng(0)
    last = None
    if mode == "step":  # the caller steps, the gather stages env.out into its two slots
        g = RankGather(env, what=what)
        for t in range(6):
            cc_all = np.stack([rng.uniform(0.3, 1, total), rng.uniform(-1, 1, total)], axis=1).astype(np.float32)
            mn_all = rng.integers(0, 4, total).astype(np.int32)
            env.step({"car_control": cc_all[lo:hi], "maneuver": mn_all[lo:hi]})
            g.step()
            if t in (2, 5):  # NOT after every step: two gathers are in flight while the env moves on
                last = g.latest()
    else:  # launch(): K steps per launch straight into the slot, gather of launch i overlapping launch i + 1
        Kl = 2
        g = RankGather(env, what=what, steps_per_launch=Kl)
        acts = []
        for t in range(6):
            cc_all = np.stack([rng.uniform(0.3, 1, total), rng.uniform(-1, 1, total)], axis=1).astype(np.float32)
            mn_all = rng.integers(0, 4, total).astype(np.int32)
            acts.append((torch.from_numpy(cc_all[lo:hi].copy()), torch.from_numpy(mn_all[lo:hi].copy())))
        seen = []
        for i in range(0, 5, Kl):  # launches of 2, 2 and a short one of 1 ... then the 6th step alone
            rows = min(Kl, 5 - i)
            g.launch(torch.stack([a[0] for a in acts[i:i + rows]]), torch.stack([a[1] for a in acts[i:i + rows]]))
        g.launch(acts[5][0], acts[5][1])  # [N, 2] / [N] form
        last = g.latest()
    if rank == 0:
        # reference: one unsharded env over all `total` envs with the same global seeds and actions
        full = OracleVecEnv(cfg, num_envs=total)
        full.reset(seed=100)
        rng = np.random.default_rng(0)
        for t in range(6):
            cc_all = np.stack([rng.uniform(0.3, 1, total), rng.uniform(-1, 1, total)], axis=1).astype(np.float32)
            mn_all = rng.integers(0, 4, total).astype(np.int32)
            full.step({"car_control": cc_all, "maneuver": mn_all})
        ok = torch.equal(last["reward"].reshape(-1), full.out["reward"])
        ok &= torch.equal(last["terminated"].reshape(-1), full.out["terminated"].bool())
        ok &= torch.equal(last["truncated"].reshape(-1), full.out["truncated"].bool())
        if what == "obs":
            ok &= torch.equal(last["obs"].reshape((total,) + tuple(full.out["obs"].shape[1:])), full.out["obs"])
        q.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["step", "launch"])
@pytest.mark.parametrize("what", ["flags", "obs"])
def test_sharded_envs_gather_to_rank0(what, mode):
    """2 ranks x 4 envs, 6 steps, rank 0 compares what it gathered with one unsharded 8-env run: post-hoc staging
    (`step`) and the double-buffered K-step launches (`launch`, incl. a short last launch and the one-step form)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 8, what, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_shard_range_partitions():
    from tinycarlo_amd.distributed import shard_range, shard_seed
    for total, world in [(8, 2), (10, 4), (4096 * 8, 8), (3, 4)]:
        spans = [shard_range(total, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert shard_seed(7, 3, 4096) == 7 + 3 * 4096
