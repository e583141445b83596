"""One-rank RCCL smoke on the GPU box: the only way to run the frame-stream -> caller-stream -> RCCL-stream hand-off for
real on a one-GPU lease.  init_process_group("nccl", world_size=1) (nccl IS RCCL on ROCm), RankGather(what="obs") over
the real HIP env, three launch()es; the rows rank 0 gathered must be the env's own rollout, which in turn is the
rollout of a twin env stepped without any collective.  No scaling figure is claimed (the driver measures
the multi-GPU curve)."""
import os
import socket

import pytest

from test_gpu_bench_shapes import bench_actions
from test_gpu_parity import make_env

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_rank_gather_over_rccl_with_one_rank():
    import torch.distributed as dist
    from tinycarlo_amd.distributed import RankGather
    assert not dist.is_initialized()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", world_size=1, rank=0, device_id=dev)
    try:
        n, K = 256, 12  # streamed calls: simulate launch on the caller's stream, frames on the internal one, then the gather on RCCL's stream
        env = make_env("simple_layout", "r64", "classes", n, autoreset=True)
        twin = make_env("simple_layout", "r64", "classes", n, autoreset=True)
        env.reset(seed=11)
        twin.reset(seed=11)
        g = RankGather(env, what="obs", steps_per_launch=K)
        ref = twin.alloc_rollout(K, keys=("obs", "reward", "terminated", "truncated"))
        for it in range(3):
            cc, man = bench_actions(n, K, seed=100 + it)
            g.launch(cc, man)                       # step_multi into slot it & 1 + async gather of that slot
            twin.step_multi(cc, man, rollout=ref)
            got = g.latest()                        # waits for the gather
            torch.cuda.synchronize()
            assert got["reward"].shape == (1, K, n) and got["obs"].shape[:3] == (1, K, n)
            assert torch.equal(got["obs"][0], ref["obs"]), ("gathered frames differ", it)
            assert torch.equal(got["reward"][0].view(torch.int64), ref["reward"].view(torch.int64)), it
            assert torch.equal(got["terminated"][0], ref["terminated"].bool()) and torch.equal(got["truncated"][0], ref["truncated"].bool())
        for k in env.state:
            assert torch.equal(env.state[k], twin.state[k]), k
        env.close()
        twin.close()
    finally:
        dist.destroy_process_group()
