"""World-size-2 gloo test (CPU) of the multi-GPU plumbing: frame sharding covers every frame exactly
once, the only collectives are a barrier and a max-reduce, and each rank's shard processed by its own
oracle instance reproduces the single-instance result (frames are independent units for history-free
configurations)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from avisynth_sangnom2_amd import clip_format, shard, synth
from oracle.oracle import Oracle
from tests.util import oracle_cfg


def test_sharding_partitions_the_stream():
    for n in (0, 1, 7, 64, 257):
        for world in (1, 2, 3, 8):
            for mode in ("round_robin", "block"):
                seen = []
                for r in range(world):
                    fr = shard.frames_for_rank(n, r, world, mode)
                    assert all(shard.owner_of(f, n, world, mode) == r for f in fr)
                    seen += fr
                assert sorted(seen) == list(range(n))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_frames, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    clip = clip_format("Y8", 64, 32)
    ora = Oracle(oracle_cfg(clip))
    mine = shard.frames_for_rank(n_frames, rank, world)
    digest = {}
    dist.barrier()
    for f in mine:
        out = ora.process(synth.frame(clip, "noise", seed=f))
        digest[f] = int(out[0].astype(np.uint64).sum())
    dist.barrier()
    elapsed = shard.max_elapsed(0.5 + rank)  # max over ranks -> 0.5 + (world - 1)
    gathered = [None] * world
    dist.all_gather_object(gathered, digest)
    if rank == 0:
        merged = {}
        for g in gathered:
            merged.update(g)
        q.put((elapsed, merged))
    dist.destroy_process_group()


def test_two_ranks_shard_a_stream():
    n_frames, world = 9, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    elapsed, merged = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert elapsed == pytest.approx(0.5 + world - 1)
    clip = clip_format("Y8", 64, 32)
    ora = Oracle(oracle_cfg(clip))
    want = {f: int(ora.process(synth.frame(clip, "noise", seed=f))[0].astype(np.uint64).sum()) for f in range(n_frames)}
    assert merged == want


# ---- bench.py --gpus N: the launcher logic (no GPU involved) -----------------------------------------------
def _bench_module():
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("sn_bench", os.path.join(root, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_gpus_flag_spawns_one_rank_per_gpu(monkeypatch):
    """`bench.py --gpus N` without a launcher must start N ranks (torch.distributed.run, 127.0.0.1 rendezvous)
    from a process that has not touched the GPU, and pass its own arguments on unchanged."""
    import subprocess
    import sys
    bench = _bench_module()
    calls = []
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: calls.append((cmd, env)) or 0)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2", "--warmup", "1"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    (cmd, env), = calls
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "2", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert "torch.cuda" not in sys.modules or not sys.modules["torch"].cuda.is_initialized()


def test_bench_gpus_flag_must_agree_with_the_launcher(monkeypatch):
    import sys
    bench = _bench_module()
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "WORLD_SIZE=2" in str(e.value.code)
