#!/usr/bin/env python3
"""bench.py -- throughput of the SangNom2 hot path on MI355X (contract: see the task statement).

A "step" is one pass of the hot path (frame assembly + the three stages, i.e. everything
SangNom2::GetFrame does, /root/reference/src/SangNom2.cpp:332-397) over one batch of
device-resident synthetic frames.  Default workload = the configuration BASELINE.json's metric is
quoted on: 2160p Y8, order=1, aa=48.  One process per GPU; frames shard across ranks with no
data-path collective (weak scaling: every rank processes its own batch); torch.distributed (RCCL)
is used only for the start/stop barrier and the max-over-ranks of the elapsed time.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process touches no GPU; it starts
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` as a child (one rank per GPU),
relays rank 0's JSON line and exits with the child's code.  Under a launcher (WORLD_SIZE set) `--gpus` must
agree with it.  `n_gpus` in the JSON line is the number of ranks that actually joined the process group.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (pixel_type, width, height, filter kwargs)
    "2160p-Y8": ("Y8", 3840, 2160, dict(order=1, aa=48)),
    "1080p-Y8": ("Y8", 1920, 1080, dict(order=1, aa=48)),
    "4320p-Y8": ("Y8", 7680, 4320, dict(order=1, aa=48)),
    "2160p-YUV420P8": ("YUV420P8", 3840, 2160, dict(order=1, aa=48, aac=48)),
    "2160p-YUV420P8-isolated": ("YUV420P8", 3840, 2160, dict(order=1, aa=48, aac=48, isolated_planes=True)),
    "2160p-YUV422P8": ("YUV422P8", 3840, 2160, dict(order=1, aa=48, aac=48)),
    "1080p-YUV420P8": ("YUV420P8", 1920, 1080, dict(order=1, aa=48, aac=48)),
    "480p-YUV420P8": ("YUV420P8", 720, 480, dict(order=1, aa=48, aac=48)),  # width % 32 != 0: history-carrying
    "480p-YUV420P8-fresh": ("YUV420P8", 720, 480, dict(order=1, aa=48, aac=48, fresh_pool=True)),
    "480p-YUV420P16": ("YUV420P16", 720, 480, dict(order=1, aa=48, aac=48)),  # the 16-bit and float chains
    "480p-YUV420PS": ("YUV420PS", 720, 480, dict(order=1, aa=48, aac=48)),
    "2160p-turned-Y8-fresh": ("Y8", 2160, 3840, dict(order=1, aa=48, fresh_pool=True)),
    "2160p-Y16": ("Y16", 3840, 2160, dict(order=1, aa=48)),
    "2160p-YUV420P16": ("YUV420P16", 3840, 2160, dict(order=1, aa=48, aac=48)),
    "2160p-YUV420P16-dh": ("YUV420P16", 3840, 1080, dict(order=1, aa=48, aac=48, dh=True)),
    "2160p-YUV444PS-dh": ("YUV444PS", 3840, 1080, dict(order=1, aa=48, aac=48, dh=True)),
    "2160p-Y32": ("Y32", 3840, 2160, dict(order=1, aa=48)),
    "2160p-YUV420PS": ("YUV420PS", 3840, 2160, dict(order=1, aa=48, aac=48)),
    "2160p-YUV422P16": ("YUV422P16", 3840, 2160, dict(order=1, aa=48, aac=48)),
    "1080p-YUV420P16": ("YUV420P16", 1920, 1080, dict(order=1, aa=48, aac=48)),
    "1080p-YUV420PS": ("YUV420PS", 1920, 1080, dict(order=1, aa=48, aac=48)),
    # DCI 4K: wider than eight strips of 480 columns -- 8-bit planes still sweep (two strips per wave), 8-bit 4:2:0 chroma in the
    # two-sweep form, 16-bit planes on the pool path
    "dci4k-Y8": ("Y8", 4096, 2160, dict(order=1, aa=48)),
    "dci4k-YUV420P8": ("YUV420P8", 4096, 2160, dict(order=1, aa=48, aac=48)),
    "dci4k-Y16": ("Y16", 4096, 2160, dict(order=1, aa=48)),
}

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8 TB/s


def algorithmic_bytes_per_frame(clip, flt, kw) -> int:
    """per processed plane read the kept field once and write the whole output plane
    (1.5 * w * h_out * B); per copied plane 2 * w * h * B.  Scratch buffers count zero."""
    total = 0
    for p in range(flt.nplanes):
        h_out, w = flt.plane_shape_out(p)
        processed = kw.get("dh", False) or (kw.get("luma", True) if p == 0 else kw.get("chroma", True))
        total += int((1.5 if processed else 2.0) * w * h_out * clip.bytes)
    return total


PROFILE_TAG = "r4"  # profiles/<tag>_counters.json is what recorded_counters() reads


def recorded_counters(workload: str, batch: int):
    """Per-launch PMC figures of this workload from the committed rocprofv3 passes: profiles/r2_counters.json
    (written by tools/pmc_summary.py --json from the passes of tools/profile_round.sh / profile_workloads.sh:
    FETCH_SIZE and WRITE_SIZE in separate passes, KiB units, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes
    for gfx950 coalesced reads; SQ counters in a pass of their own).  The counters cannot be read from inside
    this process, so these are RECORDED values: valid only for the workload and batch they were collected on
    (None otherwise), and the bench line says so in `traffic_source`."""
    path = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_counters.json")
    try:
        rec = json.load(open(path)).get(workload)
        if not rec or rec.get("frames_per_launch") != batch:
            return None
        rec = dict(rec)
        rec["source"] = f"recorded:profiles/{PROFILE_TAG}_counters.json"
        return rec
    except (OSError, ValueError):
        return None


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_topology() -> dict:
    """Logical CPUs this process may run on, and the physical cores / sockets behind them (/proc/cpuinfo)."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        allowed = list(range(os.cpu_count() or 1))
    cores, sockets = set(), set()
    try:
        cpu = phys = core = None
        for line in list(open("/proc/cpuinfo")) + ["\n"]:
            if line.startswith("processor"):
                cpu = int(line.split(":")[1])
            elif line.startswith("physical id"):
                phys = int(line.split(":")[1])
            elif line.startswith("core id"):
                core = int(line.split(":")[1])
            elif not line.strip():
                if cpu is not None and cpu in allowed:
                    cores.add((phys, core))
                    sockets.add(phys)
                cpu = phys = core = None
    except (OSError, ValueError):
        pass
    n_cores = len(cores) or len(allowed)
    quota = None  # CPU time the container may use, in CPUs (cgroup v2 cpu.max / v1 cfs quota); None = unlimited
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        quota = None if q == "max" else round(int(q) / int(per), 2)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            quota = round(q / per, 2) if q > 0 else None
        except (OSError, ValueError):
            pass
    return {"logical_cpus": len(allowed), "physical_cores": n_cores, "sockets": len(sockets) or 1,
            "smt": len(allowed) > n_cores, "cgroup_cpu_quota": quota}


def cpu_baseline(fmt, w, h, kw, seconds_target=7.0):
    """A CPU port of the reference's path timed on this box's host cores, one filter instance per thread (the
    reference's MT_MULTI_INSTANCE model), bounded sample: every thread filters frames until the time is up.  Two
    runs: one thread per PHYSICAL core the process may use (`value`, `cores`), and 16 threads (`value_16_threads`, the
    figure of rounds 1 and 2).  For 8-bit Y clips on an AVX2 host this is oracle/sangnom_vec.c -- the oracle's arithmetic
    written so that gcc vectorises it, standing in for the reference's opt=1 SSE2 path (which cannot be built here) --
    otherwise the scalar oracle."""
    from avisynth_sangnom2_amd import clip_format, synth
    from oracle.oracle import Config, Oracle, VecOracleY8, vec_lib

    clip = clip_format(fmt, w, h)
    topo = cpu_topology()
    src = synth.frame(clip, "noise", seed=1)
    vec = fmt == "Y8" and not kw.get("dh") and not kw.get("fresh_pool") and vec_lib() is not None
    if vec:
        def make():
            o = VecOracleY8(w, h, kw.get("order", 1), kw.get("aa", 48))
            return o, (lambda d, o=o: o.process(src[0], dst=d[0]))
        label = "vectorised port (oracle/sangnom_vec.c, gcc -O3 -mavx2, opt=0 arithmetic)"
    else:
        cfg = Config(width=w, height=h, bytes=clip.bytes, bits=clip.bits, planes=clip.planes, subw=clip.subw, subh=clip.subh,
                     **{k: v for k, v in kw.items() if k not in ("isolated_planes", "fresh_pool")})  # the reference's pool only
        def make():
            o = Oracle(cfg)
            return o, (lambda d, o=o: o.process(src, dst=d))
        label = "scalar port (oracle/sangnom_oracle.c)"
    out_h = h * 2 if kw.get("dh") else h
    proto = [np.zeros((out_h >> (clip.subh if p else 0), w >> (clip.subw if p else 0)), dtype=clip.dtype) for p in range(min(clip.planes, 3))]
    # single thread, one frame (after a first call that touches the memory)
    o, run = make()
    run(proto)
    t0 = time.perf_counter()
    run(proto)
    one = time.perf_counter() - t0

    def timed(threads, seconds):
        workers = [make() for _ in range(threads)]
        dsts = [[d.copy() for d in proto] for _ in range(threads)]
        for i in range(threads):
            workers[i][1](dsts[i])  # first touch outside the timed region
        done = [0] * threads
        start = threading.Barrier(threads + 1)

        def work(i):
            start.wait()
            end = time.perf_counter() + seconds
            n = 0
            while True:
                workers[i][1](dsts[i])
                n += 1
                if time.perf_counter() >= end:
                    break
            done[i] = n

        ths = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
        for t in ths:
            t.start()
        start.wait()
        t0 = time.perf_counter()
        for t in ths:
            t.join()
        dt = time.perf_counter() - t0
        return sum(done), dt

    physical = max(1, topo["physical_cores"])
    quota = topo["cgroup_cpu_quota"]
    # The cores this process can really use: one thread per physical core, but no more threads than the container's CPU
    # quota pays for (the GPU boxes show 2 x 64 cores / 256 logical CPUs under a cgroup quota of 16 CPUs: 128 threads
    # there are throttled to 16 CPUs' worth of time and thrash the caches -- 2.6 Gpixel/s against 11.2 with 16 threads).
    cores = physical if quota is None else max(1, min(physical, int(quota + 0.5)))
    frames, dt = timed(cores, seconds_target)
    rec = {"value": round(frames * w * out_h / dt / 1e6, 2), "unit": "Mpixels/s", "cores": cores, "kind": "port",
           "cpu": cpu_model(), "sockets": topo["sockets"], "physical_cores": physical, "logical_cpus": topo["logical_cpus"],
           "smt": topo["smt"], "cgroup_cpu_quota": quota}
    note = ""
    if cores != physical:  # what one thread per physical core gives under the quota, for the record
        fa, da = timed(physical, 0.4 * seconds_target)
        rec["value_all_physical_cores"] = round(fa * w * out_h / da / 1e6, 2)
        note = f"; {physical} threads (one per physical core, throttled by the quota): {fa} frames in {da:.1f} s"
    if cores > 16:
        f16, d16 = timed(16, 0.4 * seconds_target)
        rec["value_16_threads"] = round(f16 * w * out_h / d16 / 1e6, 2)
        note += f"; 16 threads: {f16} frames in {d16:.1f} s"
    rec["sample"] = (f"{label}: {frames} frames of {fmt} {w}x{h} uniform noise in {dt:.1f} s, {cores} threads, one filter instance "
                     f"per thread ({topo['sockets']} socket(s), {physical} physical cores, SMT {'on' if topo['smt'] else 'off'}, "
                     f"{topo['logical_cpus']} logical CPUs visible, cgroup CPU quota {quota}){note}; single thread "
                     f"{one * 1e3:.0f} ms/frame; {cpu_model()}")
    return rec


def verify_against_pool_path(torch, SangNom2, clip, dev_index, stream, kw, src, dst, batch, max_batch, mode):
    """History-free clips: frames 0 and batch - 1 of `dst` (written by the timed launches) against a mode="pool" context fed the
    same source frames -- a frame does not depend on its neighbours there, so a fresh pool-path instance reproduces any frame of
    the batch on its own.  History-carrying clips (width % 32 != 0, chroma only): a frame depends on every frame and every launch
    before it, so the timed output cannot be reproduced in isolation; instead a FRESH context of the benchmarked kind and a fresh
    pool-path context both replay the first frames of the batch from the same (zero) state and must agree on all of them."""
    t0 = time.perf_counter()
    with SangNom2(clip, device=dev_index, max_batch=min(max_batch, 8), mode="pool", stream=stream.cuda_stream, **kw) as ref:
        history_free = bool(ref.info().history_free)
        if history_free:
            checked, equal = [], True
            for f in sorted({0, batch - 1}):
                one_s = [t[f:f + 1] for t in src]
                one_d = [torch.empty_like(t[f:f + 1]) for t in dst]
                ref.process_batch(one_s, one_d)
                torch.cuda.synchronize()
                equal = equal and all(bool(torch.equal(a[0], b[f])) for a, b in zip(one_d, dst))
                checked.append(f)
            return {"ok": equal, "frames": len(checked), "which": checked, "against": "pool path (mode=pool context, one frame per launch)",
                    "planes": len(dst), "seconds": round(time.perf_counter() - t0, 3)}
        n = min(batch, max_batch, 8)
        part = [t[:n] for t in src]
        want = [torch.empty_like(t[:n]) for t in dst]
        got = [torch.empty_like(t[:n]) for t in dst]
        ref.process_batch(part, want)
        with SangNom2(clip, device=dev_index, max_batch=n, mode=mode, stream=stream.cuda_stream, **kw) as again:
            again.process_batch(part, got)
            torch.cuda.synchronize()
        equal = all(bool(torch.equal(a, b)) for a, b in zip(want, got))
        return {"ok": equal, "frames": n, "which": list(range(n)),
                "against": "pool path; history-carrying clip: a fresh context of the benchmarked kind and a fresh pool-path context replay the "
                           "first frames of the batch in one launch each (the timed launches' frames depend on every launch before them)",
                "planes": len(dst), "seconds": round(time.perf_counter() - t0, 3)}


def spawn_ranks(args) -> int:
    """`--gpus N` without a launcher: run the N ranks as children of this process, which has not touched the
    GPU (no torch.cuda call, no HIP library loaded) and never will -- a process that has initialised the GPU
    must not exec or fork workers.  Rank 0 prints the JSON line; it is passed through."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="2160p-Y8", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="frames per step and per GPU (0 = auto)")
    ap.add_argument("--mode", default="auto", choices=["auto", "pool", "fused"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the comparison with the pool path (knock-out builds of tools/ab_bench.sh only: the line then says so)")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ.get('WORLD_SIZE')} ranks")

    import torch
    import torch.distributed as dist

    from avisynth_sangnom2_amd import SangNom2, capi, clip_format

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not os.path.exists(capi.LIB_PATH):  # a checkout that has not run __graft_entry__.build() yet
        if local_rank == 0:
            capi.build()
        while not os.path.exists(capi.LIB_PATH):
            time.sleep(1.0)
    if "WORLD_SIZE" in os.environ:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    # Rehearsal on a box with fewer GPUs than ranks: SN_BENCH_DEVICE pins every rank to one device and
    # SN_BENCH_BACKEND=gloo replaces RCCL for the barrier and the max-reduce (RCCL refuses two ranks on one GPU).
    backend = os.environ.get("SN_BENCH_BACKEND", "nccl")
    dev_index = int(os.environ.get("SN_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    launched = "WORLD_SIZE" in os.environ and "MASTER_PORT" in os.environ  # under torchrun, also with one rank
    if launched:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    fmt, w, h, kw = WORKLOADS[args.workload]
    clip = clip_format(fmt, w, h)
    frame_in_bytes = sum((h >> (clip.subh if p else 0)) * (w >> (clip.subw if p else 0)) * clip.bytes
                         for p in range(min(clip.planes, 3)))
    # Ring of distinct frames far larger than the 256 MiB Infinity Cache (in + out), SURVEY.md 7-H7.
    # One fused-kernel workgroup sweeps one plane of one frame; 2048 waves are resident at a time (two per
    # SIMD), and a launch is sized to four such rounds so that uneven workgroup durations even out instead
    # of leaving SIMDs idle at the end of a single round.  Capped at 48 GiB of in + out.
    sweep_w = (w + 31) // 32 * 32
    strips = 1 if sweep_w <= 512 else 1 + -(-(sweep_w // 8 - 62) // 60)
    waves_per_frame = strips if clip.bytes >= 2 else (strips + 1) // 2
    out_bytes = frame_in_bytes * (2 if kw.get("dh") else 1)
    per_round = 256 * max(1, 8 // waves_per_frame)  # workgroups resident at a time (planes too wide for the sweeps: as for eight waves)
    fit = (48 << 30) // (frame_in_bytes + out_bytes)
    if clip.planes >= 3 and clip.subw + clip.subh > 0 and not (kw.get("isolated_planes") or kw.get("fresh_pool")):
        # the 4:2:0 sweeps also need hand-off pools per frame: two, or one where U and V run as one sweep (8-bit up to 3840 columns)
        pools = 1 if clip.bytes == 1 and sweep_w <= 3840 else 2
        fit = min(fit, (24 << 30) // (pools * 9 * ((h * (2 if kw.get("dh") else 1)) // 4 + 3) * waves_per_frame * 64 * 16))
    rounds = max(1, min(4, fit // per_round))
    batch = args.batch or rounds * per_round
    stream = torch.cuda.Stream(dev)  # a real (non-null) HIP stream shared with the context, so that
    # torch.cuda.Event timing below sees exactly the kernels the library launches
    flt = SangNom2(clip, device=dev_index, max_batch=batch, mode=args.mode, stream=stream.cuda_stream, **kw)
    tdt = {1: torch.uint8, 2: torch.int16, 4: torch.float32}[clip.bytes]
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    src, dst = [], []
    for p in range(flt.nplanes):
        hi, wi = flt.plane_shape_in(p)
        ho, wo = flt.plane_shape_out(p)
        if clip.bytes == 4:
            s = torch.rand((batch, hi, wi), device=dev, generator=g, dtype=torch.float32)
        elif clip.bytes == 2:
            s = torch.randint(0, 1 << clip.bits, (batch, hi, wi), device=dev, generator=g, dtype=torch.int32).to(torch.int16)
        else:
            s = torch.randint(0, 256, (batch, hi, wi), device=dev, generator=g, dtype=torch.uint8)
        src.append(s)
        dst.append(torch.empty((batch, ho, wo), device=dev, dtype=tdt))
    out_h = flt.out_height
    alg_bytes = algorithmic_bytes_per_frame(clip, flt, kw) * batch

    def barrier():
        if launched:
            dist.barrier()

    torch.cuda.synchronize(dev)  # inputs were generated on the default stream
    gpu_t0 = time.perf_counter()
    wev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    wev[0].record(stream)
    for _ in range(args.warmup):
        flt.process_batch(src, dst)
    wev[1].record(stream)
    torch.cuda.synchronize(dev)

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record(stream)
        flt.process_batch(src, dst)
        ev[i][1].record(stream)
    torch.cuda.synchronize(dev)
    barrier()
    elapsed = time.perf_counter() - t0
    own_elapsed = elapsed
    if launched:
        t = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-launch device time of the hot path on the stream it runs on (HIP events)
    dev_ms = sorted(a.elapsed_time(b) for a, b in ev)
    launch_ms = sum(dev_ms) / len(dev_ms)
    info = flt.info()
    gpu_active_ms = sum(dev_ms) + (wev[0].elapsed_time(wev[1]) if args.warmup else 0.0)

    # The line vouches for its own output (outside the timed region): the first and the last frame of the batch the
    # timed launches wrote are recomputed by a SECOND context on the three-kernel pool path over the HBM-resident pool
    # (sn_pool_kernels.hip: other kernels, other data layout, the semantic reference of the sweeps on the GPU) and must
    # be byte-equal; a mismatch is an error, not a note.  What both compute: /root/reference/src/SangNom2.cpp:332-397.
    verified = None
    if rank == 0 and args.no_verify:
        verified = {"ok": None, "frames": 0, "against": "nothing (--no-verify)"}
    elif rank == 0:
        verified = verify_against_pool_path(torch, SangNom2, clip, dev_index, stream, kw, src, dst, batch, batch, args.mode)
        if verified["ok"] is False:
            print(json.dumps({"error": "bench.py: the timed launches' output differs from the pool path", "verified": verified}), flush=True)
            raise SystemExit(3)

    joined = dist.get_world_size() if launched else 1  # ranks that really took part
    # every rank's own figures, so that a straggler shows in the line (the aggregate uses the slowest rank's time)
    mine = {"rank": rank, "device": dev_index, "frames_per_s": round(batch * args.steps / own_elapsed, 1),
            "kernel_ms_per_launch": round(launch_ms, 4)}
    per_rank = [mine]
    if launched:
        per_rank = [None] * joined
        dist.all_gather_object(per_rank, mine)
    if rank == 0:
        frames_total = batch * args.steps * joined
        mpix = frames_total * w * out_h / elapsed / 1e6
        achieved = alg_bytes / (launch_ms * 1e-3) / 1e9
        rec = recorded_counters(args.workload, batch) or {}
        traffic = rec.get("hbm_bytes_per_launch")
        # What limits the sweep is the vector ALU's issue rate, not HBM: the HBM figures below are
        # the roofline north_star asks for, `valu` says how the limiting unit is used.  Instruction counts are a
        # property of the build (recorded PMC pass), the time is this run's.
        valu = None
        if rec.get("valu_insts_per_launch"):
            simds, clock_hz = 256 * 4, 2.4e9
            insts = rec["valu_insts_per_launch"]
            valu = {"wave_insts_per_launch": insts,
                    "lane_ops_per_output_pixel": round(insts * 64 / (batch * w * out_h), 2),
                    "cycles_per_inst_per_simd": round(launch_ms * 1e-3 * clock_hz * simds / insts, 3),
                    "clock_ghz_assumed": 2.4,
                    "valu_issue_share_of_wave_life": rec.get("valu_active_frac"),
                    "stall_share_of_wave_life": rec.get("stall_frac"),
                    "source": rec["source"]}
        out = {
            "metric": "Mpixels/s", "value": round(mpix, 1), "unit": "Mpixels/s",
            "n_gpus": joined, "steps": args.steps, "warmup": args.warmup,
            "distributed": {"backend": (backend + (" (RCCL)" if backend == "nccl" else "")) if launched else None,
                            "world_size": joined, "collectives": "barrier before and after the timed steps, max-reduce of the elapsed time"
                                                                 if launched else None,
                            "per_rank_frames_per_s": [r["frames_per_s"] for r in per_rank],
                            "per_rank_kernel_ms": [r["kernel_ms_per_launch"] for r in per_rank],
                            "min_rank_frames_per_s": min(r["frames_per_s"] for r in per_rank),
                            "max_rank_frames_per_s": max(r["frames_per_s"] for r in per_rank)},
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": {1: "u8", 2: "u16", 4: "f32"}[clip.bytes],
            "data": "synthetic",
            "config": {"workload": f"{args.workload} order={kw.get('order', 1)} aa={kw.get('aa', 48)} "
                                   f"aac={kw.get('aac', 0)} dh={int(kw.get('dh', False))}, device-resident stream",
                       "frames_per_step_per_gpu": batch, "frame": f"{w}x{out_h} {fmt}",
                       "path": "fused" if info.fused_frames > 0 else "pool", "sharding": "frames, no collective"},
            "frames_per_s": round(frames_total / elapsed, 1),
            "verified": verified,
            "roofline": {"bound": "valu-issue", "roofline_reported": "hbm", "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": rec.get("source") if traffic else None,
                         "kernel_ms_per_launch": round(launch_ms, 4),
                         "algorithmic_bytes_per_launch": alg_bytes, "valu": valu},
        }
        if world == 1 and not args.no_cpu_baseline:  # (the profiling passes run with --no-cpu-baseline: the hot path only)
            # the other end of the scale: ONE frame per launch (what a synchronous GetFrame sees), device time per frame;
            # auto mode cuts such a launch into row bands where the planes allow it
            one_s, one_d = [t[:1] for t in src], [t[:1] for t in dst]
            with SangNom2(clip, device=dev_index, max_batch=1, stream=stream.cuda_stream, **kw) as lat:
                for _ in range(5):
                    lat.process_batch(one_s, one_d)
                a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a0.record(stream)
                for _ in range(50):
                    lat.process_batch(one_s, one_d)
                a1.record(stream)
                torch.cuda.synchronize(dev)
                li = lat.info()
                out["single_frame"] = {"ms": round(a0.elapsed_time(a1) / 50, 4), "banded": li.banded_frames > 0,
                                       "band_fallbacks": int(li.band_fallbacks), "launches": 50}
        # device time of the launches this process queued (HIP events: warm-up + timed steps; the single-frame and
        # verification legs add a few ms), and the wall-clock window they fell in -- the CPU leg below keeps the GPU
        # idle for ~10 s, which is what a utilisation sampler sees most of the run
        out["gpu_active_s"] = round(gpu_active_ms / 1e3, 4)
        out["gpu_window_s"] = round(time.perf_counter() - gpu_t0, 3)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(fmt, w, h, kw)
        print(json.dumps(out), flush=True)
    flt.close()
    if launched:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
