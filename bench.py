#!/usr/bin/env python3
"""bench.py -- pose-candidate NMI evaluations per second on MI355X (BASELINE.json metric).

One "step" = one search over one candidate grid with the inputs already resident in HBM: the whole
6-D grid (S renders x Wn warps) is scored by the HIP kernel, the arg-max is taken on the device and
the 8-byte winner is read back.  N=1 runs BASELINE.json configs[1]: 640x480 frames,
729-pose grid (27 renders x 27 warps), 256 bins.  N>1 (one process per GPU, launched by
torch.distributed.run) shards the render axis: every rank holds its own 27 renders and the full warp
stack (weak scaling), and the only exchange is an 8-byte MAX all-reduce of the packed winner over RCCL.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WIDTH, HEIGHT, S_PER_RANK, WN, BINS = 640, 480, 27, 27, 256
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes_per_eval(w, h):
    return 2 * w * h + 4  # one u8 read of each image + one f32 score


def cpu_baseline(wl, budget_s=12.0, max_threads=16):
    """The CPU oracle (a port: the reference has no CPU NMI path) on the host cores, bounded sample."""
    from oracle import binding as oc
    # the host cores this process may use (cgroup / affinity share of the box, not the machine's total)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    # a 1-GPU box's CPU share is 16 cores (harness rule); never oversubscribe that
    threads = max(1, min(oc.max_threads(), avail, max_threads))
    rs, ws = wl["render_stack"], wl["warp_stack"]
    oc.search_grid(rs[:threads], ws[:2], threads=threads, render_bottom_up=wl["bottom_up"])  # warm-up
    evals, t0 = 0, time.perf_counter()
    reps = 0
    while True:
        oc.search_grid(rs, ws, threads=threads, render_bottom_up=wl["bottom_up"])
        evals += rs.shape[0] * ws.shape[0]
        reps += 1
        if time.perf_counter() - t0 >= budget_s:
            break
    dt = time.perf_counter() - t0
    # SURVEY.md 8(d) also asks for the single-thread figure: one pair at a time, 256 bins and 64 bins (BASELINE configs[0])
    single = {}
    for bins, shift in ((256, 0), (64, 2)):
        n, t1 = 0, time.perf_counter()
        while time.perf_counter() - t1 < 0.7:
            oc.eval_pair(rs[n % rs.shape[0]], ws[n % ws.shape[0]], shift=shift, render_bottom_up=wl["bottom_up"])
            n += 1
        single[f"evals_per_s_{bins}_bins"] = n / (time.perf_counter() - t1)
    return {"value": evals / dt, "unit": "evals/s", "cores": threads, "kind": "port",
            "sample": f"{reps} x the same {rs.shape[0]}x{ws.shape[0]} grid at {WIDTH}x{HEIGHT}, {BINS} bins, "
                      f"OpenMP over candidates, {dt:.1f} s",
            "single_thread": single}


def kernel_source_id():
    """sha256 (first 16 hex digits) of the dominant kernel's sources: the stamp tools/summarize_profiles.py puts on a PMC
    profile when it is taken, and what bench.py compares it with -- a committed counter profile of another kernel says so."""
    import hashlib
    h = hashlib.sha256()
    for name in ("nmi_kernels.hip", "nmi_device.h", "nmi_kernels.h"):
        with open(os.path.join(ROOT, "orbslam2_nmi_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def load_pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed PMC profile (profiles/), or None."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def static_profile_stamp():
    """roofline.traffic / roofline.lds are NOT measured by this run (PMC passes need rocprofv3): they come from the committed
    profiles/pmc_*.json.  This says so, names the profile and whether the kernel sources still are the ones it was taken on."""
    out = {"static_profile": True, "kernel_source_sha16_now": kernel_source_id()}
    for key, name in (("traffic", "pmc_traffic.json"), ("lds", "pmc_lds.json")):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.load(f)
            out[key] = {"source": d.get("source"), "kernel_source_sha16": d.get("kernel_source_sha16"),
                        "stale": d.get("kernel_source_sha16") != out["kernel_source_sha16_now"]}
        except (OSError, ValueError):
            out[key] = None
    return out


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args, argv):
    """`python3 bench.py --gpus N` without a launcher (the way the round-end driver calls it): start the N ranks ourselves as
    CHILD processes of torch.distributed.run -- before this process has touched the GPU, and never by replacing it (an exec
    from a process that initialised HIP takes the box down) -- relay their output and return their exit code.
    torch.cuda.device_count() does not initialise the device on this image."""
    import subprocess

    import torch
    visible = torch.cuda.device_count()
    if not (args.all_on_device0 or args.dry_launch) and visible < args.gpus:
        print(f"bench.py: --gpus {args.gpus} needs {args.gpus} visible HIP devices, this host shows {visible} "
              f"(rehearsal on fewer devices: --backend gloo --all-on-device0)", file=sys.stderr)
        return 2
    if args.all_on_device0 and args.backend == "nccl" and args.gpus > 1:
        print("bench.py: --all-on-device0 needs --backend gloo (RCCL refuses two ranks on one device)", file=sys.stderr)
        return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL's intra-node transport needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode


def dry_launch(args):
    """--dry-launch: every rank reports the rendezvous it was given and the device it WOULD select, touching no GPU; the
    ranks meet once over gloo (so the launch, the rendezvous and a collective are exercised on a CPU-only host)."""
    import torch
    import torch.distributed as dist
    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1")))
    if world != args.gpus:
        sys.exit(f"bench.py: WORLD_SIZE {world} does not match --gpus {args.gpus}")
    mine = {"rank": rank, "local_rank": local_rank, "device": 0 if args.all_on_device0 else local_rank,
            "master": f"{os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}", "pid": os.getpid()}
    ranks = [mine]
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)
        t = torch.tensor([rank + 1], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert int(t.item()) == world
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"dry_launch": True, "n_gpus": world, "config": args.config, "backend": args.backend,
                          "devices_visible": torch.cuda.device_count(), "ranks": ranks}))
    return 0


def rank_evidence(world, dist, local_device):
    """What a reader needs to see that the collective really spanned N ranks: the size of the process group as the backend
    reports it, the backend's name and every rank's device ordinal (gathered through that same backend)."""
    import torch
    if dist is None:
        return {"collective_ranks": 1, "rccl": False, "backend": "none", "device_ordinals": [local_device]}
    backend = dist.get_backend()
    t = torch.zeros(world, dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
    t[dist.get_rank()] = local_device + 1
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return {"collective_ranks": dist.get_world_size(), "rccl": backend == "nccl", "backend": "nccl (RCCL)" if backend == "nccl" else backend,
            "device_ordinals": [int(v) - 1 for v in t.cpu().tolist()]}


def init_dist(args):
    """(rank, local_rank, world, dist or None).  One process per GPU; main() has already started the ranks when needed."""
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if args.gpus != 1:
            sys.exit(f"bench.py: WORLD_SIZE {world} does not match --gpus {args.gpus}")
        args.gpus = world  # started by torch.distributed.run without --gpus: take the launcher's word
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device (no CPU fallback for the NMI path)")
    if args.all_on_device0:
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        sys.exit(f"bench.py: rank {rank} wants device {local_rank}, only {torch.cuda.device_count()} visible")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    return rank, local_rank, world, dist


def timed_region(fn, dist):
    """barrier + synchronize on both sides of fn(); returns (MAX over ranks of the elapsed seconds, fn's result)."""
    import torch

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    sync_all()
    t0 = time.perf_counter()
    out = fn()
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, out


def call_site_rate(timeout_s=120):
    """The reference's UNCHANGED per-candidate call site (src/Tracking.cc:1886-1894: one blocking
    CUDAF::NMIWithCuda_noMask per candidate, here through host/cudaf_shim.hpp) measured by the C++ program
    examples/relocalize_demo, part C, run as a CHILD process after the timed region -- never inside it."""
    import re
    import subprocess
    exe = os.path.join(ROOT, "examples", "relocalize_demo")
    if not os.access(exe, os.X_OK):
        return {"error": "examples/relocalize_demo is not built (python -c 'import __graft_entry__ as g; g.build()')"}
    try:
        r = subprocess.run([exe], capture_output=True, text=True, timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return {"error": "examples/relocalize_demo timed out"}
    m1, m2 = re.search(r"SHIM_EVALS_PER_S (\d+)", r.stdout), re.search(r"SHIM_BATCHED_EVALS_PER_S (\d+)", r.stdout)
    if r.returncode != 0 or not m1 or not m2:
        return {"error": f"examples/relocalize_demo failed (rc {r.returncode})"}
    rate = float(m1.group(1))
    return {"evals_per_s": rate, "us_per_call": 1e6 / rate, "batched_evals_per_s": float(m2.group(1)),
            "what": "640x480, one blocking CUDAF::NMIWithCuda_noMask per candidate through the identical-signature shim "
                    "(examples/relocalize_demo part C, child process after the timed region); batched = BeginBatch / Flush around the warp loop",
            "target_evals_per_s": 50000, "target_met": rate >= 50000.0}


def load_pmc_lds():
    """LDS-side counters of the dominant kernel from the committed PMC profile (profiles/pmc_lds.json), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_lds.json")) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None
    return {k: d.get(k) for k in ("lds_wave_instructions_per_launch", "lds_idx_active_cycles_per_launch", "conflict_ratio",
                                  "lds_active_cycles_per_wave_instruction", "conflict_free_cycles_per_wave_instruction", "source")}


def native_comm(ctx, rank, world, dist):
    """ncclComm_t for the product's own collectives (nmi_level_run_rccl / nmi_stream_submit_block): rank 0 draws the unique id,
    torch.distributed ships its 128 bytes, every rank joins.  None with one rank or a non-RCCL rehearsal backend."""
    from orbslam2_nmi_amd import capi
    if dist is None or world == 1 or dist.get_backend() != "nccl":
        return None
    box = [capi.rccl_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return ctx.rccl_comm_init(box[0], rank, world)


VALU_PEAK_WAVE_INSTR_PER_S = 256 * 4 * 2.4e9 / 2   # a wave64 instruction issues over 2 cycles on a SIMD-32 (MI355X_MICROARCH.md:54)
ATOMIC_PEAK_BYTES_PER_S = 1.3e12                      # memory-side execution of well-shaped dword atomics, as 64-byte requests (same guide)


def producer_rooflines(map_key, w, h, level_ms):
    """Per-kernel roofline lines of a default-mode level.  Durations and counters are NOT measured by this run
    (the level is one graph launch; per-kernel counters need rocprofv3 passes of their own): they come from the committed profile
    profiles/r04_b/pmc_producers.json (tools/collect_profiles_producers.sh), and the block says so; level_ms is this run's."""
    path = os.path.join(ROOT, "profiles", "r04_b", "pmc_producers.json")
    try:
        with open(path) as f:
            prof = json.load(f)[map_key]
    except (OSError, ValueError, KeyError):
        return None
    out = {"static_profile": "profiles/r04_b/pmc_producers.json", "map": map_key, "level_ms_this_run": level_ms, "kernels": {}}
    in_level = ("nmi_level_prep", "nmi_level_front", "nmi_zbuf_resolve", "nmi_mesh_bin", "nmi_mesh_cull", "nmi_mesh_clip", "nmi_mesh_tile", "nmi_grid_kernel")
    for name, e in prof.items():
        if not name.startswith(in_level):   # (set-up kernels of the bench itself: term table, cloud packing, the frame's own render)
            continue
        t = e["time_us"] * 1e-6
        k = {"time_us": e["time_us"]}
        d = e.get("derived", {})
        if name.startswith("nmi_grid_kernel"):
            a = 729 * algorithmic_bytes_per_eval(w, h) / t / 1e9
            k.update({"bound": "hbm (algorithmic bytes, the contract's definition; the kernel itself is LDS-atomic-bound)", "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a / HBM_PEAK_GBS})
        elif name.startswith("nmi_mesh_tile") or name.startswith("nmi_mesh_bin"):
            a = e.get("SQ_INSTS_VALU", 0.0) / t
            k.update({"bound": "valu-issue (wave64 instructions per second over 1024 SIMD-32s)", "achieved": a, "peak": VALU_PEAK_WAVE_INSTR_PER_S, "unit": "wave-instr/s",
                      "frac": a / VALU_PEAK_WAVE_INSTR_PER_S, "wave_cycles_waiting_frac": d.get("wave_cycles_waiting_frac")})
            if name.startswith("nmi_mesh_tile"):
                k["stores_frac_of_hbm_peak"] = 27 * w * h / t / 1e9 / HBM_PEAK_GBS   # the 27 renders it writes, once
        elif name == "nmi_level_front_kernel":
            # 8.9 M atomicMin per level of this benchmark's cloud; a wavefront's 64 anchors are a 71-pixel streak = 5.4 64-byte requests
            # per instruction, 745 k requests per level (profiles/NOTES.md, round 3: the request count is MODELLED from the cloud's
            # geometry, the duration is measured)
            a = 745e3 * 64 / t
            k.update({"bound": "memory-side atomics (64-byte request bytes per second; request count modelled)", "achieved": a, "peak": ATOMIC_PEAK_BYTES_PER_S,
                      "unit": "B/s", "frac": a / ATOMIC_PEAK_BYTES_PER_S, "wave_cycles_waiting_frac": d.get("wave_cycles_waiting_frac")})
        elif name.startswith("nmi_zbuf_resolve"):
            a = (e.get("fetch_bytes", 0.0) + e.get("write_bytes", 0.0)) / t / 1e9
            k.update({"bound": "hbm (fabric-side bytes from the counters)", "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a / HBM_PEAK_GBS})
        else:
            k.update({"bound": "latency (launch + one dependent round trip; a few microseconds of fixed cost)", "wave_cycles_waiting_frac": d.get("wave_cycles_waiting_frac")})
        out["kernels"][name] = k
    return out


LEVEL_SHARDED = ("level-sharded: each level's 27 views are dealt to the ranks (render axis; the warp axis when there are fewer views "
                 "than ranks), every rank makes its own warps, one 8-byte MAX all-reduce of the packed winner per level")


def run_stream_config(args):
    """BASELINE.json configs[4]: 100-keyframe synthetic sequence, 3 search levels per keyframe (steps halved per level like
    NmiSearchKernel::resizeKernel), 3^6 candidates per level at 848x480, render stacks streamed from pinned host memory
    through the double-buffered pipeline (nmi_stream_*), warp stacks produced on the device per level.
    N ranks: the keyframes of the sequence are dealt round-robin (sharding.keyframe_share: replicas, no collective on the
    data path -- the levels of one keyframe are sequential in the reference, Tracking.cc:2088-2130, keyframes are not);
    every rank streams its own share through its own pipeline and the per-level winners are gathered with one all-reduce
    at the end, inside the timed region.  Total work is fixed (--keyframes): "scaling": "strong".
    Not the headline line: prints its own JSON (keyframes/s and evals/s, H2D included)."""
    import torch

    import orbslam2_nmi_amd as nmi
    from orbslam2_nmi_amd import capi, sharding, synthetic as sy

    rank, local_rank, world, dist = init_dist(args)
    w, h, counts, levels, pool = 848, 480, (3, 3, 3), 3, 4
    K = sy.intrinsics(w, h)
    frames, stacks, homs = [], [], []
    for kf in range(pool):
        B = sy.scene(w, h, 9000 + kf)
        frames.append(torch.from_numpy(sy.camera_frame(B, 9500 + kf)).pin_memory())
        for lvl in range(levels):
            stacks.append(torch.from_numpy(sy.render_stack(B, counts, shift_px=max(1, 4 >> lvl), zoom_step=0.02 / 2 ** lvl)).pin_memory())
            homs.append(capi.warp_homographies(K, counts, tuple(s / 2 ** lvl for s in (0.02, 0.02, 0.05))))
    ctx = nmi.NmiContext(w, h, render_bottom_up=False)
    st = nmi.NmiStream(ctx, 27, 27, depth=2)
    n_kf = args.keyframes
    red_dev = "cuda" if (dist is not None and dist.get_backend() == "nccl") else "cpu"
    sharded = args.shard == "level"
    comm = native_comm(ctx, rank, world, dist) if sharded else None
    so, sc_, wo, wc = sharding.grid_shard(27, 27, rank, world) if sharded else (0, 27, 0, 27)

    def run(nk):
        table = sharding.new_keyframe_table(nk, levels)
        pending = []

        def collect():
            kf, lvl, t = pending.pop(0)
            idx, sc = st.wait(t)
            if sharded and comm is None and dist is not None:  # rehearsal backend: the caller owns the exchange
                idx, sc = sharding.sharded_level(lambda *blk: (idx, sc), 27, 27, rank, world, dist)
            sharding.store_keyframe_result(table, kf, lvl, idx, sc)

        # replicas: this rank's keyframes, whole levels.  level-sharded: every keyframe, this rank's block of every level
        # (it uploads only its views: the H2D traffic that bounds this config is divided by the number of ranks)
        for kf in (range(nk) if sharded else sharding.keyframe_share(nk, rank, world)):
            for lvl in range(levels):
                p = kf % pool
                if sharded:
                    t = st.submit(stacks[p * levels + lvl][so:so + sc_], frames[p], homs[p * levels + lvl][wo:wo + wc],
                                  block=(so, 27, wo, 27), comm=comm)
                else:
                    t = st.submit(stacks[p * levels + lvl], frames[p], homs[p * levels + lvl])
                pending.append((kf, lvl, t))
                if len(pending) == 2:
                    collect()
        while pending:
            collect()
        if sharded:
            table = table.copy()
            table[..., 0] -= 1
            return table
        return sharding.gather_keyframe_table(table, world, dist, red_dev)

    run(4 * world)
    dt, table = timed_region(lambda: run(n_kf), dist)
    centre = 13 * 27 + 13
    if not (table[..., 0] == centre).all():
        sys.exit(f"rank {rank}: stream config: unexpected winners {sorted(set(table[..., 0].reshape(-1).tolist()))}")
    evidence = rank_evidence(world, dist, local_rank)
    if rank == 0:
        evals = n_kf * levels * 729
        h2d = n_kf * levels * (27 + (world if sharded else 1)) * w * h  # level-sharded: every rank uploads the frame
        print(json.dumps({"metric": "keyframes/s (BASELINE configs[4]: 848x480, 3 levels x 729 candidates, render stacks streamed H2D)",
                          "value": n_kf / dt, "unit": "keyframes/s", "evals_per_s": evals / dt, "n_gpus": world,
                          "keyframes": n_kf, "levels": levels, "h2d_GBps_all_ranks": h2d / dt / 1e9, "data": "synthetic", **evidence,
                          "higher_is_better": True, "scaling": "strong",
                          "config": {"workload": "BASELINE.json configs[4]: 100-keyframe sequence, coarse-to-fine 3 levels, double-buffered render stacks",
                                     "width": w, "height": h, "pipeline_depth": 2,
                                     "parallelism": "one rank" if world == 1 else LEVEL_SHARDED if sharded else
                                     "replicas: keyframes dealt round-robin to ranks, winners gathered by one all-reduce",
                                     "question": "latency of a live sequence (keyframes one after the other)" if sharded else
                                     "throughput of a recorded sequence (keyframes independent)"}}))
    st.close()
    if comm is not None:
        capi.rccl_comm_destroy(comm)
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def run_e2e_config(args):
    """Everything on the device: coloured point cloud + camera frame in HBM; per keyframe 3 search levels, each level =
    27 renders of the cloud (nmi_render_points) + 27 warps of the frame (nmi_warp_stack) + the 729-candidate search.
    Only matrices cross PCIe.  N ranks: keyframes dealt round-robin like --config stream (every rank holds the cloud).
    Not the headline line (the render / warp producers are the section-8f rows)."""
    import torch

    import orbslam2_nmi_amd as nmi
    from orbslam2_nmi_amd import capi, hostapi as H, sharding, synthetic as sy

    rank, local_rank, world, dist = init_dist(args)
    w, h, levels = 848, 480, 3
    K = sy.intrinsics(w, h)
    rp = capi.RenderParams(fx=K[0, 0], fy=K[1, 1], cx=K[0, 2], cy=K[1, 2], near_plane=5.0, far_plane=30.0, point_size=3.0)
    # a textured plane at 10 m, three views wide, plus the frame seen from the grid's centre pose:
    #   cloud (nmi_prop_RENDER 4): ~1.3 M coloured points (about 2 per pixel of the view)
    #   mesh  (nmi_prop_RENDER 1, the reference's default, allProperties.hpp:41): --mesh-quads NX x NY quads = 2 NX NY textured triangles
    B = sy.scene(2 * w, 2 * h, 77)
    mesh = args.map == "mesh"
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx = nmi.NmiContext(w, h)
    ctx.set_stream(stream.cuda_stream)
    tex = None
    if mesh:
        nx, ny = (int(v) for v in args.mesh_quads.lower().split("x"))
        uu, vv = np.meshgrid(np.linspace(-w, 2 * w, nx + 1), np.linspace(-h, 2 * h, ny + 1))
        P = np.stack([(uu - rp.cx) / rp.fx * 10.0, (vv - rp.cy) / rp.fy * 10.0, np.full_like(uu, 10.0)], -1).astype(np.float32)
        T = np.stack([(uu + w) / (3 * w), (vv + h) / (3 * h)], -1).astype(np.float32)
        p00, p10, p01, p11 = P[:-1, :-1], P[:-1, 1:], P[1:, :-1], P[1:, 1:]
        t00, t10, t01, t11 = T[:-1, :-1], T[:-1, 1:], T[1:, :-1], T[1:, 1:]
        # counter-clockwise as seen by the ORB-SLAM camera (y down): the front faces of rendering.hpp's glEnable(GL_CULL_FACE)
        xyz = np.ascontiguousarray(np.stack([p00, p11, p10, p00, p01, p11], 2).reshape(-1, 3))
        red = np.ascontiguousarray(np.stack([t00, t11, t10, t00, t01, t11], 2).reshape(-1, 2))   # uv
        tex = nmi.NmiTexture(ctx, np.stack([B, B, B], -1).astype(np.uint8))
    else:
        nu, nv = int(3 * w * 0.9), int(3 * h * 0.9)
        uu, vv = np.meshgrid(np.linspace(-w, 2 * w, nu), np.linspace(-h, 2 * h, nv))
        xyz = np.stack([(uu - rp.cx) / rp.fx * 10.0, (vv - rp.cy) / rp.fy * 10.0, np.full_like(uu, 10.0)], -1).reshape(-1, 3).astype(np.float32)
        red = (B[np.clip(((vv + h) / 3 * 2).astype(int), 0, 2 * h - 1), np.clip(((uu + w) / 3 * 2).astype(int), 0, 2 * w - 1)]
               .astype(np.float32) / np.float32(256)).reshape(-1)
    dx, dr = torch.from_numpy(xyz).cuda(), torch.from_numpy(red).cuda()
    Twc = np.eye(4, dtype=np.float32)  # ORB-SLAM camera axes: x right, y down, z forward (setupCam, ioData.cpp:177-197)
    pos, look, up = Twc[:3, 3], Twc[:3, 3] + Twc[:3, 2], Twc[:3, 1]
    grids = [H.SearchKernel.make([3] * 6, [s / 2 ** l for s in (0.2, 0.2, 0.5, 0.02, 0.02, 0.05)]) for l in range(levels)]
    cells = [(sx, sy, sz) for sz in range(3) for sy in range(3) for sx in range(3)]
    mvps = [np.stack([capi.render_mvp(rp, pos, look, up, H.calculate_translation(Twc, g, *c)) for c in cells]) for g in grids]
    homs = [capi.warp_homographies(K, (3, 3, 3), tuple(g.step[3:6])) for g in grids]
    centre_view = capi.render_mvp(rp, pos, look, up, (0, 0, 0))[None]
    if mesh:
        frame = ctx.render_mesh(dx, dr, tex, centre_view)[0]
        if float((frame != 255).float().mean()) < 0.9:  # back faces are culled: turn every triangle round (swap corners 1 and 2)
            flip = torch.tensor([0, 2, 1], device="cuda")
            dx = dx.view(-1, 3, 3)[:, flip].reshape(-1, 3).contiguous()
            dr = dr.view(-1, 3, 2)[:, flip].reshape(-1, 2).contiguous()
            frame = ctx.render_mesh(dx, dr, tex, centre_view)[0]
        if float((frame != 255).float().mean()) < 0.9:
            sys.exit("e2e config: the mesh does not cover the view")
        lut = torch.clamp(torch.round(255.0 * (torch.arange(256, device="cuda") / 255.0) ** 0.5), 0, 255).to(torch.uint8)  # modality gap
        frame = lut[torch.flip(frame, dims=[0]).long()].contiguous()
    else:
        frame = torch.flip(ctx.render_points(dx, torch.sqrt(dr), centre_view, 3.0)[0], dims=[0]).contiguous()
    # camera noise (sigma 10 grey levels, like the config-2 workload): without it the frame is a deterministic function of
    # the render and the joint histogram collapses onto a curve, which is the LDS atomic unit's worst case, not a camera's
    noise = torch.from_numpy(np.random.default_rng(4242).normal(0.0, 10.0, (h, w)).astype(np.float32)).cuda()
    frame = torch.clamp(torch.round(frame.float() + noise), 0, 255).to(torch.uint8).contiguous()
    rs = torch.empty((27, h, w), dtype=torch.uint8, device="cuda")
    ws = torch.empty((27, h, w), dtype=torch.uint8, device="cuda")

    sharded = args.shard == "level"
    if sharded and args.no_graph:
        sys.exit("--shard level runs the level graph (drop --no-graph)")
    so, sc_, wo, wc = sharding.grid_shard(27, 27, rank, world) if sharded else (0, 27, 0, 27)
    comm = native_comm(ctx, rank, world, dist) if sharded else None
    level = nmi.NmiLevel(ctx, dx, dr, frame, sc_, wc, 3.0, texture=tex, block=(so, 27, wo, 27)) if not args.no_graph else None

    # the three levels' parameter sets are fixed: converted to C pointers once (NmiLevel.bind), not once per replay
    bound = None
    if level is not None:
        bound = [level.bind(mvps[l][so:so + sc_], homs[l][wo:wo + wc], comm) if sharded else level.bind(mvps[l], homs[l]) for l in range(levels)]

    def keyframe():
        out = []
        for l in range(levels):
            if level is not None and comm is not None:  # graph replay + ncclAllReduce of the key, inside the library
                out.append(bound[l]())
            elif level is not None and sharded and dist is not None:  # rehearsal backend: the caller owns the exchange
                out.append(sharding.sharded_level(lambda *blk: bound[l](), 27, 27, rank, world, dist))
            elif level is not None:  # one hipGraphLaunch per level
                out.append(bound[l]())
            else:                  # the same operations enqueued one by one
                if mesh:
                    ctx.render_mesh(dx, dr, tex, mvps[l], out=rs, sync=False)
                else:
                    ctx.render_points(dx, dr, mvps[l], 3.0, out=rs, sync=False)
                ctx.warp_stack(frame, homs[l], out=ws, sync=False)
                out.append(ctx.search_grid(rs, ws))
        return out

    for _ in range(3):
        res = keyframe()
    # the frame was taken at the grid centre: at the coarse level the centre cell must win outright (at the finest level
    # neighbouring cells differ by sub-pixel shifts and pixel-snapped sprites plus camera noise decide between them)
    if res[0][0] != 13 * 27 + 13:
        sys.exit(f"rank {rank}: e2e config: unexpected coarse-level winner {res[0]}")
    red_dev = "cuda" if (dist is not None and dist.get_backend() == "nccl") else "cpu"
    n_kf = args.keyframes
    if sharded:  # every rank takes part in every level of every keyframe, and every rank ends up with every winner
        dt, table = timed_region(lambda: sharding.run_keyframes(n_kf, levels, 0, 1, lambda kf: keyframe()), dist)
    else:
        dt, table = timed_region(lambda: sharding.run_keyframes(n_kf, levels, rank, world, lambda kf: keyframe(), dist, red_dev), dist)
    if not (table[:, 0, 0] == 13 * 27 + 13).all():
        sys.exit(f"rank {rank}: e2e config: a coarse level lost the centre cell")
    evidence = rank_evidence(world, dist, local_rank)
    if rank == 0:
        what = f"27 renders of a {xyz.shape[0] // 3}-triangle textured mesh" if mesh else "27 cloud renders"
        print(json.dumps({"metric": f"keyframes/s (device end to end: 848x480, 3 levels x ({what} + 27 warps + 729-candidate search))",
                          "value": n_kf / dt, "unit": "keyframes/s", "evals_per_s": n_kf * levels * 729 / dt,
                          "n_gpus": world, "map": args.map, "triangles" if mesh else "points": int(xyz.shape[0] // 3 if mesh else xyz.shape[0]),
                          "data": "synthetic", **evidence,
                          "ms_per_level": dt / n_kf / levels * 1e3 * (1 if sharded else world),
                          "hip_graph": level is not None, "higher_is_better": True, "scaling": "strong",
                          "roofline": producer_rooflines("cloud" if not mesh else args.mesh_quads.lower(), w, h, dt / n_kf / levels * 1e3 * (1 if sharded else world))
                          if world == 1 else None,
                          "config": {"workload": "BASELINE.json configs[4] shape with the render / warp producers on the device",
                                     "parallelism": "one rank" if world == 1 else LEVEL_SHARDED if sharded else
                                     "replicas: keyframes dealt round-robin to ranks",
                                     "question": "latency of a live sequence (keyframes one after the other)" if sharded else
                                     "throughput of a recorded sequence (keyframes independent)"}}))
    if level is not None:
        level.close()
    if comm is not None:
        capi.rccl_comm_destroy(comm)
    if tex is not None:
        tex.close()
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c4", "stream", "e2e"],
                    help="c2 = BASELINE.json configs[1] (the headline line); c3 = configs[2] (960x540, 4096 candidates); c4 = the "
                         "per-rank share of configs[3] (848x480, 64 renders x 64 warps); stream = configs[4] shape on the local GPU")
    ap.add_argument("--keyframes", type=int, default=100)
    ap.add_argument("--shard", default="keyframes", choices=["keyframes", "level"],
                    help="stream / e2e configs with N > 1: keyframes = replicas, whole keyframes dealt round-robin (throughput of a recorded "
                         "sequence); level = every level's candidates shared by the ranks, one 8-byte all-reduce per level (latency of a live one)")
    ap.add_argument("--map", default="cloud", choices=["cloud", "mesh"], help="e2e config: coloured point cloud (nmi_prop_RENDER 4) or textured mesh (1)")
    ap.add_argument("--mesh-quads", default="60x40", help="e2e config, --map mesh: tessellation of the plane, NXxNY quads (60x40 = 4,800 triangles)")
    ap.add_argument("--no-graph", action="store_true", help="e2e config: enqueue the level's operations one by one instead of a HIP graph")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000,
                    help="timed steps (default 2000 searches = 1.46 M candidate evaluations, ~0.17 s on one MI355X)")
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--clock-warmup-ms", type=float, default=200.0,
                    help="untimed searches for this long before the warm-up steps, to bring the device to its sustained clock (0 = none)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-call-site", action="store_true", help="skip the call_site key (examples/relocalize_demo as a child process)")
    ap.add_argument("--blocking", action="store_true", help="one blocking search per step (SURVEY.md 8d's definition): the default with one GPU")
    ap.add_argument("--throughput", action="store_true",
                    help="steps enqueued back to back, every winner read back and checked in the timed region: the default with N > 1; "
                         "with one GPU it is measured as well and reported as throughput_evals_per_s")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="c2 / c3 / c4 with N > 1: weak = every rank scores a whole grid of its own renders (per-GPU work fixed); strong = "
                         "ONE grid, its cells dealt to the ranks (sharding.grid_shard), total work fixed")
    ap.add_argument("--streams", type=int, default=1,
                    help="searches kept in flight in throughput mode (one context + stream each); 2 fills the idle CUs of the "
                         "85%%-full last round (+5%%) but makes per-launch durations overlap, so the default stays 1")
    ap.add_argument("--allreduce-bucket", type=int, default=32,
                    help="N>1, throughput mode: winners of this many consecutive steps share one MAX all-reduce (1 = one per step)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal: run the N>1 code path (process group, async all-reduce per step) even with one rank")
    ap.add_argument("--all-on-device0", action="store_true",
                    help="rehearsal on a 1-GPU box: every rank uses cuda:0 (needs --backend gloo; numbers are meaningless)")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    ap.add_argument("--cpu-threads", type=int, default=16, help="cap on oracle threads for the cpu_baseline leg")
    ap.add_argument("--dry-launch", action="store_true",
                    help="start the ranks, have each report its rendezvous and the device it would select, touch no GPU, exit")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.gpus > 1 and "RANK" not in os.environ:  # plain `python3 bench.py --gpus N`: we are the launcher
        sys.exit(self_launch(args, sys.argv[1:]))
    if args.dry_launch:
        sys.exit(dry_launch(args))
    if args.config == "stream":
        return run_stream_config(args)
    if args.config == "e2e":
        return run_e2e_config(args)
    global WIDTH, HEIGHT, S_PER_RANK, WN
    workload_name = "BASELINE.json configs[1]: 640x480 frame, 729-pose grid (27 renders x 27 warps) per GPU"
    if args.config == "c3":
        WIDTH, HEIGHT, S_PER_RANK, WN = 960, 540, 64, 64
        workload_name = "BASELINE.json configs[2]: 960x540 frame, 4096-pose grid (64 renders x 64 warps) per GPU"
    elif args.config == "c4":
        WIDTH, HEIGHT, S_PER_RANK, WN = 848, 480, 64, 64
        workload_name = "BASELINE.json configs[3], one rank's share: 848x480 frames, 64 renders x 64 warps per GPU (x8 ranks = 32768)"

    import torch

    import orbslam2_nmi_amd as nmi
    from orbslam2_nmi_amd import synthetic as sy

    rank, local_rank, world, dist = init_dist(args)
    nmi.load_library()
    if args.blocking and args.throughput:
        sys.exit("bench.py: --blocking and --throughput exclude each other")
    # SURVEY.md 8(d) defines the metric on ONE blocking nmi_search_grid call including its 8-byte read-back: that is `value` on
    # one GPU (the back-to-back figure goes beside it as throughput_evals_per_s).  N > 1 measures throughput and says so.
    blocking_mode = args.blocking or (world == 1 and not args.throughput and dist is None)
    strong = args.scaling == "strong" and world > 1

    # ---- synthetic inputs, resident in HBM before any timing ------------------------------------------
    # rank 0 holds the renders of the planted scene; other ranks render a different scene (their candidates score
    # lower), so the global winner must come out of the collective as rank 0's centre cell.
    wl = sy.workload(WIDTH, HEIGHT, S_PER_RANK, WN, seed=1234)
    from orbslam2_nmi_amd import sharding
    if strong:
        # ONE grid for all ranks: rank r scores the block sharding.grid_shard deals it (render axis; the warp axis when there
        # are fewer renders than ranks), indices stay global, the winner comes out of the collective
        s_offset, s_local, w_offset, w_local = sharding.grid_shard(S_PER_RANK, WN, rank, world)
        S_total, WN_total = S_PER_RANK, WN
        rs = torch.from_numpy(np.ascontiguousarray(wl["render_stack"][s_offset:s_offset + s_local])).cuda()
        ws = torch.from_numpy(np.ascontiguousarray(wl["warp_stack"][w_offset:w_offset + w_local])).cuda()
        planted_global = wl["planted"]
    else:
        if rank > 0:
            other = sy.scene(WIDTH, HEIGHT, 5000 + rank)
            wl["render_stack"] = sy.render_stack(other, wl["s_counts"], bottom_up=True)
        S_total, WN_total = S_PER_RANK * world, WN
        s_offset, s_local, w_offset, w_local = S_PER_RANK * rank, S_PER_RANK, 0, WN
        rs = torch.from_numpy(wl["render_stack"]).cuda()
        ws = torch.from_numpy(wl["warp_stack"]).cuda()
        w_c, s_c = divmod(wl["planted"], S_PER_RANK)
        planted_global = w_c * S_total + s_c

    # Throughput mode keeps --streams searches in flight, each on its own context and non-default stream: a 729-candidate
    # search is 2.85 rounds of the chip, so the workgroups of the next search fill the CUs that the last round (and the
    # launch / completion latency) of the previous one leaves idle.  Every stream carries its kernels, the collective's
    # dependency and the read-backs.  --blocking uses one stream.
    n_streams = 1 if blocking_mode else max(1, args.streams)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    ctxs = []
    for st_ in streams:
        c = nmi.NmiContext(WIDTH, HEIGHT, bins=BINS, max_candidates=S_PER_RANK * WN)
        c.set_stream(st_.cuda_stream)
        ctxs.append(c)
    ctx, stream = ctxs[0], streams[0]
    torch.cuda.set_stream(stream)
    if dist is not None:
        # The search kernel holds one workgroup per CU for its whole run (LDS-limited).  Leave one CU per XCD to the
        # collective's kernel so that the 8-byte all-reduce of step i can run beside step i+1 instead of queueing for a
        # CU: 729 candidates are 3 rounds on 248 workgroups exactly as on 256, so this costs the search nothing.
        cus = ctx.info()["compute_units"]
        if cus >= 64:
            for c in ctxs:
                c.set_option(c.OPT_WORKGROUPS, cus - 8)
    key = torch.zeros(1, dtype=torch.int64, device="cuda")

    # Steps are enqueued back to back (throughput mode): step i's kernel writes its packed winner to keys[i]; with N>1
    # the MAX all-reduce of the winners (8 bytes per step, --allreduce-bucket steps per message) is issued asynchronously
    # and overlaps the following kernels.  All winners are read back and checked inside the timed region.  --blocking times the latency-bound form instead
    # (every step = one blocking nmi_search_grid call / kernel + collective + read-back before the next launch).
    n_slots = max(args.steps, args.warmup, 1)
    keys = torch.zeros(n_slots, dtype=torch.int64, device="cuda")
    keys_host = torch.zeros(n_slots, dtype=torch.int64).pin_memory()

    # one blocking nmi_search_grid call per step: the arguments are converted to C once (NmiContext.bind_search), the timed loop
    # makes the call and compares the winner -- the interpreter's per-call checks and conversions are harness, not library
    bound_search = ctx.bind_search(rs, ws) if dist is None else None

    def run_blocking(n):
        res = None
        for _ in range(n):
            if dist is None:
                res = bound_search()
            else:
                ctx.search_grid_shard(rs, s_offset, S_total, ws, key_out=key, blocking=False, w_offset=w_offset, wn_total=WN_total)
                sharding.allreduce_key(key, dist)
                res = nmi.key_unpack(int(key.item()))
            if res[0] != planted_global:
                sys.exit(f"rank {rank}: wrong winner {res} (expected index {planted_global})")

    region = {}

    bucket = max(1, args.allreduce_bucket)

    def run_async(n):
        works = []
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(streams[0])  # HIP events on the launch stream bracket the kernels of the timed region
        for i in range(n):
            slot = keys[i:i + 1]
            k = i % n_streams
            torch.cuda.set_stream(streams[k])
            ctxs[k].search_grid_shard(rs, s_offset, S_total, ws, key_out=slot, blocking=False, w_offset=w_offset, wn_total=WN_total)
            if dist is not None and ((i + 1) % bucket == 0 or i == n - 1):
                # The search's only exchange: MAX over ranks of the 8-byte packed winner.  The winners of `bucket`
                # consecutive (independent) steps travel in one message: a per-step collective costs ~10 us of the
                # compute stream per step (event hand-over between the streams), a bucketed one nothing measurable.
                lo = (i // bucket) * bucket
                for st_ in streams:
                    if st_ is not streams[k]:
                        streams[k].wait_stream(st_)
                works.append(dist.all_reduce(keys[lo:i + 1], op=dist.ReduceOp.MAX, async_op=True))  # RCCL over xGMI
        torch.cuda.set_stream(streams[0])
        ev1.record(streams[0])
        region["events"] = (ev0, ev1, n)
        for wk in works:
            wk.wait()
        for st_ in streams[1:]:
            streams[0].wait_stream(st_)
        keys_host[:n].copy_(keys[:n], non_blocking=True)  # one read-back of all winners into pinned memory ...
        torch.cuda.current_stream().synchronize()           # ... and the only wait of the region
        got = keys_host[:n].numpy().view(np.uint64)
        # packed key = score bits << 32 | (0xFFFFFFFF - index), 0 = no winner (nmi_key_unpack, vectorised)
        idx = np.where(got == 0, -1, 0xFFFFFFFF - (got & np.uint64(0xFFFFFFFF)).astype(np.int64))
        bad = got[idx != planted_global]
        if bad.size:
            sys.exit(f"rank {rank}: {bad.size} wrong winners, e.g. {nmi.key_unpack(int(bad[0]))} (expected index {planted_global})")

    run = run_blocking if blocking_mode else run_async

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Clock warm-up, untimed and reported in the line (clock_warmup_ms): an idle MI355X needs tens of milliseconds of work to
    # reach its sustained clock, and a timed region of K = 20 steps is 1.8 ms -- measured cold it reads 8 % low (kernel 86 us
    # instead of 79).  Same searches as the steps; then the W warm-up steps and the K timed steps as the contract says.
    if args.clock_warmup_ms > 0:
        # a step COUNT fixed by the arguments alone (every rank must issue the same collectives), sized from the
        # expected step time: 0.08 ms for 729 candidates at 640x480, scaled by candidates x pixels
        est_ms = 0.08 * (S_PER_RANK * WN * WIDTH * HEIGHT) / (729.0 * 640 * 480)
        todo = max(1, int(round(args.clock_warmup_ms / est_ms)))
        chunk = max(1, min(n_slots, 100))
        while todo > 0:
            run(min(chunk, todo))
            todo -= chunk
    run(args.warmup)
    sync_all()
    t0 = time.perf_counter()
    run(args.steps)
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # the other mode, for the record: with one GPU the timed region above was blocking calls, so the back-to-back form is
    # measured here over the same K steps (its HIP events give the kernel's average launch duration); otherwise 50 blocking calls
    throughput_ms_per_step = None
    if blocking_mode and world == 1 and dist is None:
        run_async(min(args.warmup, 20))
        torch.cuda.synchronize()
        tt = time.perf_counter()
        run_async(args.steps)
        torch.cuda.synchronize()
        throughput_ms_per_step = (time.perf_counter() - tt) / args.steps * 1e3
        blocking_call_ms = elapsed / args.steps * 1e3
    else:
        ctx.search_grid_shard(rs, s_offset, S_total, ws, key_out=key, blocking=True, w_offset=w_offset, wn_total=WN_total)
        torch.cuda.synchronize()
        tb = time.perf_counter()
        for _ in range(50):
            ctx.search_grid_shard(rs, s_offset, S_total, ws, key_out=key, blocking=True, w_offset=w_offset, wn_total=WN_total)
        blocking_call_ms = (time.perf_counter() - tb) / 50 * 1e3

    # ---- dominant kernel: live HIP-event timing on the launch stream ----------------------------------
    ctx.set_profiling(True)
    durs = []
    for _ in range(min(args.steps, 100)):
        ctx.search_grid_shard(rs, s_offset, S_total, ws, key_out=key, blocking=True, w_offset=w_offset, wn_total=WN_total)
        durs.append(ctx.last_kernel_ms())
    ctx.set_profiling(False)
    kernel_ms_exclusive = float(np.mean(durs))
    # Average launch duration over the timed region itself: HIP events recorded on the launch stream before the first and
    # after the last of the K back-to-back launches (one stream: the launches run one after the other, so this is the
    # kernel's duration plus the ~1 us hand-over between launches).  With several streams the launches overlap and a
    # per-launch duration is not defined: the exclusive figure (one launch at a time, events around it) is used.
    if n_streams == 1 and "events" in region:
        e0, e1, n_timed = region["events"]
        kernel_ms = e0.elapsed_time(e1) / n_timed
    else:
        kernel_ms = kernel_ms_exclusive

    evidence = rank_evidence(world, dist, local_rank)
    if rank == 0:
        evals_per_step = S_total * WN_total
        per_launch_evals = s_local * w_local
        pix = ctx.pix_status()["last_launch_ranges"]
        achieved = per_launch_evals * algorithmic_bytes_per_eval(WIDTH, HEIGHT) / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": f"pose-candidate NMI evals/sec ({WIDTH}x{HEIGHT}, 256 bins)" +
                      (", one blocking nmi_search_grid call per step incl. its 8-byte read-back (SURVEY.md 8d's definition)" if blocking_mode else
                       ", throughput mode: steps enqueued back to back, every winner read back and checked in the timed region") +
                      (f"; beside the W warm-up steps an untimed clock warm-up of clock_warmup_ms = {args.clock_warmup_ms:g} ms precedes the timed region"
                       if args.clock_warmup_ms > 0 else ""),
            "extra_warmup": "clock_warmup_ms",
            "value": evals_per_step * args.steps / elapsed,
            # one blocking call of THIS rank's block incl. its 8-byte read-back (== value with one GPU and no --throughput)
            "blocking_call_evals_per_s": per_launch_evals / (blocking_call_ms * 1e-3),
            "throughput_evals_per_s": (evals_per_step / (throughput_ms_per_step * 1e-3) if throughput_ms_per_step else
                                       (None if blocking_mode else evals_per_step * args.steps / elapsed)),
            "unit": "evals/s",
            "n_gpus": world,
            **evidence,
            "steps": args.steps,
            "warmup": args.warmup,
            "clock_warmup_ms": args.clock_warmup_ms,
            "ms_per_step": elapsed / args.steps * 1e3,
            "step_mode": ("blocking call per step" + (": nmi_search_grid through NmiContext.bind_search (arguments converted to C once), winner checked every step"
                                                       if dist is None else "")) if blocking_mode else
                         "steps enqueued back to back; every step's winner read back and checked inside the timed region",
            "blocking_call_ms": blocking_call_ms,
            "streams": n_streams,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "u8", "dtype_detail": "u8 pixels -> u32 histogram counts (integer, exact); f32 entropy terms and score",
            "data": "synthetic",
            "config": {"workload": (workload_name.replace(" per GPU", " in all, its cells dealt to the ranks") if strong else workload_name) +
                                   ", 256-bin NMI (SUC), render axis sharded by rank",
                       "width": WIDTH, "height": HEIGHT, "renders_per_gpu": s_local, "warps_per_gpu": w_local, "warps": WN_total,
                       "candidates_per_gpu": per_launch_evals, "candidates_total": evals_per_step, "bins": BINS,
                       "collective": "none" if dist is None else
                       ("8-byte MAX all-reduce per step" if blocking_mode or bucket == 1 else
                        f"MAX all-reduce of the 8-byte winners, {bucket} steps per message")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": load_pmc_traffic() if args.config == "c2" else None,
                         "lds": load_pmc_lds() if args.config == "c2" else None,
                         "static": static_profile_stamp() if args.config == "c2" else None,
                         "note": "frac is the contract's ALGORITHMIC fraction (2*W*H+4 bytes per evaluation / kernel time / 8 TB/s); "
                                 "the inputs live in L2 / Infinity Cache (traffic) and the kernel is bound by LDS atomic issue, see lds",
                         "kernel": f"nmi_pix_kernel ({pix} pixel ranges per candidate)" if pix else "nmi_grid_kernel", "kernel_ms": kernel_ms, "kernel_ms_exclusive": kernel_ms_exclusive,
                         "algorithmic_bytes_per_launch": per_launch_evals * algorithmic_bytes_per_eval(WIDTH, HEIGHT)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl, args.cpu_budget, args.cpu_threads)
        if world == 1 and not args.no_call_site and args.config == "c2":
            out["call_site"] = call_site_rate()
        print(json.dumps(out))
    for c in ctxs:
        c.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
