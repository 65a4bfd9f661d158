"""world_size-2 (and 3) CPU tests of the multi-GPU logic over gloo: render-axis sharding, packed keys, MAX all-reduce.
The per-rank scoring is done by the CPU oracle here (no GPU in this tier); the key packing / unpacking is the product's
C code (nmi_key_pack / nmi_key_unpack) and the collective pattern is the one bench.py runs over RCCL."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT
from orbslam2_nmi_amd import sharding


def test_render_shard_partitions():
    for s_total in (0, 1, 5, 27, 64, 512):
        for world in (1, 2, 3, 8):
            blocks = [sharding.render_shard(s_total, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and sum(c for _, c in blocks) == s_total
            for (o1, c1), (o2, _) in zip(blocks, blocks[1:]):
                assert o1 + c1 == o2
            assert max(c for _, c in blocks) - min(c for _, c in blocks) <= 1
    assert sharding.render_shard(512, 3, 8) == (192, 64)  # BASELINE.json config 4: 64 renders per rank
    with pytest.raises(ValueError):
        sharding.render_shard(8, 8, 8)


def test_grid_shard_picks_the_axis():
    # enough renders: render axis, all warps everywhere
    assert sharding.grid_shard(27, 27, 3, 8) == (12, 3, 0, 27)
    assert sharding.grid_shard(512, 64, 3, 8) == (192, 64, 0, 64)
    # fewer renders than ranks: warp axis, all renders everywhere
    assert sharding.grid_shard(1, 27, 2, 8) == (0, 1, 8, 4)
    blocks = [sharding.grid_shard(3, 27, r, 8) for r in range(8)]
    assert all(b[:2] == (0, 3) for b in blocks) and sum(b[3] for b in blocks) == 27 and blocks[0][2] == 0
    for (_, _, o1, c1), (_, _, o2, _) in zip(blocks, blocks[1:]):
        assert o1 + c1 == o2
    # tiny grid on both axes: render axis, surplus ranks empty
    assert [sharding.grid_shard(2, 3, r, 4)[1] for r in range(4)] == [1, 1, 0, 0]


WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, os.environ["NMI_ROOT"])
    from oracle import binding as oc
    from orbslam2_nmi_amd import capi, sharding, synthetic as sy
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    case = os.environ["NMI_CASE"]
    wl = sy.workload(64, 48, 8, 6, seed=9)
    rs, ws = wl["render_stack"], wl["warp_stack"]
    if case == "ties":       # the same best render on both sides of the shard boundary: lowest global index must win
        rs = rs.copy(); rs[7] = rs[wl["planted"] % 8]; rs[0] = rs[wl["planted"] % 8]
    if case == "allzero":    # constant renders: every score 0 -> winner index 0 (first exact zero)
        rs = np.full_like(rs, 255)
    if case == "warpaxis":   # fewer renders than ranks: the warp axis is sharded instead (SURVEY.md 8e)
        rs = rs[:world - 1] if world > 1 else rs[:1]
    if case == "c4shape":    # BASELINE.json configs[3]: 512 renders (8^3) x 64 warps (4^3) = 32,768 candidates over 8 ranks, tiny images
        wl = sy.workload(24, 16, 512, 64, seed=11)
        rs, ws = wl["render_stack"], wl["warp_stack"]
    S, Wn = rs.shape[0], ws.shape[0]
    off, cnt, woff, wcnt = sharding.grid_shard(S, Wn, rank, world)
    assert (case == "warpaxis") == (wcnt != Wn or world == 1)
    if case == "c4shape":
        assert (off, cnt, woff, wcnt) == (64 * rank, 64, 0, 64) and world == 8
        assert sharding.keyframe_share(100, rank, world) == list(range(rank, 100, 8))   # configs[4] dealt over the same ranks
    if cnt and wcnt:
        local, _, _ = oc.search_grid(rs[off:off + cnt], ws[woff:woff + wcnt])
    else:
        local = np.zeros((wcnt, cnt), np.float32)
    key = torch.tensor([sharding.local_key_from_ratings(local, off, S, woff)], dtype=torch.int64)
    sharding.allreduce_key(key, dist)
    got = capi.key_unpack(int(key.item()))
    if case == "c4shape" and rank:   # the unsharded table is computed once, on rank 0
        ref = [None]
        dist.broadcast_object_list(ref, src=0)
        full, idx, best = ref[0]
    else:
        full, idx, best = oc.search_grid(rs, ws, threads=1 if case != "c4shape" else 2)
        if case == "c4shape":
            dist.broadcast_object_list([(full, idx, best)], src=0)
    assert got == (idx, best), (rank, got, idx, best)
    # the gathered blocks reassemble the full rating table (optional all-gather of SURVEY.md 8e)
    parts = [None] * world
    dist.all_gather_object(parts, (off, woff, local))
    parts = sorted(parts, key=lambda p: (p[1], p[0]))
    table = np.concatenate([p[2] for p in parts], axis=0 if case == "warpaxis" else 1)
    assert (table == full).all()
    dist.barrier(); dist.destroy_process_group()
    print("rank", rank, "ok", got)
""")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,case", [(2, "planted"), (2, "ties"), (2, "allzero"), (3, "planted"), (3, "warpaxis"), (8, "c4shape")])
def test_sharded_argmax_over_gloo(world, case, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   NMI_ROOT=ROOT, NMI_CASE=case, OMP_NUM_THREADS="2" if world < 8 else "1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=180)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode())
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\\n{out}"
        assert f"rank {rank} ok" in out


def test_keyframe_share_is_a_partition():
    for n in (0, 1, 7, 100):
        for world in (1, 2, 3, 8):
            shares = [sharding.keyframe_share(n, r, world) for r in range(world)]
            assert sorted(k for s in shares for k in s) == list(range(n))
            assert max(map(len, shares)) - min(map(len, shares)) <= 1
    assert sharding.keyframe_share(100, 3, 8)[:3] == [3, 11, 19]   # BASELINE.json configs[4]: 100 keyframes over 8 ranks
    with pytest.raises(ValueError):
        sharding.keyframe_share(10, 2, 2)


KEYFRAME_WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, os.environ["NMI_ROOT"])
    from oracle import binding as oc
    from orbslam2_nmi_amd import sharding, synthetic as sy
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_kf, levels = 7, 3
    calls = []
    def process_keyframe(kf):   # the per-keyframe coarse-to-fine loop (Tracking.cc:2088-2130) with the CPU oracle as the scorer
        calls.append(kf)
        B = sy.scene(64, 48, 300 + kf)
        F = sy.camera_frame(B, 400 + kf)
        out = []
        for lvl in range(levels):
            rs = sy.render_stack(B, (2, 2, 1), shift_px=max(1, 4 >> lvl))
            ws = sy.warp_stack(F, (1, 1, 3), tuple(s / 2 ** lvl for s in (0.02, 0.02, 0.05)))
            _, idx, best = oc.search_grid(rs, ws, render_bottom_up=False)
            out.append((idx, best))
        return out
    table = sharding.run_keyframes(n_kf, levels, rank, world, process_keyframe, dist)
    assert calls == sharding.keyframe_share(n_kf, rank, world), (rank, calls)          # only this rank's share was computed
    got = sharding.unpack_keyframe_results(table)
    calls.clear()
    expect = [process_keyframe(kf) for kf in range(n_kf)]                               # the whole sequence on one rank
    assert got == [[(i, np.float32(s)) for i, s in kf] for kf in expect], (rank, got[:2], expect[:2])
    dist.barrier(); dist.destroy_process_group()
    print("rank", rank, "ok", len(got))
""")


@pytest.mark.parametrize("world", [2, 3])
def test_keyframe_stream_orchestration_over_gloo(world, tmp_path):
    """The multi-rank form of BASELINE.json configs[4] (bench.py --config stream|e2e with WORLD_SIZE > 1): keyframes dealt
    round-robin, every rank runs the per-keyframe level loop on its share, results gathered in keyframe order and equal to
    the single-rank sequence."""
    script = tmp_path / "kf_worker.py"
    script.write_text(KEYFRAME_WORKER)
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   NMI_ROOT=ROOT, OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode())
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\\n{out}"
        assert f"rank {rank} ok" in out


LEVEL_WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, os.environ["NMI_ROOT"])
    from oracle import binding as oc
    from orbslam2_nmi_amd import sharding, synthetic as sy
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    case = os.environ["NMI_CASE"]
    s_counts, w_counts = ((3, 3, 3), (3, 3, 3)) if case == "views27" else ((2, 1, 1), (3, 3, 1))   # 27 x 27, or 2 views x 9 warps
    n_kf, levels = 2, 3
    blocks = []
    def level_inputs(kf, lvl, won):
        # level l+1 depends on level l's winner (Tracking.cc:2088-2130: the next grid is centred on it): here the winner's
        # index picks the scene shift of the next level, so a wrong or rank-dependent winner derails everything after it
        nudge = sum(i for i, _ in won) % 3
        B = sy.scene(48, 32, 700 + kf)
        F = sy.camera_frame(B, 800 + kf)
        rs = sy.render_stack(B, s_counts, shift_px=max(1, 3 >> lvl) + nudge)
        ws = sy.warp_stack(F, w_counts, tuple(s / 2 ** lvl for s in (0.02, 0.02, 0.05)))
        return rs, ws
    def process_level(kf, lvl, won):
        rs, ws = level_inputs(kf, lvl, won)
        S, Wn = rs.shape[0], ws.shape[0]
        def run_block(so, sc, wo, wc):      # this rank sees ONLY its block of views / warps
            blocks.append((so, sc, wo, wc))
            local, _, _ = oc.search_grid(rs[so:so + sc], ws[wo:wo + wc], render_bottom_up=False)
            return sharding.capi.key_unpack(sharding.local_key_from_ratings(local, so, S, wo))
        return sharding.sharded_level(run_block, S, Wn, rank, world, dist)
    table = sharding.run_keyframes_level_sharded(n_kf, levels, process_level)
    S_total = 27 if case == "views27" else 2
    if case == "views27":
        assert all(b == sharding.grid_shard(27, 27, rank, world) and b[3] == 27 for b in blocks), blocks      # render axis
    else:
        assert all(b[:2] == (0, 2) and b[3] < 9 for b in blocks) or world == 1, blocks                         # fewer views than ranks: warp axis
    # the whole sequence on one rank, unsharded
    def whole_level(kf, lvl, won):
        rs, ws = level_inputs(kf, lvl, won)
        _, idx, best = oc.search_grid(rs, ws, render_bottom_up=False)
        return idx, best
    expect = sharding.run_keyframes_level_sharded(n_kf, levels, whole_level)
    assert (table == expect).all(), (rank, table.tolist(), expect.tolist())
    dist.barrier(); dist.destroy_process_group()
    print("rank", rank, "ok", table[..., 0].reshape(-1).tolist())
""")


@pytest.mark.parametrize("world,case", [(2, "views27"), (3, "views27"), (3, "warpaxis")])
def test_level_sharded_orchestration_over_gloo(world, case, tmp_path):
    """The level-sharded multi-rank form (bench.py --config stream|e2e --shard level; nmi_level_run_rccl /
    nmi_stream_submit_block on the GPU): every rank scores only its block of each level -- 27 views over 2 / 3 ranks on the
    render axis; 2 views over 3 ranks falls to the warp axis -- one 8-byte MAX all-reduce per level, dependent levels, and
    the sequence of winners equals the unsharded one on every rank.  The oracle is the scorer here."""
    script = tmp_path / "level_worker.py"
    script.write_text(LEVEL_WORKER)
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   NMI_ROOT=ROOT, NMI_CASE=case, OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode())
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\\n{out}"
        assert f"rank {rank} ok" in out
