"""Multi-GPU form of the search (new design; the reference is single-GPU).

The candidate grid shards along the render (synth) axis: rank r owns a contiguous block of renders and the whole
warp stack, scores its block with the HIP kernel, and the ranks exchange one packed 64-bit key with a MAX
all-reduce (torch.distributed: backend "nccl" is RCCL on ROCm; "gloo" in the CPU tests).  The key
(nmi_key_pack, include/nmi_hip.h) orders candidates exactly like helperFunctions::find_max_elements
(helperFunctions.cpp:50-103): larger score first, then lower global linear index.
"""
import numpy as np

from . import capi


def render_shard(s_total, rank, world):
    """Contiguous block [offset, offset + count) of the render axis for `rank`; blocks differ by at most one."""
    if not (0 <= rank < world) or s_total < 0:
        raise ValueError("bad shard request")
    base, extra = divmod(s_total, world)
    count = base + (1 if rank < extra else 0)
    offset = rank * base + min(rank, extra)
    return offset, count


def grid_shard(s_total, wn_total, rank, world):
    """Block of the S x Wn grid for `rank`: (s_offset, s_count, w_offset, w_count).  The render axis is sharded when it has at
    least one render per rank (no render is then replicated); otherwise the warp axis, every rank holding all
    renders.  With fewer than `world` cells on both axes the render axis is used and the surplus ranks get empty blocks."""
    if s_total >= world or wn_total < world:
        off, cnt = render_shard(s_total, rank, world)
        return off, cnt, 0, wn_total
    off, cnt = render_shard(wn_total, rank, world)
    return 0, s_total, off, cnt


def global_index(w, s_global, s_total):
    """Linear index of candidate (warp w, render s_global) in the unsharded rating table [Wn][S_total]."""
    return w * s_total + s_global


def allreduce_key(key_tensor, dist, group=None):
    """MAX all-reduce of the one-element int64 key tensor in place (keys are < 2^63 for non-negative scores)."""
    dist.all_reduce(key_tensor, op=dist.ReduceOp.MAX, group=group)
    return key_tensor


def sharded_search(ctx, render_shard_stack, s_offset, s_total, warp_stack, key_tensor, dist=None, group=None):
    """One rank's step: HIP kernel on the local shard -> key on the device -> all-reduce -> (index, score) on host.

    ctx must run on the current torch stream, and that stream must not be the legacy default one
    (s = torch.cuda.Stream(); torch.cuda.set_stream(s); ctx.set_stream(s.cuda_stream)), so that the collective and
    the read-back are ordered after the kernel."""
    ctx.search_grid_shard(render_shard_stack, s_offset, s_total, warp_stack, key_out=key_tensor, blocking=False)
    if dist is not None:
        allreduce_key(key_tensor, dist, group)
    return capi.key_unpack(int(key_tensor.item()))


def local_key_from_ratings(ratings_local, s_offset, s_total, w_offset=0):
    """Host restatement of what the kernel's atomicMax computes, for ratings [Wn_local][S_local] of one block.
    Used by the CPU (gloo) tests of the collective logic."""
    r = np.asarray(ratings_local, np.float32)
    best = 0
    wn, s_local = r.shape
    for w in range(wn):
        for s in range(s_local):
            k = capi.key_pack(float(r[w, s]), global_index(w_offset + w, s_offset + s, s_total))
            best = max(best, k)
    return best


# ---- keyframe streams (BASELINE.json configs[4]) -------------------------------------------------------------------------
# The streamed loop (src/Tracking.cc:2088-2130: <= 4 coarse-to-fine levels per keyframe) is sequential inside a keyframe
# -- level i+1 is centred on level i's winner -- but keyframes of a recorded sequence are independent searches.  With N
# ranks the keyframes are dealt round-robin: rank r takes keyframes r, r + N, ...  ("replicas": no collective on the data
# path; every rank holds the map and streams its own frames and render stacks).  Only the tiny per-keyframe results are
# gathered at the end, in keyframe order.

def keyframe_share(n_keyframes, rank, world):
    """Keyframes of rank `rank`: r, r + world, r + 2 world, ..."""
    if not (0 <= rank < world) or n_keyframes < 0:
        raise ValueError("bad keyframe share request")
    return list(range(rank, n_keyframes, world))


def new_keyframe_table(n_keyframes, levels):
    """Result table [n_keyframes, levels, 2] = (best index + 1, float32 score bits), zero = not computed on this rank."""
    return np.zeros((n_keyframes, levels, 2), np.int64)


def store_keyframe_result(table, kf, lvl, best_index, best_score):
    table[kf, lvl, 0] = int(best_index) + 1  # -1 ("no candidate") -> 0
    table[kf, lvl, 1] = int(np.float32(best_score).view(np.uint32))


def gather_keyframe_table(table, world, dist=None, device="cpu", group=None):
    """One SUM all-reduce of the result table (each row is written by exactly one rank) -> the complete table on every
    rank, indices back in their -1-based form."""
    import torch
    if dist is not None and world > 1:
        t = torch.from_numpy(table).to(device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        table = t.cpu().numpy()
    table = table.copy()
    table[..., 0] -= 1
    return table


def run_keyframes(n_keyframes, levels, rank, world, process_keyframe, dist=None, device="cpu", group=None):
    """Every rank runs process_keyframe(kf) -> [(best_index, best_score)] * levels on its share; the results of all
    keyframes come back on every rank as an int64 array [n_keyframes, levels, 2] = (index, float32 score bits), in
    keyframe order."""
    table = new_keyframe_table(n_keyframes, levels)
    for kf in keyframe_share(n_keyframes, rank, world):
        res = process_keyframe(kf)
        if len(res) != levels:
            raise ValueError("process_keyframe must return one (index, score) per level")
        for lvl, (idx, score) in enumerate(res):
            store_keyframe_result(table, kf, lvl, idx, score)
    return gather_keyframe_table(table, world, dist, device, group)


def unpack_keyframe_results(table):
    """[n_keyframes, levels, 2] -> list over keyframes of [(index, float32 score)] per level."""
    return [[(int(i), np.array([b], np.uint32).view(np.float32)[0]) for i, b in kf] for kf in table]


# ---- level-sharded form (the latency form of BASELINE.json configs[4]) ---------------------------------------------------
# Keyframes of a LIVE sequence are not independent: each search is seeded from the drift accumulated since the previous NMI
# fix (src/Tracking.cc:2001-2053, accumulated at :651-661, reset at :603-607) inside the sequential Track() (:598-616).  A live
# level can therefore only be made faster by sharing ITS candidates: rank r renders / receives only its block of views
# (grid_shard: render axis, or the warp axis when there are fewer views than ranks), makes its warps locally, scores its
# cells with global indices, and the ranks exchange one packed 8-byte key per level (MAX all-reduce); every rank then
# holds the level's winner and derives the next level from it.  The replica form above answers "how fast is a RECORDED
# sequence processed", this one "how long does one live level take".

def sharded_level(run_block, s_total, wn_total, rank, world, dist=None, device="cpu", group=None):
    """One level over `world` ranks.  run_block(s_offset, s_count, w_offset, w_count) scores this rank's block and returns
    (global linear index, score) of its winner, (-1, 0) for none; it is not called for an empty block.  The keys are
    MAX-all-reduced through torch.distributed (`dist`; None = single rank).  -> (index, score) of the level's winner.
    (With RCCL the product does the same inside nmi_level_run_rccl / nmi_stream_submit_block; this is the form for callers
    that own the collective, and the one the CPU tests drive with the oracle as the scorer.)"""
    import torch
    so, sc, wo, wc = grid_shard(s_total, wn_total, rank, world)
    idx, score = run_block(so, sc, wo, wc) if sc and wc else (-1, 0.0)
    key = capi.key_pack(float(score), int(idx)) if idx >= 0 else 0
    if dist is not None and world > 1:
        t = torch.tensor([key], dtype=torch.int64, device=device)
        allreduce_key(t, dist, group)
        key = int(t.item())
    return capi.key_unpack(key)


def run_keyframes_level_sharded(n_keyframes, levels, process_level, table=None):
    """Keyframes one after the other, every level on all ranks: process_level(kf, lvl, previous winners of this keyframe)
    -> (index, score) of the level's winner (already reduced over the ranks, e.g. by sharded_level).  -> the same
    [n_keyframes, levels, 2] table run_keyframes returns (no gather needed: every rank has every winner)."""
    table = new_keyframe_table(n_keyframes, levels) if table is None else table
    for kf in range(n_keyframes):
        won = []
        for lvl in range(levels):
            idx, score = process_level(kf, lvl, won)
            won.append((idx, score))
            store_keyframe_result(table, kf, lvl, idx, score)
    table = table.copy()
    table[..., 0] -= 1
    return table
