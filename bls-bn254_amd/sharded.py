"""Sharding of a verify batch over one-process-per-GPU ranks (torch.distributed; backend "nccl" is
RCCL over xGMI on MI355X, "gloo" in the CPU tests).

Every (pk, msg, sig) tuple is independent (mirrors the per-term independence of multi_miller_loop,
pairings.rs:819-824), so rank g verifies the contiguous range [g*n/G, (g+1)*n/G) with no data-path
collective.  The only exchange is the validity bitmap: each rank places its bits into a zeroed
full-length int32 word array and the arrays are summed (RCCL has no bitwise OR; the bit sets are
disjoint, so SUM == OR and no carry can occur).  1 MiB at n = 8M: latency-bound, not link-bound.
"""
import numpy as np


def shard_range(n, rank, world):
    return (n * rank) // world, (n * (rank + 1)) // world


def bitmap_bytes_to_bits(bm, m, torch):
    """uint8 LSB-first bitmap tensor -> 0/1 uint8 tensor of length m (on bm's device)."""
    shifts = torch.arange(8, device=bm.device, dtype=torch.uint8)
    return ((bm[:, None] >> shifts[None, :]) & 1).reshape(-1)[:m]


def bits_to_words(bits, torch):
    """0/1 tensor (length multiple of 32) -> int32 words, bit i of the batch = bit (i & 31) of word i >> 5."""
    w = (bits.reshape(-1, 32).to(torch.int64) << torch.arange(32, device=bits.device, dtype=torch.int64)[None, :]).sum(dim=1)
    return w.to(torch.int32)       # wraps the sign bit, fine for a bit container


def words_to_bitmap_bytes(words, n):
    """int32 words (host numpy) -> ceil(n/8) bytes LSB-first."""
    b = np.ascontiguousarray(words).view(np.uint8)
    return b[:(n + 7) // 8].tobytes()


def allreduce_bitmap(local_bm, lo, m, n, dist, torch, group=None):
    """local_bm: uint8 tensor, LSB-first bitmap of this rank's m tuples (global indices lo..lo+m).
    Returns the full-batch bitmap as an int32 word tensor, identical on every rank."""
    nwords = (n + 31) // 32
    if (lo % 32 == 0 and m % 32 == 0 and local_bm.numel() * 8 >= m and local_bm.is_contiguous()
            and local_bm.storage_offset() % 4 == 0):
        # word-aligned shard (every power-of-two batch): the local bytes ARE the shard's words
        words = torch.zeros(nwords, dtype=torch.int32, device=local_bm.device)
        if m:
            words[lo // 32:(lo + m) // 32] = local_bm[:m // 8].view(torch.int32)
    else:
        bits = torch.zeros(nwords * 32, dtype=torch.uint8, device=local_bm.device)
        if m:
            bits[lo:lo + m] = bitmap_bytes_to_bits(local_bm, m, torch)
        words = bits_to_words(bits, torch)
    if dist is not None and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(words, op=dist.ReduceOp.SUM, group=group)
    return words


def verify_batch_sharded(verify_local, n, rank, world, dist, torch, device, group=None):
    """verify_local(lo, hi) -> uint8 bitmap tensor (on `device`) for global tuples lo..hi.
    Returns the full bitmap as int32 words on `device`, identical on every rank."""
    lo, hi = shard_range(n, rank, world)
    local = verify_local(lo, hi) if hi > lo else torch.zeros(0, dtype=torch.uint8, device=device)
    return allreduce_bitmap(local, lo, hi - lo, n, dist, torch, group)


def aggregate_verify_sharded(partial_local, finish, n, rank, world, dist, torch, device, group=None):
    """Aggregate verify over ranks: rank g computes the Fp12 partial product of its contiguous
    share, the 384-byte partials are all-gathered (Fp12 multiplication is not an RCCL reduce op), every rank
    finishes with the single final exponentiation.
      partial_local(lo, hi) -> (384 bytes, all_pks_ok: bool)     finish(partials_bytes, k) -> bool"""
    lo, hi = shard_range(n, rank, world)
    ml, ok = partial_local(lo, hi)
    mine = torch.frombuffer(bytearray(ml), dtype=torch.uint8).to(device)
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    if dist is not None and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        allp = torch.cat(parts)
    else:
        allp = mine
    k = allp.numel() // 384
    return bool(flag.item()) and n > 0 and finish(bytes(allp.cpu().numpy()), k)
