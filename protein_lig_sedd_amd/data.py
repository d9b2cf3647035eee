"""Host side of the data path that feeds the hot path — the parts that need no downloads:

  * fixed-pad batches: what the reference's tokenizer calls produce (`padding='max_length', max_length=512 | 128,
    truncation=True`, data/tokenize.py:14,25) from already-tokenised id lists;
  * variable-length packing for `DDiTBlock.forward(seqlens=...)` (model/transformer.py:173-179): several sequences per
    padded row, `seqlens` in the form the reference's hook expects (leading 0, `cumsum` = flash-attn's cu_seqlens);
  * the per-rank index lists of `torch.utils.data.DistributedSampler` (data/process_full_plinder.py:396-398), so every
    rank of the RCCL data-parallel trainer draws exactly the shard the reference's DDP loaders would.

The tokenizers themselves (ESM-2 / MoLFormer, HuggingFace downloads) are out of scope: token ids come in as lists."""
import math

import torch


def pad_to_fixed(token_lists, max_length, pad_id, truncation=True):
    """-> (input_ids int64 [B, max_length], attention_mask int64 [B, max_length]) like a HF tokenizer called with
    padding='max_length', truncation=True (right padding, right truncation)."""
    B = len(token_lists)
    ids = torch.full((B, max_length), int(pad_id), dtype=torch.int64)
    mask = torch.zeros(B, max_length, dtype=torch.int64)
    for i, t in enumerate(token_lists):
        t = list(t)
        if len(t) > max_length:
            if not truncation:
                raise ValueError(f"sequence {i} has {len(t)} tokens > max_length {max_length}")
            t = t[:max_length]
        ids[i, :len(t)] = torch.as_tensor(t, dtype=torch.int64)
        mask[i, :len(t)] = 1
    return ids, mask


def pack_sequences(token_lists, row_length, pad_id):
    """Greedy first-fit packing of variable-length sequences into rows of `row_length` tokens.
    -> (tokens int64 [R, row_length], seqlens int32 [1 + nseg]) where seqlens = [0, len_0, len_1, ...] walks the flattened
    (R x row_length) token axis: every sequence is one segment, every run of padding at the end of a row is a segment
    of its own (so padding never attends to data), and seqlens.sum() == R * row_length — the form
    `DDiTBlock.forward(..., seqlens=seqlens)` / flash-attn's cu_seqlens = seqlens.cumsum(-1) expects."""
    rows, fill = [], []
    for t in token_lists:
        t = list(t)
        if len(t) > row_length:
            raise ValueError(f"a sequence of {len(t)} tokens does not fit a row of {row_length}")
        for r in range(len(rows)):
            if fill[r] + len(t) <= row_length:
                rows[r].append(t)
                fill[r] += len(t)
                break
        else:
            rows.append([t])
            fill.append(len(t))
    tokens = torch.full((max(len(rows), 1), row_length), int(pad_id), dtype=torch.int64)
    seqlens = [0]
    for r, seqs in enumerate(rows):
        pos = 0
        for t in seqs:
            tokens[r, pos:pos + len(t)] = torch.as_tensor(t, dtype=torch.int64)
            seqlens.append(len(t))
            pos += len(t)
        if pos < row_length:
            seqlens.append(row_length - pos)
    if not rows:
        seqlens.append(row_length)
    return tokens, torch.tensor(seqlens, dtype=torch.int32)


def distributed_indices(n, rank, world, shuffle=True, seed=0, epoch=0, drop_last=False):
    """The index list `torch.utils.data.DistributedSampler(dataset_of_n, num_replicas=world, rank=rank, shuffle, seed,
    drop_last)` yields after `set_epoch(epoch)` — same permutation, same padding by wrap-around, same strided split."""
    if not 0 <= rank < world:
        raise ValueError(f"Invalid rank {rank}, rank should be in the interval [0, {world - 1}]")
    if drop_last and n % world != 0:
        num = math.ceil((n - world) / world)
    else:
        num = math.ceil(n / world)
    total = num * world
    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed + epoch)
        idx = torch.randperm(n, generator=g).tolist()
    else:
        idx = list(range(n))
    if not drop_last:
        pad = total - len(idx)
        if pad <= len(idx):
            idx += idx[:pad]
        else:
            idx += (idx * math.ceil(pad / len(idx)))[:pad]
    else:
        idx = idx[:total]
    return idx[rank:total:world]
