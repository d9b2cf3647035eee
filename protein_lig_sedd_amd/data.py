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


# ------------------------------------------------------------------------------- PLINDER list dataset + collate
class ProteinLigandDataset:
    """data/process_full_plinder.py:31-115 (`ImprovedProteinLigandDataset`): a list of per-complex dicts, as the PLINDER
    preprocessing leaves it in a `.pt` file — `protein_seq` / `ligand_smiles` (strings; here also already-tokenised id
    lists, since the HF tokenizers the reference calls on the fly need downloads) and optional `protein_coords`,
    `ligand_coords`, `binding_value`, `interaction_mask`.  `__getitem__` returns the reference's processed sample: float
    masks of the fixed lengths with ones over min(len, max_len), the sequences untouched, optional fields as fp32 tensors.

    data: a path (loaded with torch.load(weights_only=True): files written by this package or by torch.save of plain
    lists / dicts / strings / tensors load; nothing is unpickled) or an in-memory list."""

    def __init__(self, data, max_protein_len=512, max_ligand_len=128):
        self.max_protein_len, self.max_ligand_len = max_protein_len, max_ligand_len
        if isinstance(data, (list, tuple)):
            self.data = list(data)
        else:
            import os
            if not os.path.exists(str(data)):
                raise FileNotFoundError(f"Data file not found: {data}")
            self.data = torch.load(str(data), weights_only=True)

    def __len__(self):
        return len(self.data)

    def __getitem__(self, idx):
        sample = self.data[idx]
        out = {}
        for key, mkey, mlen in (("protein_seq", "protein_mask", self.max_protein_len),
                                ("ligand_smiles", "ligand_mask", self.max_ligand_len)):
            if sample.get(key) is not None:
                mask = torch.zeros(mlen, dtype=torch.float)
                mask[:min(len(sample[key]), mlen)] = 1.0
                out[mkey] = mask
                out[key] = sample[key]
        for key in ("protein_coords", "ligand_coords", "binding_value", "interaction_mask"):
            if sample.get(key) is not None:
                out[key] = torch.as_tensor(sample[key], dtype=torch.float)
        return out


def protein_ligand_collate_fn(batch, protein_pad_id=1, ligand_pad_id=2, token_ids=None):
    """data/process_full_plinder.py:118-243 (`improved_protein_ligand_collate_fn`): masks stacked to the longest mask of
    the batch, sequences as lists, optional coordinates / binding values / interaction masks zero-padded — same keys,
    shapes and dtypes.  Extension for the hot path: when the sequences are token-id lists (or `token_ids=True`) the batch
    also carries `prot_tokens` / `lig_tokens`, int64 [B, mask length], right-padded / truncated like the reference's
    tokenizer calls (data/tokenize.py:14,25) — the batch dict CrossAttTrainer.train_step takes."""
    B = len(batch)
    has_p = all("protein_seq" in s for s in batch)
    has_l = all("ligand_smiles" in s for s in batch)
    out = {}
    Lp = max(len(s["protein_mask"]) for s in batch) if has_p else 0
    Ll = max(len(s["ligand_mask"]) for s in batch) if has_l else 0
    if has_p:
        pm = torch.zeros(B, Lp, dtype=torch.float)
        for i, s in enumerate(batch):
            pm[i, :len(s["protein_mask"])] = s["protein_mask"]
        out["protein_mask"], out["protein_seq"] = pm, [s["protein_seq"] for s in batch]
    if has_l:
        lm = torch.zeros(B, Ll, dtype=torch.float)
        for i, s in enumerate(batch):
            lm[i, :len(s["ligand_mask"])] = s["ligand_mask"]
        out["ligand_mask"], out["ligand_smiles"] = lm, [s["ligand_smiles"] for s in batch]
    if all("protein_coords" in s for s in batch):
        c0 = batch[0]["protein_coords"]
        pc = torch.zeros((B, Lp, 3, 3) if c0.dim() == 3 else (B, Lp, 3), dtype=torch.float)
        for i, s in enumerate(batch):
            pc[i, :s["protein_coords"].shape[0]] = s["protein_coords"]
        out["protein_coords"] = pc
    if all("ligand_coords" in s for s in batch):
        na = max(s["ligand_coords"].shape[0] for s in batch)
        lc = torch.zeros(B, na, 3, dtype=torch.float)
        for i, s in enumerate(batch):
            lc[i, :s["ligand_coords"].shape[0]] = s["ligand_coords"]
        out["ligand_coords"] = lc
    if all("binding_value" in s for s in batch):
        out["binding_value"] = torch.stack([s["binding_value"].reshape(()) for s in batch]).float()
    if all("interaction_mask" in s for s in batch):
        im = torch.zeros(B, Lp, dtype=torch.float)
        for i, s in enumerate(batch):
            im[i, :len(s["interaction_mask"])] = s["interaction_mask"]
        out["interaction_mask"] = im

    def is_ids(seq):
        return not isinstance(seq, str)

    if token_ids is None:
        token_ids = (has_p and all(is_ids(s["protein_seq"]) for s in batch)) or \
                    (has_l and all(is_ids(s["ligand_smiles"]) for s in batch))
    if token_ids:
        if has_p:
            out["prot_tokens"], _ = pad_to_fixed(out["protein_seq"], Lp, protein_pad_id)
        if has_l:
            out["lig_tokens"], _ = pad_to_fixed(out["ligand_smiles"], Ll, ligand_pad_id)
    return out


def ddp_concat_batch(lig_ids, prot_ids, protein_offset=2363):
    """run_train_ddp_protlig_transf.py:276: the single-track SEDD of the DDP trainer sees ligand tokens followed by protein
    tokens shifted past the ligand vocabulary — batch = cat([lig_ids, prot_ids + 2363], 1)."""
    return torch.cat([lig_ids, prot_ids + protein_offset], dim=1)


def split_indices(n, train_ratio=0.8, val_ratio=0.1, seed=42):
    """the three index lists of `random_split(dataset, [train, val, test], generator=Generator().manual_seed(seed))`
    (data/process_full_plinder.py:293-301): one permutation, cut at int(train_ratio n) and int(val_ratio n)."""
    n_train, n_val = int(train_ratio * n), int(val_ratio * n)
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(seed)).tolist()
    return perm[:n_train], perm[n_train:n_train + n_val], perm[n_train + n_val:]


class BatchLoader:
    """What the reference builds with DataLoader(subset, batch_size, collate_fn, sampler=DistributedSampler(subset)) /
    shuffle=True (data/process_full_plinder.py:306-331, :396-430), without worker processes (token-id lists collate in
    microseconds): iterates `indices` of `dataset` in this rank's DistributedSampler order (`distributed_indices`) —
    or, world == 1 and shuffle, a seeded permutation per epoch — in batches through `collate_fn`."""

    def __init__(self, dataset, indices, batch_size, collate_fn=protein_ligand_collate_fn, rank=0, world=1, shuffle=False,
                 seed=0, drop_last=False, distributed=False):
        self.dataset, self.indices, self.batch_size, self.collate_fn = dataset, list(indices), batch_size, collate_fn
        self.rank, self.world, self.shuffle, self.seed, self.drop_last = rank, world, shuffle, seed, drop_last
        self.distributed = distributed or world > 1
        self.epoch = 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def order(self):
        n = len(self.indices)
        if self.distributed:
            pos = distributed_indices(n, self.rank, self.world, shuffle=self.shuffle, seed=self.seed, epoch=self.epoch)
        elif self.shuffle:
            pos = torch.randperm(n, generator=torch.Generator().manual_seed(self.seed + self.epoch)).tolist()
        else:
            pos = list(range(n))
        return [self.indices[p] for p in pos]

    def __len__(self):
        n = len(self.order())
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        order = self.order()
        for i in range(0, len(order), self.batch_size):
            chunk = order[i:i + self.batch_size]
            if self.drop_last and len(chunk) < self.batch_size:
                return
            yield self.collate_fn([self.dataset[j] for j in chunk])


def create_data_loaders(data, batch_size=32, train_ratio=0.8, val_ratio=0.1, max_protein_len=1024, max_ligand_len=128,
                        seed=42, rank=0, world=1, distributed=False, collate_fn=protein_ligand_collate_fn):
    """create_improved_data_loaders (data/process_full_plinder.py:246-333) and, with distributed=True / world > 1,
    ddp_data_loaders (:337-432): one dataset, the seeded random_split, train shuffled (DDP: every loader through the
    DistributedSampler, which shuffles with seed 0) -> (train_loader, val_loader, test_loader)."""
    ds = ProteinLigandDataset(data, max_protein_len=max_protein_len, max_ligand_len=max_ligand_len)
    tr, va, te = split_indices(len(ds), train_ratio, val_ratio, seed)
    dist_ = distributed or world > 1
    mk = lambda idx, shuffle: BatchLoader(ds, idx, batch_size, collate_fn, rank, world, shuffle=shuffle or dist_,
                                          seed=0 if dist_ else seed, distributed=dist_)
    return mk(tr, True), mk(va, False), mk(te, False)
