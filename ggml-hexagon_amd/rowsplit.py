"""ggml row split for one-process-per-GPU runs: which weight rows a rank owns, and the concat step.

Partitioning follows the reference tree's only row-split implementation (behavioural spec,
ggml/src/ggml-cuda/ggml-cuda.cu:727-753): cumulative fractions `tensor_split`, row_low = nrows*split[id]
rounded DOWN to `rounding`, the last device takes the remainder.  The exchange is a concat, not a sum:
every rank computes dst[:, row_low:row_high] and the slices are gathered over RCCL (xGMI); no reduction.
"""
from __future__ import annotations

ROW_ROUNDING = 256       # the prefill kernels' row tile and the plugin's SPLIT_ROW_ROUNDING (ggml-mi355x.cpp): a slice never starts inside a 256-row tile


def rounding_for(nrows: int, world: int, rounding: int = ROW_ROUNDING) -> int:
    """the row rounding of one matrix: ROW_ROUNDING where every rank still gets rows, else the largest power of two (>= 32) that
    leaves none empty.  A 1024-row wk / wv over 8 ranks would otherwise come out as 0 / 256 / 0 / 256 ... rows: half
    the devices idle on it and the group's slices stop being equal, which costs the one-all-gather-per-group exchange."""
    r = rounding
    while r > 32 and nrows // world < r:
        r //= 2
    return r


def row_range(nrows: int, rank: int, world: int, tensor_split=None, rounding: int | None = None):
    if tensor_split is None:
        tensor_split = [i / world for i in range(world)]           # equal devices: cumulative starts
    if rounding is None:
        rounding = rounding_for(nrows, world)
    lo = int(nrows * tensor_split[rank])
    lo -= lo % rounding
    if rank == world - 1:
        hi = nrows
    else:
        hi = int(nrows * tensor_split[rank + 1])
        hi -= hi % rounding
    return lo, hi


def all_ranges(nrows: int, world: int, tensor_split=None, rounding: int | None = None):
    return [row_range(nrows, r, world, tensor_split, rounding) for r in range(world)]


class RowConcat:
    """Gathers per-rank dst slices [N, rows_r] into the full dst [N, M] on every rank.

    Equal slices use one all_gather_into_tensor (RCCL ring/direct over xGMI) into a [world, N, rows] staging
    buffer followed by one permute-copy; ragged slices (last rank takes the remainder) are padded to the widest.  Works unchanged with the gloo backend on CPU tensors (tests)."""

    def __init__(self, group=None, always_collective=False):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.always = always_collective      # a world of one still goes through the backend (rehearsal of RCCL + hipGraph capture on a one-GPU box)
        self._stage = {}          # staging buffers by (rows, cols, dtype, device): allocated once, the exchange runs per MUL_MAT group

    def _buf(self, rows, cols, like, zero=False):
        import torch
        key = (rows, cols, like.dtype, like.device, zero)
        b = self._stage.get(key)
        if b is None:
            b = (torch.zeros if zero else torch.empty)((rows, cols), dtype=like.dtype, device=like.device)
            self._stage[key] = b
        return b

    def concat(self, local, ranges, out=None):
        import torch
        n = local.shape[0]
        m = ranges[-1][1]
        sizes = [hi - lo for lo, hi in ranges]
        if out is None:
            out = torch.empty((n, m), dtype=local.dtype, device=local.device)
        if self.world == 1 and not self.always:
            out.copy_(local)
            return out
        equal = len(set(sizes)) == 1 and local.is_contiguous()
        if equal and n == 1 and out.is_contiguous():
            # one token (token generation): [world, rows] in rank order IS dst; gather straight into it
            self.dist.all_gather_into_tensor(out.view(self.world, sizes[0]), local, group=self.group)
        elif equal:
            stage = self._buf(self.world * n, sizes[0], local)
            self.dist.all_gather_into_tensor(stage, local, group=self.group)
            out.view(n, self.world, sizes[0]).copy_(stage.view(self.world, n, sizes[0]).permute(1, 0, 2))
        else:
            # ragged slices (ggml gives the remainder rows to the last device): pad to the widest slice so that one
            # equal-size all-gather still does the exchange, then drop the padding while concatenating
            smax = max(sizes)
            padded = self._buf(n, smax, local, zero=True)
            padded[:, :local.shape[1]] = local
            stage = self._buf(self.world * n, smax, local)
            self.dist.all_gather_into_tensor(stage, padded, group=self.group)
            stage = stage.view(self.world, n, smax)
            for r, (lo, hi) in enumerate(ranges):
                if hi > lo:
                    out[:, lo:hi] = stage[r, :, :hi - lo]
        return out

    def concat_group(self, local, cols, ranges_list, outs):
        """ONE all-gather for a group of MUL_MATs that share src1 (attn_q/k/v, ffn_gate/up): `local` is this rank's [N, sum(cols)]
        buffer whose column slices the group's launch wrote (cols[i] = this rank's rows of matrix i, the same on every rank);
        outs[i] [N, M_i] receives matrix i's full dst.  The gathered [world, N, sum(cols)] block is taken apart by one strided copy
        per matrix (the placement the reference's per-device cudaMemcpy2D into the main device's dst does, ggml-cuda.cu:1603-1625)."""
        n, tot = local.shape
        assert tot == sum(cols) and local.is_contiguous()
        if self.world == 1 and not self.always:
            off = 0
            for c, o in zip(cols, outs):
                o.copy_(local[:, off:off + c])
                off += c
            return outs
        stage = self._buf(self.world * n, tot, local)
        self.dist.all_gather_into_tensor(stage, local, group=self.group)
        st = stage.view(self.world, n, tot)
        off = 0
        for c, ranges, o in zip(cols, ranges_list, outs):
            assert all(hi - lo == c for lo, hi in ranges), "concat_group needs equal slices (use concat for ragged ones)"
            o.view(n, self.world, c).copy_(st[:, :, off:off + c].permute(1, 0, 2))
            off += c
        return outs

