"""Drives the offloaded surface — every MUL_MAT / MUL_MAT_ID of one forward pass — through the C-ABI,
with weights, activations and outputs resident in HBM.  This is what bench.py times: one `run(n_tokens)`
is what llama.cpp's scheduler hands to the backend for one ubatch (n_tokens = 512 for pp512, 1 for tg128),
minus the host-side glue ops that are outside the offloaded surface.

Multi-GPU (one process per GPU): ggml row split — rank r holds rows [lo_r, hi_r) of every weight
(rowsplit.row_range), computes its slice of dst, and the slices are concatenated over RCCL after each
group of MUL_MATs that share src1.  MUL_MAT_ID is not row-split (the reference tree's row split
excludes it too, ggml-cuda.cu:1973), so MoE workloads run on one GPU.
"""
from __future__ import annotations

import torch

from . import rowsplit, synth
from .workload import Workload


class HotPath:
    def __init__(self, qmm, wl: Workload, device, rank: int = 0, world: int = 1, concat=None, seed: int = 1234, planar: bool = True):
        self.q, self.wl, self.dev, self.rank, self.world, self.concat = qmm, wl, device, rank, world, concat
        self.split = world > 1 or (concat is not None and getattr(concat, "always", False))   # exchange after every group (a forced world of one included)
        self.weights = {}       # name -> (uint8 tensor [rows, row_bytes] | [n_expert, M, row_bytes], ranges)
        self.types = {}         # name -> type code for the C-ABI: the planar code where the rows were re-laid, as the
        #                         plugin does with Q4_0 / Q8_0 / Q6_K weights at their first use; same bytes, same results
        i = 0
        for g in wl.groups:
            for m in g.mats:
                i += 1
                if m.n_expert:
                    if world != 1:
                        raise RuntimeError("MUL_MAT_ID workloads are not row-split; run with --gpus 1")
                    w = synth.synth_weights_torch(m.type, m.n_expert * m.M, m.K, device, seed + i).reshape(m.n_expert, m.M, -1)
                    self.weights[m.name] = (w, None)
                    self.types[m.name] = self._layout(m.type, w, m.K, planar)
                else:
                    ranges = rowsplit.all_ranges(m.M, world)
                    lo, hi = ranges[rank]
                    w = synth.synth_weights_torch(m.type, hi - lo, m.K, device, seed + 1000 * rank + i)
                    self.weights[m.name] = (w, ranges)
                    self.types[m.name] = self._layout(m.type, w, m.K, planar)
        self.io = {}
        self._calls = {}
        self.chain = False       # persistent chains for token generation (bench.py --chain): measured slower than launches, DESIGN.md

    def _layout(self, t, w, k, planar):
        if planar and w.numel() and self.q.planar_type(t, k, w.stride(-2)):
            return self.q.repack_rows(t, w, k, True)
        return t

    def weight_bytes_local(self) -> int:
        return sum(w.numel() for w, _ in self.weights.values())

    def prepare(self, n_tokens: int, seed: int = 7):
        """activations (one per distinct K) and dst buffers for a batch size; synthetic, resident in HBM"""
        if n_tokens in self.io:
            return self.io[n_tokens]
        g = torch.Generator(device=self.dev)
        g.manual_seed(seed + n_tokens)
        x, dst_local, dst_full, ids = {}, {}, {}, None
        for grp in self.wl.groups:
            for m in grp.mats:
                if m.n_expert:
                    ne11 = m.n_used if m.name.endswith("down_exps") else 1
                    key = (m.K, ne11)
                    if key not in x:
                        x[key] = torch.rand((n_tokens, ne11, m.K), device=self.dev, generator=g) * 2 - 1
                    if ids is None:     # per token: a shuffle of the experts, first n_used taken (test-backend-ops.cpp:2113-2132)
                        ids = torch.stack([torch.randperm(m.n_expert, device=self.dev, generator=g) for _ in range(n_tokens)]).to(torch.int32)
                    dkey = ("id", m.name.split(".")[-1], m.M)     # paired launches (gate_exps + up_exps) write distinct buffers
                    if dkey not in dst_local:
                        dst_local[dkey] = torch.empty((n_tokens, m.n_used, m.M), device=self.dev)
                else:
                    if m.K not in x:
                        x[m.K] = torch.rand((n_tokens, m.K), device=self.dev, generator=g) * 2 - 1
                    w, ranges = self.weights[m.name]
                    rows = w.shape[0]
                    # grouped launches write distinct buffers; different groups may share them
                    dkey = (m.name.split(".")[-1], rows)
                    if dkey not in dst_local:
                        dst_local[dkey] = torch.empty((n_tokens, rows), device=self.dev)
                    if self.split and dkey not in dst_full:
                        dst_full[dkey] = torch.empty((n_tokens, m.M), device=self.dev)
        if self.split:
            # row split: a group's local outputs are column slices of ONE buffer, so that its dst slices travel in one all-gather
            # (one exchange per group, not per matrix: 129 instead of 225 per token for llama3-8b).  Groups with a ragged split
            # (rows not a multiple of 64 x world: the output matrix) keep their own buffers and the padded per-matrix exchange.
            for grp in self.wl.groups:
                if grp.mats[0].n_expert or len(grp.mats) < 2:
                    continue
                rows = [self.weights[m.name][0].shape[0] for m in grp.mats]
                if any(len({hi - lo for lo, hi in self.weights[m.name][1]}) != 1 for m in grp.mats):
                    continue
                gkey = ("group",) + tuple((m.name.split(".")[-1], r) for m, r in zip(grp.mats, rows))
                if gkey not in dst_local:
                    buf = torch.empty((n_tokens, sum(rows)), device=self.dev)
                    offs = [sum(rows[:i]) for i in range(len(rows))]
                    dst_local[gkey] = (buf, [buf[:, o:o + r] for o, r in zip(offs, rows)])   # the group's launch writes these views
        self.io[n_tokens] = (x, dst_local, dst_full, ids)
        return self.io[n_tokens]

    def run(self, n_tokens: int, n_outputs=None):
        """issue the whole pass on the current stream (asynchronous).  n_outputs: how many of the batch's tokens want
        logits; the groups llama.cpp computes on the output rows only (workload.Group.outputs_only) run at that size
        (llama-bench's prompt test: 1).  None = every token (token generation: n_tokens = 1 anyway)."""
        io_all = self.prepare(n_tokens)
        io_out = io_all if n_outputs is None or n_outputs >= n_tokens else self.prepare(n_outputs)
        q = self.q
        # token generation on one GPU: the pass is one chain of dependent groups (every MUL_MAT of a decode step consumes what the
        # previous one produced, through the glue ops), issued as persistent launches (csrc/qmm_chain.hiph).  The row split keeps
        # one launch per group: an RCCL concat sits between two groups.
        chain = self.chain and n_tokens == 1 and self.world == 1
        if chain:
            q.chain_begin()
        try:
            self._issue(q, io_all, io_out)
        finally:
            if chain:
                q.chain_end()

    def group_call(self, grp, io):
        """the launches of ONE group of the pass as a closure: what `_issue` runs and what bench.py's roofline leg times, so the two
        cannot drift apart (the leg used to cut q/k/v into same-type runs that the timed pass never issues).  Built once
        per (group, batch) and kept: the argument marshalling is not part of a pass."""
        key = (id(grp), id(io))
        fn = self._calls.get(key)
        if fn is None:
            fn = self._calls[key] = self._group_call(grp, io)
        return fn

    def _group_call(self, grp, io):
        q = self.q
        x, dst_local, dst_full, ids = io
        m0 = grp.mats[0]
        if m0.n_expert:
            ne11 = m0.n_used if m0.name.endswith("down_exps") else 1
            outs = [dst_local[("id", m.name.split(".")[-1], m.M)] for m in grp.mats]
            if len(grp.mats) == 2 and grp.mats[1].type == m0.type:      # ffn_gate_exps + ffn_up_exps: same b, same ids
                t, w0, w1, xx, ii = self.types[m0.name], self.weights[grp.mats[0].name][0], self.weights[grp.mats[1].name][0], x[(m0.K, ne11)], ids[:, :m0.n_used]
                return lambda: q.mul_mat_id_pair(t, w0, w1, m0.K, xx, ii, outs[0], outs[1])
            calls = [(self.types[m.name], self.weights[m.name][0], m.K, x[(m.K, ne11)], ids[:, :m.n_used], o) for m, o in zip(grp.mats, outs)]

            def run_ids():
                for t, w, k, xx, ii, o in calls:
                    q.mul_mat_id(t, w, k, xx, ii, out=o)
            return run_ids
        ws, outs, keys = [], [], []
        for m in grp.mats:
            w, ranges = self.weights[m.name]
            dkey = (m.name.split(".")[-1], w.shape[0])
            ws.append((self.types[m.name], w))
            outs.append(dst_local[dkey])
            keys.append((dkey, ranges))
        gkey = ("group",) + tuple(dkey for dkey, _ in keys)
        xk = x[m0.K]
        # marshalling once where the binding offers it (capi.Qmm); any object with mul_mat_group will do (the CPU stand-in of the tests)
        prepared = getattr(q, "mul_mat_group_call", None) or (lambda w_, k_, x_, o_: (lambda: q.mul_mat_group(w_, k_, x_, o_)))
        if self.split and gkey in dst_local:
            buf, gouts = dst_local[gkey]
            cols, rngs, fulls = [dkey[1] for dkey, _ in keys], [r for _, r in keys], [dst_full[dkey] for dkey, _ in keys]
            mm = prepared(ws, m0.K, xk, gouts)

            def run_split_group():
                mm()
                self.concat.concat_group(buf, cols, rngs, fulls)
            return run_split_group
        mm = prepared(ws, m0.K, xk, outs)
        if self.split:
            def run_split():
                mm()
                for (dkey, ranges), o in zip(keys, outs):
                    self.concat.concat(o, ranges, out=dst_full[dkey])
            return run_split
        return mm

    def _issue(self, q, io_all, io_out):
        for grp in self.wl.groups:
            self.group_call(grp, io_out if grp.outputs_only else io_all)()

    def capture(self, n_tokens: int, n_outputs=None):
        """hipGraph of one pass: removes the per-launch host cost from the timed loops (token generation, and the prompt pass: ~430
        launches from Python in 11 ms leave the host no slack).  With a row split the RCCL all-gathers are captured with the launches
        (collectives are capturable; tests/test_gpu_rccl.py rehearses it on one GPU)"""
        self.prepare(n_tokens)
        self.run(n_tokens, n_outputs)       # warm: lazy module loads, workspace growth happen outside the capture
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.run(n_tokens, n_outputs)
        return g
