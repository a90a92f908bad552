"""The MUL_MAT / MUL_MAT_ID sequence one llama.cpp forward pass issues, per model config of BASELINE.json.

llama.cpp builds, per layer (src/llama-model.cpp:4191-4350 llm_build_llama): wq, wk, wv (same src1), wo,
ffn gate + up (same src1), ffn down; after the last layer the output projection.  MoE models replace the
FFN by three MUL_MAT_ID over n_expert stacked matrices (src/llama-graph.cpp:870-894).  Weight types follow
llama-quant.cpp's recipes (Q4_K_M: use_more_bits layers get Q6_K for attn_v and ffn_down, output Q6_K;
src/llama-quant.cpp:129-131, 166-168, 235-236, 291-297).  Only the shapes and types matter here: the
weights are synthetic (synth.py) and llama-bench feeds random token ids (examples/llama-bench/llama-bench.cpp:1443-1466).
"""
from __future__ import annotations

from dataclasses import dataclass, field

from .synth import IQ4_XS, Q2_K, Q3_K, Q4_0, Q4_K, Q5_K, Q6_K, Q8_0, row_size


@dataclass
class MatMul:
    name: str
    type: int
    K: int
    M: int
    n_expert: int = 0          # > 0: MUL_MAT_ID over n_expert matrices, n_used of them per token
    n_used: int = 0

    @property
    def weight_bytes(self) -> int:
        return row_size(self.type, self.K) * self.M * max(self.n_expert, 1)

    def algo_bytes(self, n_tokens: int) -> int:
        """weights read once (for MoE: the experts actually used, bounded by n_expert) + src1 + dst"""
        if self.n_expert:
            used = min(self.n_expert, self.n_used * n_tokens)
            return row_size(self.type, self.K) * self.M * used + n_tokens * self.K * 4 + n_tokens * self.n_used * self.M * 4
        return self.weight_bytes + n_tokens * self.K * 4 + n_tokens * self.M * 4

    def flops(self, n_tokens: int) -> int:
        return 2 * self.M * self.K * n_tokens * (self.n_used if self.n_expert else 1)


@dataclass
class Group:
    """MUL_MATs that share src1 (issued as one grouped launch).  outputs_only: in a prompt batch llama.cpp computes this
    group on the rows whose logits are wanted only: after the last layer's attention `cur` is cut down to the output rows
    (`ggml_get_rows(cur, inp_out_ids)`, src/llama-model.cpp:4270-4275), so the last layer's FFN and the output projection
    see n_outputs tokens; llama-bench's prompt test asks for the last token's logits only (llama_batch_get_one leaves
    batch.logits NULL -> n_outputs_all = 1, src/llama-context.cpp:1232-1244)."""
    mats: list
    outputs_only: bool = False


@dataclass
class Workload:
    name: str
    n_layer: int
    n_embd: int
    groups: list = field(default_factory=list)     # per-token sequence of Groups (all layers + output)

    def all_mats(self):
        return [m for g in self.groups for m in g.mats]

    def weight_bytes(self) -> int:
        return sum(m.weight_bytes for m in self.all_mats())

    def group_tokens(self, g: Group, n_tokens: int, n_outputs=None) -> int:
        return n_tokens if n_outputs is None or not g.outputs_only else min(n_tokens, n_outputs)

    def algo_bytes(self, n_tokens: int, n_outputs=None) -> int:
        return sum(m.algo_bytes(self.group_tokens(g, n_tokens, n_outputs)) for g in self.groups for m in g.mats)

    def flops(self, n_tokens: int, n_outputs=None) -> int:
        return sum(m.flops(self.group_tokens(g, n_tokens, n_outputs)) for g in self.groups for m in g.mats)


def use_more_bits(i: int, n: int) -> bool:      # src/llama-quant.cpp:129-131
    return i < n // 8 or i >= 7 * n // 8 or (i - n // 8) % 3 == 2


def _llama(name, n_layer, n_embd, n_ff, n_head, n_head_kv, n_vocab, recipe, n_expert=0, n_used=0) -> Workload:
    kv = n_embd // n_head * n_head_kv
    w = Workload(name, n_layer, n_embd)
    for i in range(n_layer):
        more = use_more_bits(i, n_layer)
        if recipe == "q4_0":
            tq = tk = tv = to = tg = td = Q4_0
        elif recipe == "q4_k":                       # north-star synthetic: every matmul weight Q4_K
            tq = tk = tv = to = tg = td = Q4_K
        elif recipe == "iq4_xs":                     # round 3: llama-quant.cpp:232-234 (attn_v at n_gqa >= 4), :299-301 (ffn_down of the first eighth)
            tq = tk = tv = to = tg = td = IQ4_XS
            if n_head // n_head_kv >= 4:
                tv = Q5_K
            if i < n_layer // 8:
                td = Q5_K
        elif recipe == "q3_k_m":                     # llama-quant.cpp:228-230 (attn_v), :279-283 (ffn_down), :326 (attn_output)
            tq = tk = tg = Q3_K
            tv = Q5_K if i < 2 else Q4_K
            td = Q5_K if i < n_layer // 16 else Q4_K
            to = Q4_K
        elif recipe == "q2_k":                       # llama-quant.cpp:213-215 (attn_v), :272 (ffn_down), :324 (attn_output)
            tq = tk = tg = Q2_K
            tv = Q4_K if n_head // n_head_kv >= 4 else Q3_K
            td = to = Q3_K
        else:                                        # q4_k_m
            tq = tk = to = tg = Q4_K
            tv = td = Q6_K if more else Q4_K
            if n_layer >= 80 and tv == Q4_K:         # 70B rule: attn_v Q4_K -> Q5_K (llama-quant.cpp:237-243)
                tv = Q5_K
            if n_expert == 8:                        # llama-quant.cpp:244-255, 314-322
                tk = tv = Q8_0
                to = Q5_K
        w.groups.append(Group([MatMul(f"blk.{i}.attn_q", tq, n_embd, n_embd), MatMul(f"blk.{i}.attn_k", tk, n_embd, kv),
                               MatMul(f"blk.{i}.attn_v", tv, n_embd, kv)]))
        w.groups.append(Group([MatMul(f"blk.{i}.attn_output", to, n_embd, n_embd)]))
        last = i == n_layer - 1                      # its FFN runs on the output rows only in a prompt batch (Group docstring)
        if n_expert:
            w.groups.append(Group([MatMul(f"blk.{i}.ffn_gate_exps", tg, n_embd, n_ff, n_expert, n_used),
                                   MatMul(f"blk.{i}.ffn_up_exps", tg, n_embd, n_ff, n_expert, n_used)], last))
            w.groups.append(Group([MatMul(f"blk.{i}.ffn_down_exps", td, n_ff, n_embd, n_expert, n_used)], last))
        else:
            w.groups.append(Group([MatMul(f"blk.{i}.ffn_gate", tg, n_embd, n_ff), MatMul(f"blk.{i}.ffn_up", tg, n_embd, n_ff)], last))
            w.groups.append(Group([MatMul(f"blk.{i}.ffn_down", td, n_ff, n_embd)], last))
    out_t = {"q4_0": Q6_K, "q4_k": Q4_K}.get(recipe, Q6_K)   # stock Q4_0 files carry a Q6_K output tensor
    w.groups.append(Group([MatMul("output", out_t, n_embd, n_vocab)], True))
    return w


WORKLOADS = {
    # BASELINE.json configs[1..4] + the north-star synthetic model
    "llama2-7b-q4_0":     lambda: _llama("llama2-7b-q4_0", 32, 4096, 11008, 32, 32, 32000, "q4_0"),
    "llama3-8b-q4_k_m":   lambda: _llama("llama3-8b-q4_k_m", 32, 4096, 14336, 32, 8, 128256, "q4_k_m"),
    "llama3-70b-q4_k_m":  lambda: _llama("llama3-70b-q4_k_m", 80, 8192, 28672, 64, 8, 128256, "q4_k_m"),
    "mixtral-8x7b-q4_k_m": lambda: _llama("mixtral-8x7b-q4_k_m", 32, 4096, 14336, 32, 8, 32000, "q4_k_m", 8, 2),
    "synth-7b-q4_k":      lambda: _llama("synth-7b-q4_k", 32, 4096, 11008, 32, 32, 32000, "q4_k"),
    # not BASELINE configs: other weight formats on the llama3-8b shapes (`bench.py --workload llama3-8b-iq4_xs`)
    "llama3-8b-iq4_xs":   lambda: _llama("llama3-8b-iq4_xs", 32, 4096, 14336, 32, 8, 128256, "iq4_xs"),
    "llama3-8b-q3_k_m":   lambda: _llama("llama3-8b-q3_k_m", 32, 4096, 14336, 32, 8, 128256, "q3_k_m"),
    "llama3-8b-q2_k":     lambda: _llama("llama3-8b-q2_k", 32, 4096, 14336, 32, 8, 128256, "q2_k"),
}


def get(name: str) -> Workload:
    return WORKLOADS[name]()
