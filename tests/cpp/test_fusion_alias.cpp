// test_fusion_alias.cpp — the aliasing cases behind the plugin's multi-node launches (csrc/ggml-mi355x.cpp early_write_ok).
// A fused launch writes a LATER node's buffer at an EARLIER point.  ggml-alloc may legally give that later node the block of a tensor
// that is dead by then in the graph's order but is still an OPERAND of the fused launch: the residual ADD behind wo placed on wo's
// src1, the SwiGLU product placed on ffn_norm's output, q placed on the un-normed row whose norm the mat-vec forms itself, the
// merged-heads CONT placed on the soft-max output.  Here those placements are FORCED (ggml_backend_tensor_alloc at chosen addresses),
// for 1, 4 and 40 tokens, and the results are compared with the CPU backend computing the same placements node by node
// (NMSE <= 5e-4 per output; the fusion must either be safe or be declined).  Public ggml API only; GGML_BACKEND_PATH=<module>.
#include "ggml.h"
#include "ggml-alloc.h"
#include "ggml-backend.h"
#include "ggml-cpu.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <random>
#include <string>
#include <vector>

static int n_ok = 0, n_fail = 0;
struct Placed { ggml_tensor * t; size_t off; };

struct Graph {
    ggml_context * ctx = nullptr;
    std::vector<Placed> place;                       // every non-view tensor and its byte offset in the one buffer
    std::vector<ggml_tensor *> views;
    std::vector<std::pair<ggml_tensor *, std::vector<uint8_t>>> init;
    std::vector<ggml_tensor *> outs;
    size_t top = 0;
    size_t fresh(ggml_tensor * t) { const size_t o = top; place.push_back({ t, o }); top += (ggml_nbytes(t) + 255) & ~(size_t) 255; return o; }
    void at(ggml_tensor * t, size_t off) { place.push_back({ t, off }); }
};

static std::vector<std::vector<float>> run(ggml_backend_t be, const std::function<void(Graph &, std::mt19937 &)> & build) {
    Graph G;
    ggml_init_params ip = { ggml_tensor_overhead() * 256 + ggml_graph_overhead(), nullptr, true };
    G.ctx = ggml_init(ip);
    std::mt19937 rng(99);
    build(G, rng);
    ggml_cgraph * g = ggml_new_graph(G.ctx);
    for (ggml_tensor * o : G.outs) ggml_build_forward_expand(g, o);
    ggml_backend_buffer_t buf = ggml_backend_alloc_buffer(be, G.top + 4096);
    char * base = (char *) ggml_backend_buffer_get_base(buf);
    for (auto & p : G.place) ggml_backend_tensor_alloc(buf, p.t, base + p.off);
    for (int i = 0; i < ggml_graph_n_nodes(g); ++i) {                               // views: after their sources have addresses
        ggml_tensor * t = ggml_graph_node(g, i);
        if (t->view_src && !t->data) ggml_backend_view_init(t);
    }
    for (auto & in : G.init) ggml_backend_tensor_set(in.first, in.second.data(), 0, in.second.size());
    if (ggml_backend_graph_compute(be, g) != GGML_STATUS_SUCCESS) { fprintf(stderr, "graph_compute failed\n"); exit(2); }
    std::vector<std::vector<float>> res;
    for (ggml_tensor * o : G.outs) { std::vector<float> v(ggml_nelements(o)); ggml_backend_tensor_get(o, v.data(), 0, v.size() * 4); res.push_back(v); }
    ggml_backend_buffer_free(buf);
    ggml_free(G.ctx);
    return res;
}

static ggml_tensor * weight(Graph & G, std::mt19937 & rng, ggml_type type, int64_t k, int64_t m, float sigma) {
    ggml_tensor * t = ggml_new_tensor_2d(G.ctx, type, k, m);
    std::uniform_real_distribution<float> u(-sigma, sigma);
    std::vector<float> f((size_t) k * m);
    for (auto & v : f) v = u(rng);
    std::vector<uint8_t> q(ggml_row_size(type, k) * m);
    if (type == GGML_TYPE_F32) memcpy(q.data(), f.data(), q.size());
    else ggml_quantize_chunk(type, f.data(), q.data(), 0, m, k, nullptr);
    G.fresh(t);
    G.init.push_back({ t, q });
    return t;
}

static ggml_tensor * weight3(Graph & G, std::mt19937 & rng, ggml_type type, int64_t k, int64_t m, int64_t e, float sigma) {
    ggml_tensor * t = ggml_new_tensor_3d(G.ctx, type, k, m, e);
    std::uniform_real_distribution<float> u(-sigma, sigma);
    std::vector<float> f((size_t) k * m * e);
    for (auto & v : f) v = u(rng);
    std::vector<uint8_t> q(ggml_row_size(type, k) * m * e);
    ggml_quantize_chunk(type, f.data(), q.data(), 0, m * e, k, nullptr);
    G.fresh(t);
    G.init.push_back({ t, q });
    return t;
}

// build_moe_ffn (src/llama-graph.cpp:800-917) with softmax gating and normalised weights, as Mixtral runs it.  `where` picks the forced
// placement: 1 = the DIV (normalised weights, rows of 4 * n_used bytes) ON the router logits (dead at the div's place in the graph,
// still read by the fused router launch, rows of 4 * n_expert bytes); 2 = the argsort's ids exactly on the logits (same rows: a wave
// reads its row before it writes it, legal in place); 3 = the ids one row further; 4 = the last expert ADD on the expert outputs (the
// operand of the fused combine launch); 5 = the last ADD on the weights
static void moe_block(Graph & G, std::mt19937 & rng, int64_t N, int where) {
    const int64_t E = 1024, F = 512, NE = 8, NU = 2;
    ggml_tensor * x = weight(G, rng, GGML_TYPE_F32, E, N, 1.0f);
    ggml_tensor * gi = weight(G, rng, GGML_TYPE_F32, E, NE, 0.05f);
    ggml_tensor * up_e = weight3(G, rng, GGML_TYPE_Q4_K, E, F, NE, 0.05f), * gate_e = weight3(G, rng, GGML_TYPE_Q4_K, E, F, NE, 0.05f);
    ggml_tensor * down_e = weight3(G, rng, GGML_TYPE_Q4_K, F, E, NE, 0.05f);
    // two spare regions nobody reads: the forced placements go there, so that a shifted or oversized tensor never runs into a live one
    ggml_tensor * spare = weight(G, rng, GGML_TYPE_F32, E, N + 1, 1.0f), * spare2 = weight(G, rng, GGML_TYPE_F32, E, N + 1, 1.0f);
    size_t o_logits = 0, o_dv = 0;
    for (auto & p : G.place) { if (p.t == spare) o_logits = p.off; if (p.t == spare2) o_dv = p.off; }
    ggml_tensor * logits = ggml_mul_mat(G.ctx, gi, x);                              G.at(logits, o_logits);
    ggml_tensor * probs = ggml_soft_max(G.ctx, logits);                             G.fresh(probs);
    ggml_tensor * sel = ggml_top_k(G.ctx, probs, NU);                               // argsort + view
    ggml_tensor * as = sel->src[0];
    if (where == 2) G.at(as, o_logits); else if (where == 3) G.at(as, o_logits + NE * 4); else if (where == 6) G.at(as, G.place[0].off); else G.fresh(as);
    ggml_tensor * wts = ggml_get_rows(G.ctx, ggml_reshape_3d(G.ctx, probs, 1, NE, N), sel);   G.fresh(wts);
    ggml_tensor * w2 = ggml_reshape_2d(G.ctx, wts, NU, N);
    ggml_tensor * ws = ggml_sum_rows(G.ctx, w2);                                    G.fresh(ws);
    ggml_tensor * dv = ggml_div(G.ctx, w2, ws);                                     G.at(dv, where == 1 ? o_logits : where == 7 ? G.place[0].off : o_dv);
    ggml_tensor * w3 = ggml_reshape_3d(G.ctx, dv, 1, NU, N);
    ggml_tensor * cur = ggml_reshape_3d(G.ctx, x, E, 1, N);
    ggml_tensor * up = ggml_mul_mat_id(G.ctx, up_e, cur, sel);                      G.fresh(up);
    ggml_tensor * gate = ggml_mul_mat_id(G.ctx, gate_e, cur, sel);                  G.fresh(gate);
    ggml_tensor * sl = ggml_silu(G.ctx, gate);                                      G.fresh(sl);
    ggml_tensor * par = ggml_mul(G.ctx, up, sl);                                    if (where == 8) G.at(par, G.place[0].off); else G.fresh(par);   // 8: on x ([F, NU, N] f32 = x's bytes)
    ggml_tensor * ex = ggml_mul_mat_id(G.ctx, down_e, par, sel);                    const size_t o_ex = G.fresh(ex);
    ggml_tensor * exw = ggml_mul(G.ctx, ex, w3);                                    G.fresh(exw);
    ggml_tensor * out = nullptr;
    for (int i = 0; i < NU; ++i) {
        ggml_tensor * ce = ggml_view_2d(G.ctx, exw, E, N, exw->nb[2], i * exw->nb[1]);
        if (i == 0) { out = ce; continue; }
        out = ggml_add(G.ctx, out, ce);
        if (i == NU - 1 && where == 4) G.at(out, o_ex);                             // on the expert outputs ([E, NU, N]: the sum fits)
        else if (i == NU - 1 && where == 5) G.at(out, o_dv);                        // on the weights (8 N bytes at the start of spare2)
        else G.fresh(out);
    }
    if (where >= 9) {
        // the layer's tail behind the block (round 3: one launch with the combine): l_out = out + x, rms_norm(l_out) * nw.
        // 10: l_out on the expert outputs, 11: the normed row exactly on the residual (in place: allowed), 12: the normed row on the
        // residual one row further, 13: l_out on the expert weights
        ggml_tensor * nw = weight(G, rng, GGML_TYPE_F32, E, 1, 1.0f);
        ggml_tensor * lo = ggml_add(G.ctx, out, x);
        if (where == 10) G.at(lo, o_ex); else if (where == 13) G.at(lo, o_dv); else G.fresh(lo);
        ggml_tensor * rn = ggml_rms_norm(G.ctx, lo, 1e-5f);                         G.fresh(rn);
        ggml_tensor * y = ggml_mul(G.ctx, rn, nw);
        if (where == 11) G.at(y, G.place[0].off); else if (where == 12) G.at(y, G.place[0].off + E * 4); else G.fresh(y);
        ggml_tensor * f1 = ggml_scale(G.ctx, y, 1.0f);                              G.fresh(f1);
        // (l_out is read last, so that a normed row placed on the residual does not change what the CPU computes for it)
        ggml_tensor * f0 = ggml_scale(G.ctx, lo, 1.0f);                             G.fresh(f0);
        G.outs = { f1, f0 };
        return;
    }
    ggml_tensor * fin = ggml_scale(G.ctx, out, 1.0f);                               G.fresh(fin);
    G.outs = { fin };
}

static void compare(const char * what, int64_t n, ggml_backend_t gpu, ggml_backend_t cpu, const std::function<void(Graph &, std::mt19937 &)> & build) {
    const auto a = run(gpu, build), b = run(cpu, build);
    bool ok = a.size() == b.size();
    double worst = 0;
    for (size_t j = 0; ok && j < a.size(); ++j) {
        double num = 0, den = 0;
        for (size_t i = 0; i < a[j].size(); ++i) { const double d = (double) a[j][i] - b[j][i]; num += d * d; den += (double) b[j][i] * b[j][i]; ok = ok && std::isfinite(a[j][i]); }
        worst = std::max(worst, den > 0 ? num / den : num);
    }
    ok = ok && worst <= 5e-4;
    printf("  %-58s N=%-3lld nmse %.1e : %s\n", what, (long long) n, worst, ok ? "OK" : "FAIL");
    ok ? ++n_ok : ++n_fail;
}

int main() {
    ggml_backend_load_all();
    ggml_backend_reg_t reg = ggml_backend_reg_by_name("MI355X");
    if (!reg) { fprintf(stderr, "MI355X backend not loaded (GGML_BACKEND_PATH?)\n"); return 2; }
    ggml_backend_t gpu = ggml_backend_dev_init(ggml_backend_reg_dev_get(reg, 0), nullptr);
    ggml_backend_t cpu = ggml_backend_init_by_type(GGML_BACKEND_DEVICE_TYPE_CPU, nullptr);
    const int64_t E = 1024, F = 2048;
    for (int64_t N : { (int64_t) 1, (int64_t) 4, (int64_t) 40 }) {
        // 1. dst = W x + r with the ADD's buffer ON x (x is dead once wo has read it, in the graph's order)
        compare("residual ADD placed on the MUL_MAT's src1", N, gpu, cpu, [&](Graph & G, std::mt19937 & rng) {
            ggml_tensor * x = weight(G, rng, GGML_TYPE_F32, E, N, 1.0f), * r = weight(G, rng, GGML_TYPE_F32, E, N, 1.0f);
            ggml_tensor * W = weight(G, rng, GGML_TYPE_Q4_K, E, E, 0.05f);
            ggml_tensor * mm = ggml_mul_mat(G.ctx, W, x);  G.fresh(mm);
            ggml_tensor * out = ggml_add(G.ctx, mm, r);    G.at(out, G.place[0].off);                 // == x
            ggml_tensor * fin = ggml_scale(G.ctx, out, 1.0f);  G.fresh(fin);
            G.outs = { fin };
        });
        // 2. SwiGLU product placed on the activations gate / up read
        compare("silu(gate) * up placed on ffn_gate / ffn_up's src1", N, gpu, cpu, [&](Graph & G, std::mt19937 & rng) {
            ggml_tensor * x = weight(G, rng, GGML_TYPE_F32, E, N, 1.0f);
            ggml_tensor * big = weight(G, rng, GGML_TYPE_F32, F, N, 1.0f);                             // room behind x for an [F, N] result
            ggml_tensor * Wg = weight(G, rng, GGML_TYPE_Q4_K, E, F, 0.05f), * Wu = weight(G, rng, GGML_TYPE_Q4_K, E, F, 0.05f);
            ggml_tensor * gt = ggml_mul_mat(G.ctx, Wg, x);  G.fresh(gt);
            ggml_tensor * sl = ggml_silu(G.ctx, gt);        G.fresh(sl);
            ggml_tensor * up = ggml_mul_mat(G.ctx, Wu, x);  G.fresh(up);
            ggml_tensor * pr = ggml_mul(G.ctx, sl, up);     G.at(pr, G.place[0].off);                  // starts at x, spills into `big`
            ggml_tensor * Wd = weight(G, rng, GGML_TYPE_Q4_K, F, E, 0.05f);
            ggml_tensor * dn = ggml_mul_mat(G.ctx, Wd, pr); G.fresh(dn);
            (void) big;
            G.outs = { dn };
        });
        // 3. q placed on the un-normed row (dead behind RMS_NORM * w), whose norm the mat-vec kernels form while staging
        compare("wq's result placed on the input of the folded RMS_NORM", N, gpu, cpu, [&](Graph & G, std::mt19937 & rng) {
            ggml_tensor * h = weight(G, rng, GGML_TYPE_F32, E, N, 1.0f);
            ggml_tensor * nw = weight(G, rng, GGML_TYPE_F32, E, 1, 1.0f);
            ggml_tensor * Wq = weight(G, rng, GGML_TYPE_Q4_K, E, E, 0.05f), * Wk = weight(G, rng, GGML_TYPE_Q4_K, E, 256, 0.05f);
            ggml_tensor * rn = ggml_rms_norm(G.ctx, h, 1e-5f);  G.fresh(rn);
            ggml_tensor * xn = ggml_mul(G.ctx, rn, nw);         G.fresh(xn);
            ggml_tensor * q = ggml_mul_mat(G.ctx, Wq, xn);      G.at(q, G.place[0].off);              // == h
            ggml_tensor * k = ggml_mul_mat(G.ctx, Wk, xn);      G.fresh(k);
            ggml_tensor * fq = ggml_scale(G.ctx, q, 1.0f), * fk = ggml_scale(G.ctx, k, 1.0f);
            G.fresh(fq); G.fresh(fk);
            G.outs = { fq, fk };
        });
        // 4. ADD -> RMS_NORM -> MUL(w) as one launch with the MUL's buffer on an ADD operand, shifted by one row's worth of bytes
        compare("normed row placed on an operand of the residual ADD", N, gpu, cpu, [&](Graph & G, std::mt19937 & rng) {
            ggml_tensor * a = weight(G, rng, GGML_TYPE_F32, E, N + 1, 1.0f);                           // one spare row: the shifted result fits
            ggml_tensor * b = weight(G, rng, GGML_TYPE_F32, E, N, 1.0f), * nw = weight(G, rng, GGML_TYPE_F32, E, 1, 1.0f);
            ggml_tensor * av = ggml_view_2d(G.ctx, a, E, N, E * 4, 0);
            ggml_tensor * sum = ggml_add(G.ctx, av, b);         G.fresh(sum);
            ggml_tensor * rn = ggml_rms_norm(G.ctx, sum, 1e-5f);  G.fresh(rn);
            ggml_tensor * xn = ggml_mul(G.ctx, rn, nw);         G.at(xn, G.place[0].off + (N > 1 ? E * 4 : 0));   // on a, one row further
            ggml_tensor * f1 = ggml_scale(G.ctx, xn, 1.0f), * f2 = ggml_scale(G.ctx, sum, 1.0f);
            G.fresh(f1); G.fresh(f2);
            G.outs = { f1, f2 };
        });
        // 5./6. build_attn_mha's chain (kq -> soft_max -> kqv -> permute -> cont) as ONE launch with the merged-heads CONT on the dead Q:
        // exactly on it (llama.cpp's graphs get this placement from ggml-alloc in every layer; head for head in place, the fused
        // launch may keep it) and one head further (must be declined or survived)
        for (int shift = 0; shift < 2; ++shift)
            compare(shift ? "merged heads placed on Q, one head further" : "merged heads placed exactly on Q", N, gpu, cpu, [&](Graph & G, std::mt19937 & rng) {
                const int64_t D = 128, H = 8, Hk = 2, n_kv = 256;
                ggml_tensor * qf = weight(G, rng, GGML_TYPE_F32, D * H, N + 1, 1.0f);                     // one spare row behind Q
                ggml_tensor * kc = weight(G, rng, GGML_TYPE_F16, D, n_kv * Hk, 1.0f), * vc = weight(G, rng, GGML_TYPE_F16, n_kv, D * Hk, 1.0f);
                ggml_tensor * mk = weight(G, rng, GGML_TYPE_F32, n_kv, N, 1.0f);
                ggml_tensor * q3 = ggml_view_3d(G.ctx, qf, D, H, N, D * 4, D * H * 4, 0);
                ggml_tensor * qp = ggml_permute(G.ctx, q3, 0, 2, 1, 3);
                ggml_tensor * k3 = ggml_view_3d(G.ctx, kc, D, n_kv, Hk, D * 2, D * n_kv * 2, 0);
                ggml_tensor * v3 = ggml_view_3d(G.ctx, vc, n_kv, D, Hk, n_kv * 2, n_kv * D * 2, 0);
                ggml_tensor * kq = ggml_mul_mat(G.ctx, k3, qp);                                           G.fresh(kq);
                ggml_tensor * sm = ggml_soft_max_ext(G.ctx, kq, mk, 0.0883883f, 0.0f);                    G.fresh(sm);
                ggml_tensor * kqv = ggml_mul_mat(G.ctx, v3, sm);                                          G.fresh(kqv);
                ggml_tensor * pm = ggml_permute(G.ctx, kqv, 0, 2, 1, 3);
                ggml_tensor * ct = ggml_cont_2d(G.ctx, pm, D * H, N);          G.at(ct, G.place[0].off + (shift ? D * 4 : 0));   // on Q
                ggml_tensor * fin = ggml_scale(G.ctx, ct, 1.0f);               G.fresh(fin);
                G.outs = { fin };
            });
        // 7.-11. build_moe_ffn: the router launch (soft_max .. div in one) and the combine launch (experts * weights + the adds in one)
        // with their early-written results forced onto operands of the same launch
        const char * moe_what[] = { "MoE block, free placement", "normalised weights (DIV) placed on the router logits", "argsort ids placed exactly on the logits",
                                    "argsort ids placed on the logits, one row further", "last expert ADD placed on the expert outputs",
                                    "last expert ADD placed on the expert weights",
                                    // round 3: for a few tokens the launch also computes the logits, i.e. reads the block's input while it writes
                                    "argsort ids placed on the router MUL_MAT's src1", "normalised weights (DIV) placed on the router MUL_MAT's src1",
                                    // ... and the expert pair writes silu(gate) * up itself, while src1 is still being staged
                                    "silu(gate) * up of the experts placed on their src1",
                                    // ... and the combine takes the residual add and the RMS norm behind the block along
                                    "MoE block + residual add + RMS norm, free placement", "l_out placed on the expert outputs",
                                    "normed row placed exactly on the residual", "normed row placed on the residual, one row further",
                                    "l_out placed on the expert weights" };
        for (int where = 0; where <= 13; ++where)
            compare(moe_what[where], N, gpu, cpu, [&](Graph & G, std::mt19937 & rng) { moe_block(G, rng, N, where); });
    }
    // LAST (it leaves the device in bf16 prefill mode): a prompt batch over a weight whose blocks exceed the f16 range (d * sc * q ~ 1e5:
    // valid Q4_K bits) must come out right, not fail: the module re-issues the graph in QMM_PREC_BF16
    compare("prefill over weights beyond the f16 range (bf16 re-issue)", 40, gpu, cpu, [&](Graph & G, std::mt19937 & rng) {
        ggml_tensor * x = weight(G, rng, GGML_TYPE_F32, E, 40, 1.0f);
        ggml_tensor * W = weight(G, rng, GGML_TYPE_Q4_K, E, E, 2.0e5f);
        ggml_tensor * mm = ggml_mul_mat(G.ctx, W, x);  G.fresh(mm);
        ggml_tensor * fin = ggml_scale(G.ctx, mm, 1.0f);  G.fresh(fin);
        G.outs = { fin };
    });
    ggml_backend_free(gpu);
    ggml_backend_free(cpu);
    printf("%d OK, %d FAILED\n", n_ok, n_fail);
    return n_fail ? 1 : 0;
}
