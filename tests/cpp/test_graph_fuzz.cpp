// test_graph_fuzz.cpp — pins the plugin's graph scheduler (fusions, MUL_MAT grouping, hoists) on graphs it was NOT written against.
//
// backend_graph_compute recognises llama.cpp's layer motifs and issues several nodes as one launch, sometimes writing a later node's
// buffer early (csrc/ggml-mi355x.cpp).  The reference's test-backend-ops only ever hands it single-op graphs, so this program builds
// random transformer-block graphs on the public ggml API: llama / qwen / gemma / gpt-neox style motifs (biases behind q/k/v, QK-norm,
// parallel residual, GELU and ungated MLPs, scaled residuals, a second reader of a tensor a fusion would elide, odd head counts),
// random widths, any of the eleven quantized weight types, 1 .. 40 tokens, and allocates them with ggml's own graph allocator
// (ggml_gallocr), i.e. with the buffer reuse the fusions have to survive.  Every graph is run
//   (a) on the MI355X backend with multi-node launches on,
//   (b) on the MI355X backend one launch per node ("ggml_backend_mi355x_set_fuse" proc address),
//   (c) on the ggml CPU backend,
// and every flagged output is compared: (a) vs (b) NMSE <= 1e-6 (same kernels' arithmetic, different schedule) and (a) vs (c)
// NMSE <= 2e-3 (every MUL_MAT re-quantizes its input to int8, so a last-bit difference upstream flips roundings downstream;
// the per-op bars are held by test-backend-ops).  Public API only; GGML_BACKEND_PATH=<module>.  argv[1] = number of graphs.
#include "ggml.h"
#include "ggml-alloc.h"
#include "ggml-backend.h"
#include "ggml-cpu.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

struct Rng {
    std::mt19937 g;
    explicit Rng(uint32_t s) : g(s) {}
    int  pick(int n) { return (int) (g() % (uint32_t) n); }
    bool coin(int pct = 50) { return pick(100) < pct; }
    float uni() { return (float) (g() >> 8) * (1.0f / 8388608.0f) - 1.0f; }
};

static const ggml_type QTYPES[] = { GGML_TYPE_Q4_0, GGML_TYPE_Q4_1, GGML_TYPE_Q5_0, GGML_TYPE_Q5_1, GGML_TYPE_Q8_0, GGML_TYPE_Q2_K, GGML_TYPE_Q3_K,
                                    GGML_TYPE_Q4_K, GGML_TYPE_Q5_K, GGML_TYPE_Q6_K, GGML_TYPE_IQ4_NL };

struct Case {
    // static tensors (weights, inputs) with their host data, created identically for every backend
    struct Static { std::string name; ggml_type type; int64_t ne[2]; std::vector<uint8_t> data; };
    std::vector<Static> statics;
    std::string desc;
};

struct Built {
    ggml_context * ctx_w = nullptr, * ctx_g = nullptr;
    ggml_backend_buffer_t buf_w = nullptr;
    ggml_gallocr_t ga = nullptr;
    ggml_cgraph * graph = nullptr;
    std::vector<ggml_tensor *> outs;
};

// the recipe of one graph: drawn once, replayed per backend so that all three see the same structure and data
struct Recipe {
    uint32_t seed;
    int n_embd, n_head, n_head_kv, head_dim, n_ff, n_tokens, n_layer;
    bool bias, qk_norm, parallel_residual, gated, gelu, scale_residual, second_reader, post_norm;
    ggml_type tq, tk, tv, to, tg, tu, td;
    int n_expert, n_used;                    // > 0: the MLP is build_moe_ffn's block (router, MUL_MAT_IDs, weighted sum of the used experts)
};

static Recipe draw(uint32_t seed) {
    Rng r(seed);
    Recipe c{};
    c.seed = seed;
    c.head_dim = r.coin() ? 64 : 128;
    c.n_head = 2 + r.pick(3) * 2;                       // 2, 4, 6
    c.n_head_kv = r.coin() ? c.n_head : c.n_head / 2;
    c.n_embd = 256 * (1 + r.pick(3));                   // 256 .. 768 (multiples of 256: every quant type fits)
    c.n_ff = 256 * (2 + r.pick(4));
    const int toks[] = { 1, 1, 2, 3, 7, 8, 9, 33, 40 };
    c.n_tokens = toks[r.pick(9)];
    c.n_layer = 1 + r.pick(2);
    c.bias = r.coin(40); c.qk_norm = r.coin(35); c.parallel_residual = r.coin(25); c.gated = r.coin(70); c.gelu = r.coin(35);
    c.scale_residual = r.coin(20); c.second_reader = r.coin(30); c.post_norm = r.coin(20);
    auto qt = [&] { return QTYPES[r.pick(11)]; };
    c.tq = qt(); c.tk = r.coin(60) ? c.tq : qt(); c.tv = qt(); c.to = qt(); c.tg = qt(); c.tu = r.coin(70) ? c.tg : qt(); c.td = qt();
    if ((c.n_head * c.head_dim) % 256) {                // wo's K = heads x head_dim: the 256-weight super-block types need a multiple of 256
        const ggml_type legacy[] = { GGML_TYPE_Q4_0, GGML_TYPE_Q4_1, GGML_TYPE_Q5_0, GGML_TYPE_Q5_1, GGML_TYPE_Q8_0, GGML_TYPE_IQ4_NL };
        c.to = legacy[r.pick(6)];
    }
    // drawn last, so that the seeds of the rounds before keep their graphs: a quarter of the graphs get Mixtral's MoE block
    c.n_expert = r.coin(25) ? (r.coin() ? 8 : 4) : 0;
    c.n_used = c.n_expert ? 2 + r.pick(2) : 0;
    if (c.n_expert) { c.gated = true; c.tu = c.tg; }
    return c;
}

static std::string describe(const Recipe & c) {
    char b[576], moe[48] = "";
    if (c.n_expert) snprintf(moe, sizeof(moe), ", MoE %d of %d experts", c.n_used, c.n_expert);
    snprintf(b, sizeof(b), "seed %u: %d layer(s), embd %d, heads %d/%d x %d, ff %d, N %d, q/k/v/o %s/%s/%s/%s, gate/up/down %s/%s/%s%s%s%s%s%s%s%s%s%s",
             c.seed, c.n_layer, c.n_embd, c.n_head, c.n_head_kv, c.head_dim, c.n_ff, c.n_tokens, ggml_type_name(c.tq), ggml_type_name(c.tk),
             ggml_type_name(c.tv), ggml_type_name(c.to), ggml_type_name(c.tg), ggml_type_name(c.tu), ggml_type_name(c.td), c.bias ? ", biases" : "",
             c.qk_norm ? ", QK-norm" : "", c.parallel_residual ? ", parallel residual" : "", c.gated ? "" : ", ungated MLP", c.gelu ? ", GELU" : "",
             c.scale_residual ? ", scaled residual" : "", c.second_reader ? ", extra readers" : "", c.post_norm ? ", post-norm" : "", moe);
    return b;
}

// builds the graph for `be`; weights and inputs are generated from the recipe's seed (same bytes on every backend)
static bool build(const Recipe & c, ggml_backend_t be, Built & B) {
    Rng r(c.seed * 2654435761u + 17);
    ggml_init_params ipw = { ggml_tensor_overhead() * 256, nullptr, true };
    ggml_init_params ipg = { ggml_tensor_overhead() * 2048 + ggml_graph_overhead(), nullptr, true };
    B.ctx_w = ggml_init(ipw);
    B.ctx_g = ggml_init(ipg);
    ggml_context * cw = B.ctx_w, * cg = B.ctx_g;
    struct Pending { ggml_tensor * t; std::vector<uint8_t> bytes; };
    std::vector<Pending> pend;
    auto weight = [&](ggml_type type, int64_t k, int64_t m, float sigma) {
        ggml_tensor * t = ggml_new_tensor_2d(cw, type, k, m);
        std::vector<float> f((size_t) k * m);
        for (auto & v : f) v = r.uni() * sigma;
        std::vector<uint8_t> q(ggml_row_size(type, k) * m);
        if (type == GGML_TYPE_F32) memcpy(q.data(), f.data(), q.size());
        else ggml_quantize_chunk(type, f.data(), q.data(), 0, m, k, nullptr);
        pend.push_back({ t, std::move(q) });
        return t;
    };
    auto weight3 = [&](ggml_type type, int64_t k, int64_t m, int64_t e, float sigma) {
        ggml_tensor * t = ggml_new_tensor_3d(cw, type, k, m, e);
        std::vector<float> f((size_t) k * m * e);
        for (auto & v : f) v = r.uni() * sigma;
        std::vector<uint8_t> q(ggml_row_size(type, k) * m * e);
        ggml_quantize_chunk(type, f.data(), q.data(), 0, m * e, k, nullptr);
        pend.push_back({ t, std::move(q) });
        return t;
    };
    const int64_t E = c.n_embd, N = c.n_tokens, H = c.n_head, HK = c.n_head_kv, D = c.head_dim, F = c.n_ff;
    const float ws = 1.0f / sqrtf((float) E);
    ggml_tensor * inp  = weight(GGML_TYPE_F32, E, N, 1.0f);                  // the token activations (an input, kept with the statics)
    ggml_tensor * pos  = ggml_new_tensor_1d(cw, GGML_TYPE_I32, N);
    { std::vector<uint8_t> p(N * 4); for (int64_t i = 0; i < N; ++i) { int32_t v = (int32_t) i; memcpy(p.data() + 4 * i, &v, 4); } pend.push_back({ pos, p }); }
    ggml_tensor * mask = ggml_new_tensor_2d(cw, GGML_TYPE_F32, N, N);        // causal
    { std::vector<uint8_t> m(N * N * 4); for (int64_t i = 0; i < N; ++i) for (int64_t j = 0; j < N; ++j) { float v = j <= i ? 0.0f : -INFINITY; memcpy(m.data() + 4 * (i * N + j), &v, 4); } pend.push_back({ mask, m }); }

    ggml_tensor * cur = inp;
    auto norm = [&](ggml_tensor * x, int64_t width) {
        ggml_tensor * w = weight(GGML_TYPE_F32, width, 1, 1.0f);
        return ggml_mul(cg, ggml_rms_norm(cg, x, 1e-5f), w);
    };
    for (int il = 0; il < c.n_layer; ++il) {
        ggml_tensor * resid = cur;
        ggml_tensor * x = norm(cur, E);
        if (c.second_reader && il == 0) { ggml_tensor * o = ggml_scale(cg, x, 0.5f); ggml_set_output(o); B.outs.push_back(o); }   // the normed row has a reader outside the group
        ggml_tensor * q = ggml_mul_mat(cg, weight(c.tq, E, H * D, ws), x);
        if (c.bias) q = ggml_add(cg, q, weight(GGML_TYPE_F32, H * D, 1, 0.1f));
        ggml_tensor * k = ggml_mul_mat(cg, weight(c.tk, E, HK * D, ws), x);
        if (c.bias) k = ggml_add(cg, k, weight(GGML_TYPE_F32, HK * D, 1, 0.1f));
        ggml_tensor * v = ggml_mul_mat(cg, weight(c.tv, E, HK * D, ws), x);
        if (c.bias) v = ggml_add(cg, v, weight(GGML_TYPE_F32, HK * D, 1, 0.1f));
        q = ggml_reshape_3d(cg, q, D, H, N);
        k = ggml_reshape_3d(cg, k, D, HK, N);
        if (c.qk_norm) { q = norm(q, D); k = norm(k, D); }
        q = ggml_rope(cg, q, pos, (int) D, 0);
        k = ggml_rope(cg, k, pos, (int) D, 0);
        // attention without a cache: K and V of this batch as f16, as build_attn_mha reads them (src/llama-graph.cpp:1167-1205)
        ggml_tensor * kf = ggml_cast(cg, ggml_permute(cg, k, 0, 2, 1, 3), GGML_TYPE_F16);                                     // [D, N, HK]
        ggml_tensor * vf = ggml_cast(cg, ggml_cont(cg, ggml_permute(cg, ggml_reshape_3d(cg, v, D, HK, N), 1, 2, 0, 3)), GGML_TYPE_F16);   // [N, D, HK]
        ggml_tensor * qp = ggml_permute(cg, q, 0, 2, 1, 3);                                                                   // [D, N, H]
        ggml_tensor * kq = ggml_mul_mat(cg, kf, qp);                                                                          // [N, N, H]
        kq = ggml_soft_max_ext(cg, kq, mask, 1.0f / sqrtf((float) D), 0.0f);
        ggml_tensor * kqv = ggml_mul_mat(cg, vf, kq);                                                                         // [D, N, H]
        ggml_tensor * merged = ggml_cont_2d(cg, ggml_permute(cg, kqv, 0, 2, 1, 3), H * D, N);
        ggml_tensor * att = ggml_mul_mat(cg, weight(c.to, H * D, E, 1.0f / sqrtf((float) (H * D))), merged);
        if (c.bias) att = ggml_add(cg, att, weight(GGML_TYPE_F32, E, 1, 0.1f));
        if (c.scale_residual) att = ggml_scale(cg, att, 0.7f);
        ggml_tensor * ffn_in, * h;
        if (c.parallel_residual) { h = nullptr; ffn_in = norm(resid, E); }         // gpt-neox: both branches read the layer input
        else { h = ggml_add(cg, att, resid); ffn_in = norm(h, E); }
        ggml_tensor * f;
        if (c.n_expert) {
            // build_moe_ffn (src/llama-graph.cpp:800-917): softmax gating, top-k, normalised weights, SiLU experts, sum over the used ones
            const int64_t NE = c.n_expert, NU = c.n_used;
            ggml_tensor * lg = ggml_mul_mat(cg, weight(GGML_TYPE_F32, E, NE, 1.0f), ffn_in);
            ggml_tensor * probs = ggml_soft_max(cg, lg);
            ggml_tensor * sel = ggml_top_k(cg, probs, (int) NU);
            ggml_tensor * wts = ggml_get_rows(cg, ggml_reshape_3d(cg, probs, 1, NE, N), sel);
            wts = ggml_reshape_2d(cg, wts, NU, N);
            wts = ggml_div(cg, wts, ggml_sum_rows(cg, wts));
            wts = ggml_reshape_3d(cg, wts, 1, NU, N);
            ggml_tensor * x3 = ggml_reshape_3d(cg, ffn_in, E, 1, N);
            ggml_tensor * up = ggml_mul_mat_id(cg, weight3(c.tu, E, F, NE, ws), x3, sel);
            ggml_tensor * gt = ggml_mul_mat_id(cg, weight3(c.tg, E, F, NE, ws), x3, sel);
            gt = ggml_silu(cg, gt);
            ggml_tensor * par = ggml_mul(cg, up, gt);
            ggml_tensor * ex = ggml_mul_mat_id(cg, weight3(c.td, F, E, NE, 1.0f / sqrtf((float) F)), par, sel);
            ex = ggml_mul(cg, ex, wts);
            f = nullptr;
            for (int64_t u = 0; u < NU; ++u) {
                ggml_tensor * ce = ggml_view_2d(cg, ex, E, N, ex->nb[2], u * ex->nb[1]);
                f = u == 0 ? ce : ggml_add(cg, f, ce);
            }
        } else if (c.gated) {
            ggml_tensor * g = ggml_mul_mat(cg, weight(c.tg, E, F, ws), ffn_in);
            ggml_tensor * u = ggml_mul_mat(cg, weight(c.tu, E, F, ws), ffn_in);
            if (c.second_reader && il == c.n_layer - 1) { ggml_tensor * o = ggml_scale(cg, u, 2.0f); ggml_set_output(o); B.outs.push_back(o); }   // `up` is read twice
            g = c.gelu ? ggml_gelu(cg, g) : ggml_silu(cg, g);
            f = ggml_mul(cg, g, u);
        } else {
            f = ggml_mul_mat(cg, weight(c.tu, E, F, ws), ffn_in);
            if (c.bias) f = ggml_add(cg, f, weight(GGML_TYPE_F32, F, 1, 0.1f));
            f = c.gelu ? ggml_gelu(cg, f) : ggml_silu(cg, f);
        }
        if (!c.n_expert) f = ggml_mul_mat(cg, weight(c.td, F, E, 1.0f / sqrtf((float) F)), f);
        if (c.post_norm) f = norm(f, E);                                           // gemma-style norm behind the block
        cur = c.parallel_residual ? ggml_add(cg, ggml_add(cg, f, att), resid) : ggml_add(cg, f, h);
    }
    cur = norm(cur, E);
    ggml_tensor * logits = ggml_mul_mat(cg, weight(GGML_TYPE_Q6_K, E, 512, ws), cur);
    ggml_set_output(logits);
    B.outs.push_back(logits);

    B.graph = ggml_new_graph_custom(cg, 2048, false);
    for (ggml_tensor * o : B.outs) ggml_build_forward_expand(B.graph, o);
    B.buf_w = ggml_backend_alloc_ctx_tensors_from_buft(cw, ggml_backend_get_default_buffer_type(be));
    if (!B.buf_w) return false;
    for (auto & p : pend) ggml_backend_tensor_set(p.t, p.bytes.data(), 0, p.bytes.size());
    B.ga = ggml_gallocr_new(ggml_backend_get_default_buffer_type(be));
    return ggml_gallocr_alloc_graph(B.ga, B.graph);
}

static void destroy(Built & B) {
    if (B.ga) ggml_gallocr_free(B.ga);
    if (B.buf_w) ggml_backend_buffer_free(B.buf_w);
    if (B.ctx_g) ggml_free(B.ctx_g);
    if (B.ctx_w) ggml_free(B.ctx_w);
}

static bool run(const Recipe & c, ggml_backend_t be, std::vector<std::vector<float>> & res) {
    Built B;
    if (!build(c, be, B)) { fprintf(stderr, "build/alloc failed\n"); destroy(B); return false; }
    // the MI355X device must take every node of these graphs (a CPU fallback would test nothing)
    if (ggml_backend_graph_compute(be, B.graph) != GGML_STATUS_SUCCESS) { fprintf(stderr, "graph_compute failed\n"); destroy(B); return false; }
    res.clear();
    for (ggml_tensor * o : B.outs) {
        std::vector<float> v(ggml_nelements(o));
        ggml_backend_tensor_get(o, v.data(), 0, v.size() * sizeof(float));
        res.push_back(std::move(v));
    }
    destroy(B);
    return true;
}

static double nmse(const std::vector<float> & a, const std::vector<float> & ref) {
    double num = 0, den = 0;
    for (size_t i = 0; i < a.size(); ++i) { const double d = (double) a[i] - ref[i]; num += d * d; den += (double) ref[i] * ref[i]; }
    return den > 0 ? num / den : (num > 0 ? 1.0 : 0.0);
}

int main(int argc, char ** argv) {
    const int n_graphs = argc > 1 ? atoi(argv[1]) : 40;
    // QMM_FUZZ_PLAN_ONLY=1: the module is linked against tests/cpp/qmm_stub.cpp (no device, no arithmetic): both schedules are driven for the
    // sake of the host logic under the sanitizers (tests/test_host_sanitizers.py); results are not compared
    const bool plan_only = getenv("QMM_FUZZ_PLAN_ONLY") != nullptr;
    ggml_backend_load_all();
    ggml_backend_reg_t reg = ggml_backend_reg_by_name("MI355X");
    if (!reg) { fprintf(stderr, "MI355X backend not loaded (GGML_BACKEND_PATH?)\n"); return 2; }
    auto set_fuse = (void (*)(int)) ggml_backend_reg_get_proc_address(reg, "ggml_backend_mi355x_set_fuse");
    if (!set_fuse) { fprintf(stderr, "module has no ggml_backend_mi355x_set_fuse\n"); return 2; }
    ggml_backend_t gpu = ggml_backend_dev_init(ggml_backend_reg_dev_get(reg, 0), nullptr);
    ggml_backend_t cpu = ggml_backend_init_by_type(GGML_BACKEND_DEVICE_TYPE_CPU, nullptr);
    int n_ok = 0, n_fail = 0;
    double worst_sched = 0, worst_cpu = 0;
    for (int i = 0; i < n_graphs; ++i) {
        const Recipe c = draw(1000 + i);
        std::vector<std::vector<float>> fused, plain, ref;
        set_fuse(1);
        const bool a = run(c, gpu, fused);
        set_fuse(0);
        const bool b = run(c, gpu, plain);
        set_fuse(-1);
        if (plan_only) { printf("  %s : %s\n", describe(c).c_str(), a && b ? "planned" : "FAIL"); a && b ? ++n_ok : ++n_fail; continue; }
        const bool d = run(c, cpu, ref);
        bool ok = a && b && d && fused.size() == plain.size() && fused.size() == ref.size();
        double e_sched = 0, e_cpu = 0;
        for (size_t j = 0; ok && j < fused.size(); ++j) {
            for (float v : fused[j]) ok = ok && std::isfinite(v);
            e_sched = std::max(e_sched, nmse(fused[j], plain[j]));
            e_cpu = std::max(e_cpu, nmse(fused[j], ref[j]));
        }
        ok = ok && e_sched <= 1e-6 && e_cpu <= 2e-3;
        worst_sched = std::max(worst_sched, e_sched);
        worst_cpu = std::max(worst_cpu, e_cpu);
        printf("  %s : fused vs per-node %.1e, vs CPU %.1e : %s\n", describe(c).c_str(), e_sched, e_cpu, ok ? "OK" : "FAIL");
        ok ? ++n_ok : ++n_fail;
    }
    ggml_backend_free(gpu);
    ggml_backend_free(cpu);
    printf("worst: fused vs per-node %.2e, vs CPU %.2e\n", worst_sched, worst_cpu);
    printf("%d OK, %d FAILED\n", n_ok, n_fail);
    return n_fail ? 1 : 0;
}
