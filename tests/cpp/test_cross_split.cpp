// test_cross_split.cpp — a tensor that a multi-node launch would elide or redirect has a reader in ANOTHER scheduler split
// (ADVICE r1 / r2).  ggml_backend_sched hands this device one split = a contiguous run of nodes and, after
// graph_compute, copies every tensor a later split reads out of its t->data (ggml/src/ggml-backend.cpp:1355-1448).  The module's
// reader counts see one split only; it may skip or redirect a tensor only where it can prove that all readers are in the split
// (csrc/ggml-mi355x.cpp analyze_readers: a later node of the same cgraph reuses the tensor's memory).  Here an op the device refuses
// (GGML_OP_SQR) sits INSIDE a layer and reads
//   1. the normed row x * rms_norm * w, whose other readers are wq / wk (few tokens: the norm is formed in the mat-vec's staging);
//   2. silu(gate), whose other reader is the SwiGLU multiply (one launch with the up projection);
//   3. the soft-max probabilities of the attention chain (one launch kq -> soft_max -> kqv -> permute -> cont);
//   4. the merged-heads CONT (redirected to the scratch in llama.cpp's layers);
// through ggml_backend_sched over { MI355X, CPU }, for 1, 4 and 40 tokens, against the same graph on the CPU alone (NMSE <= 5e-4).
// Public ggml API only; GGML_BACKEND_PATH=<module>.
#include "ggml.h"
#include "ggml-alloc.h"
#include "ggml-backend.h"
#include "ggml-cpu.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

static int n_ok = 0, n_fail = 0;

struct Pending { ggml_tensor * t; std::vector<uint8_t> bytes; };

static std::vector<std::vector<float>> run(ggml_backend_t main_be, ggml_backend_t cpu, int which, int64_t N, int * n_splits) {
    ggml_init_params ipw = { ggml_tensor_overhead() * 64, nullptr, true };
    ggml_init_params ipg = { ggml_tensor_overhead() * 256 + ggml_graph_overhead(), nullptr, true };
    ggml_context * cw = ggml_init(ipw), * cg = ggml_init(ipg);
    std::mt19937 rng(7 + which);
    std::vector<Pending> pend;
    auto weight = [&](ggml_type type, int64_t k, int64_t m, float sigma) {
        ggml_tensor * t = ggml_new_tensor_2d(cw, type, k, m);
        std::uniform_real_distribution<float> u(-sigma, sigma);
        std::vector<float> f((size_t) k * m);
        for (auto & v : f) v = u(rng);
        std::vector<uint8_t> q(ggml_row_size(type, k) * m);
        if (type == GGML_TYPE_F32) memcpy(q.data(), f.data(), q.size());
        else ggml_quantize_chunk(type, f.data(), q.data(), 0, m, k, nullptr);
        pend.push_back({ t, std::move(q) });
        return t;
    };
    const int64_t E = 1024, F = 2048, D = 128, H = 8;
    ggml_tensor * x = weight(GGML_TYPE_F32, E, N, 1.0f), * nw = weight(GGML_TYPE_F32, E, 1, 1.0f);
    std::vector<ggml_tensor *> outs;
    ggml_tensor * xn = ggml_mul(cg, ggml_rms_norm(cg, x, 1e-5f), nw);
    if (which == 0) {
        ggml_tensor * q = ggml_mul_mat(cg, weight(GGML_TYPE_Q4_K, E, E, 0.05f), xn), * k = ggml_mul_mat(cg, weight(GGML_TYPE_Q4_K, E, 256, 0.05f), xn);
        ggml_tensor * z = ggml_sqr(cg, xn);                                                    // CPU: reads the normed row across the split
        outs = { ggml_add(cg, q, z), ggml_scale(cg, k, 1.0f) };
    } else if (which == 1) {
        ggml_tensor * g = ggml_mul_mat(cg, weight(GGML_TYPE_Q4_K, E, F, 0.05f), xn), * u = ggml_mul_mat(cg, weight(GGML_TYPE_Q4_K, E, F, 0.05f), xn);
        ggml_tensor * sl = ggml_silu(cg, g);
        ggml_tensor * p = ggml_mul(cg, sl, u);
        ggml_tensor * dn = ggml_mul_mat(cg, weight(GGML_TYPE_Q4_K, F, E, 0.05f), p);
        ggml_tensor * z = ggml_sqr(cg, sl);                                                    // CPU: reads silu(gate)
        outs = { dn, ggml_scale(cg, z, 1.0f) };
    } else {
        const int64_t n_kv = 256;
        ggml_tensor * qf = ggml_mul_mat(cg, weight(GGML_TYPE_Q4_K, E, D * H, 0.05f), xn);
        ggml_tensor * kc = weight(GGML_TYPE_F16, D, n_kv * 2, 1.0f), * vc = weight(GGML_TYPE_F16, n_kv, D * 2, 1.0f), * mk = weight(GGML_TYPE_F32, n_kv, N, 1.0f);
        ggml_tensor * qp = ggml_permute(cg, ggml_reshape_3d(cg, qf, D, H, N), 0, 2, 1, 3);
        ggml_tensor * kq = ggml_mul_mat(cg, ggml_view_3d(cg, kc, D, n_kv, 2, D * 2, D * n_kv * 2, 0), qp);
        ggml_tensor * sm = ggml_soft_max_ext(cg, kq, mk, 0.0883883f, 0.0f);
        ggml_tensor * kqv = ggml_mul_mat(cg, ggml_view_3d(cg, vc, n_kv, D, 2, n_kv * 2, n_kv * D * 2, 0), sm);
        ggml_tensor * ct = ggml_cont_2d(cg, ggml_permute(cg, kqv, 0, 2, 1, 3), D * H, N);
        ggml_tensor * wo = ggml_mul_mat(cg, weight(GGML_TYPE_Q4_K, D * H, E, 0.05f), ct);
        ggml_tensor * z = which == 2 ? ggml_sqr(cg, sm) : ggml_sqr(cg, ct);                   // CPU: the probabilities / the merged heads
        outs = { wo, ggml_scale(cg, z, 1.0f) };
    }
    for (ggml_tensor * o : outs) ggml_set_output(o);
    ggml_cgraph * g = ggml_new_graph(cg);
    for (ggml_tensor * o : outs) ggml_build_forward_expand(g, o);
    ggml_backend_buffer_t bw = ggml_backend_alloc_ctx_tensors(cw, main_be);
    for (auto & p : pend) ggml_backend_tensor_set(p.t, p.bytes.data(), 0, p.bytes.size());
    ggml_backend_t bes[2] = { main_be, cpu };
    const int nbe = main_be == cpu ? 1 : 2;
    ggml_backend_sched_t sched = ggml_backend_sched_new(nbe == 2 ? bes : &cpu, nullptr, nbe, 512, false);
    std::vector<std::vector<float>> res;
    if (!ggml_backend_sched_alloc_graph(sched, g) || ggml_backend_sched_graph_compute(sched, g) != GGML_STATUS_SUCCESS) {
        fprintf(stderr, "sched failed\n");
        exit(2);
    }
    if (n_splits) *n_splits = ggml_backend_sched_get_n_splits(sched);
    for (ggml_tensor * o : outs) { std::vector<float> v(ggml_nelements(o)); ggml_backend_tensor_get(o, v.data(), 0, v.size() * 4); res.push_back(v); }
    ggml_backend_sched_free(sched);
    ggml_backend_buffer_free(bw);
    ggml_free(cg);
    ggml_free(cw);
    return res;
}

int main() {
    ggml_backend_load_all();
    ggml_backend_reg_t reg = ggml_backend_reg_by_name("MI355X");
    if (!reg) { fprintf(stderr, "MI355X backend not loaded (GGML_BACKEND_PATH?)\n"); return 2; }
    ggml_backend_t gpu = ggml_backend_dev_init(ggml_backend_reg_dev_get(reg, 0), nullptr);
    ggml_backend_t cpu = ggml_backend_init_by_type(GGML_BACKEND_DEVICE_TYPE_CPU, nullptr);
    const char * what[] = { "CPU op reads the normed row folded into wq / wk", "CPU op reads silu(gate) of the SwiGLU launch",
                            "CPU op reads the attention launch's probabilities", "CPU op reads the merged heads (scratch redirect)" };
    for (int64_t N : { (int64_t) 1, (int64_t) 4, (int64_t) 40 })
        for (int which = 0; which < 4; ++which) {
            int splits = 0;
            const auto a = run(gpu, cpu, which, N, &splits), b = run(cpu, cpu, which, N, nullptr);
            bool ok = a.size() == b.size() && splits >= 2;                  // the refused op must really have made a second split
            double worst = 0;
            for (size_t j = 0; ok && j < a.size(); ++j) {
                double num = 0, den = 0;
                for (size_t i = 0; i < a[j].size(); ++i) { const double d = (double) a[j][i] - b[j][i]; num += d * d; den += (double) b[j][i] * b[j][i]; ok = ok && std::isfinite(a[j][i]); }
                worst = std::max(worst, den > 0 ? num / den : num);
            }
            ok = ok && worst <= 5e-4;
            printf("  %-52s N=%-3lld splits %d nmse %.1e : %s\n", what[which], (long long) N, splits, worst, ok ? "OK" : "FAIL");
            ok ? ++n_ok : ++n_fail;
        }
    ggml_backend_free(gpu);
    ggml_backend_free(cpu);
    printf("%d OK, %d FAILED\n", n_ok, n_fail);
    return n_fail ? 1 : 0;
}
