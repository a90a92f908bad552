// test_moe_pair.cpp — two MUL_MAT_ID nodes on the same src1 and ids, consecutive in one graph (ffn_up_exps / ffn_gate_exps of
// llama.cpp's MoE block): the plugin issues them as one qmm_mul_mat_id_pair call.  Both results are compared with the CPU
// backend's (NMSE <= 5e-4, the bar of tests/test-backend-ops.cpp:2075-2077), for a mat-vec sized batch and a prefill sized one.
// Public ggml API only; run with GGML_BACKEND_PATH=<module>.
#include "ggml.h"
#include "ggml-backend.h"
#include "ggml-cpu.h"

#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

int main() {
    ggml_backend_load_all();
    ggml_backend_reg_t reg = ggml_backend_reg_by_name("MI355X");
    if (!reg) { fprintf(stderr, "MI355X backend not loaded (GGML_BACKEND_PATH?)\n"); return 2; }
    ggml_backend_dev_t dev = ggml_backend_reg_dev_get(reg, 0);
    ggml_backend_t gpu = ggml_backend_dev_init(dev, nullptr);
    ggml_backend_t cpu = ggml_backend_init_by_type(GGML_BACKEND_DEVICE_TYPE_CPU, nullptr);
    std::mt19937 rng(11);
    std::uniform_real_distribution<float> u(-1.0f, 1.0f);
    const ggml_type types[] = { GGML_TYPE_Q4_K, GGML_TYPE_Q6_K, GGML_TYPE_Q4_0 };
    const int64_t K = 512, M = 192, n_expert = 8, n_used = 2;
    int n_ok = 0, n_fail = 0;
    for (ggml_type type : types) for (int64_t n_tokens : { (int64_t) 1, (int64_t) 5, (int64_t) 130 }) {
        std::vector<float> wf((size_t) 2 * n_expert * M * K), x((size_t) n_tokens * K);
        for (auto & v : wf) v = u(rng);
        for (auto & v : x) v = u(rng);
        const size_t wbytes = ggml_row_size(type, K) * M * n_expert;
        std::vector<uint8_t> wq(2 * wbytes);
        ggml_quantize_chunk(type, wf.data(), wq.data(), 0, 2 * n_expert * M, K, nullptr);
        std::vector<int32_t> ids((size_t) n_tokens * n_used);
        for (int64_t t = 0; t < n_tokens; ++t) { const int e = (int) (rng() % n_expert); ids[t * n_used] = e; ids[t * n_used + 1] = (e + 1 + (int) (rng() % (n_expert - 1))) % n_expert; }
        std::vector<float> res[2][2];
        for (int which = 0; which < 2; ++which) {
            ggml_backend_t be = which ? gpu : cpu;
            ggml_init_params ip = { ggml_tensor_overhead() * 16 + ggml_graph_overhead(), nullptr, true };
            ggml_context * ctx = ggml_init(ip);
            ggml_tensor * up   = ggml_new_tensor_3d(ctx, type, K, M, n_expert);
            ggml_tensor * gate = ggml_new_tensor_3d(ctx, type, K, M, n_expert);
            ggml_tensor * b    = ggml_new_tensor_3d(ctx, GGML_TYPE_F32, K, 1, n_tokens);
            ggml_tensor * idt  = ggml_new_tensor_2d(ctx, GGML_TYPE_I32, n_used, n_tokens);
            ggml_tensor * o_up   = ggml_mul_mat_id(ctx, up, b, idt);
            ggml_tensor * o_gate = ggml_mul_mat_id(ctx, gate, b, idt);
            ggml_cgraph * g = ggml_new_graph(ctx);
            ggml_build_forward_expand(g, o_up);
            ggml_build_forward_expand(g, o_gate);
            ggml_backend_buffer_t buf = ggml_backend_alloc_ctx_tensors_from_buft(ctx, ggml_backend_get_default_buffer_type(be));
            if (!buf) { fprintf(stderr, "alloc failed\n"); return 2; }
            ggml_backend_tensor_set(up, wq.data(), 0, wbytes);
            ggml_backend_tensor_set(gate, wq.data() + wbytes, 0, wbytes);
            ggml_backend_tensor_set(b, x.data(), 0, x.size() * sizeof(float));
            ggml_backend_tensor_set(idt, ids.data(), 0, ids.size() * sizeof(int32_t));
            if (ggml_backend_graph_compute(be, g) != GGML_STATUS_SUCCESS) { fprintf(stderr, "graph_compute failed\n"); return 2; }
            ggml_tensor * outs[2] = { o_up, o_gate };
            for (int j = 0; j < 2; ++j) {
                res[which][j].resize((size_t) M * n_used * n_tokens);
                ggml_backend_tensor_get(outs[j], res[which][j].data(), 0, res[which][j].size() * sizeof(float));
            }
            ggml_backend_buffer_free(buf);
            ggml_free(ctx);
        }
        for (int j = 0; j < 2; ++j) {
            double num = 0, den = 0;
            for (size_t i = 0; i < res[0][j].size(); ++i) { const double d = res[1][j][i] - res[0][j][i]; num += d * d; den += (double) res[0][j][i] * res[0][j][i]; }
            const bool ok = num / den <= 5e-4;
            printf("  %-5s n_tokens=%-4lld %s: nmse_vs_cpu %.2e : %s\n", ggml_type_name(type), (long long) n_tokens, j ? "gate" : "up", num / den, ok ? "OK" : "FAIL");
            ok ? ++n_ok : ++n_fail;
        }
    }
    ggml_backend_free(gpu);
    ggml_backend_free(cpu);
    printf("%d OK, %d FAILED\n", n_ok, n_fail);
    return n_fail ? 1 : 0;
}
