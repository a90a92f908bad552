// test_async_events.cpp — the plugin's asynchronous interface (ggml-backend-impl.h:93-95, 114-116, 176-178) through
// ggml's public API, the way the scheduler uses it for pipelined splits (ggml-backend.cpp:1363-1396):
//   set_tensor_async(x) -> graph_compute_async(MUL_MAT) on backend A -> event_record(A) -> event_wait(B) ->
//   tensor_copy_async(A -> B) -> get_tensor_async(B) -> synchronize(B)
// The result read on B must equal the CPU backend's MUL_MAT within the reference harness' bar (NMSE <= 5e-4), and the
// device must advertise caps.async and caps.events.  Run with GGML_BACKEND_PATH=<module> GGML_MI355X_VIRTUAL_DEVICES=2.
#include "ggml.h"
#include "ggml-backend.h"
#include "ggml-cpu.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

static int fails = 0;
#define CHECK(c, ...) do { if (!(c)) { printf("FAIL: " __VA_ARGS__); printf("\n"); ++fails; } } while (0)

int main() {
    ggml_backend_load_all();
    ggml_backend_reg_t reg = ggml_backend_reg_by_name("MI355X");
    if (!reg || ggml_backend_reg_dev_count(reg) < 2) { fprintf(stderr, "need the MI355X module with >= 2 (virtual) devices\n"); return 2; }
    ggml_backend_dev_t devA = ggml_backend_reg_dev_get(reg, 0), devB = ggml_backend_reg_dev_get(reg, 1);
    ggml_backend_dev_props pa;
    ggml_backend_dev_get_props(devA, &pa);
    CHECK(pa.caps.async && pa.caps.events, "caps.async / caps.events not advertised");
    ggml_backend_t A = ggml_backend_dev_init(devA, nullptr), B = ggml_backend_dev_init(devB, nullptr);
    ggml_backend_t cpu = ggml_backend_init_by_type(GGML_BACKEND_DEVICE_TYPE_CPU, nullptr);
    ggml_backend_event_t ev = ggml_backend_event_new(devA);
    CHECK(ev != nullptr, "event_new returned NULL");

    std::mt19937 rng(7);
    std::uniform_real_distribution<float> u(-1.0f, 1.0f);
    const ggml_type types[] = { GGML_TYPE_Q4_K, GGML_TYPE_Q6_K, GGML_TYPE_Q8_0 };
    const int64_t batches[] = { 1, 40, 300 };
    int n_ok = 0;
    for (ggml_type type : types) for (int64_t N : batches) {
        const int64_t M = 1536, K = 1024;
        std::vector<float> wf((size_t) M * K), x((size_t) N * K);
        for (auto & v : wf) v = u(rng);
        for (auto & v : x) v = u(rng);
        std::vector<uint8_t> wq(ggml_row_size(type, K) * M);
        ggml_quantize_chunk(type, wf.data(), wq.data(), 0, M, K, nullptr);

        auto build = [&](ggml_backend_buffer_type_t buft, ggml_context ** ctx, ggml_tensor ** w, ggml_tensor ** xt, ggml_tensor ** out, ggml_cgraph ** g) {
            ggml_init_params ip = { ggml_tensor_overhead() * 8 + ggml_graph_overhead(), nullptr, true };
            *ctx = ggml_init(ip);
            *w = ggml_new_tensor_2d(*ctx, type, K, M);
            *xt = ggml_new_tensor_2d(*ctx, GGML_TYPE_F32, K, N);
            *out = ggml_mul_mat(*ctx, *w, *xt);
            *g = ggml_new_graph(*ctx);
            ggml_build_forward_expand(*g, *out);
            return ggml_backend_alloc_ctx_tensors_from_buft(*ctx, buft);
        };
        ggml_context * cc, * ca, * cb;
        ggml_tensor * w, * xt, * out, * w2, * x2, * out2;
        ggml_cgraph * g, * g2;
        // CPU reference
        ggml_backend_buffer_t bufc = build(ggml_backend_get_default_buffer_type(cpu), &cc, &w, &xt, &out, &g);
        ggml_backend_tensor_set(w, wq.data(), 0, wq.size());
        ggml_backend_tensor_set(xt, x.data(), 0, x.size() * sizeof(float));
        ggml_backend_graph_compute(cpu, g);
        std::vector<float> ref((size_t) M * N), got((size_t) M * N, -1.0f);
        ggml_backend_tensor_get(out, ref.data(), 0, ref.size() * sizeof(float));
        // device A computes, device B receives
        ggml_backend_buffer_t bufa = build(ggml_backend_dev_buffer_type(devA), &ca, &w2, &x2, &out2, &g2);
        ggml_init_params ipb = { ggml_tensor_overhead() * 2, nullptr, true };
        cb = ggml_init(ipb);
        ggml_tensor * tb = ggml_new_tensor_2d(cb, GGML_TYPE_F32, M, N);
        ggml_backend_buffer_t bufb = ggml_backend_alloc_ctx_tensors_from_buft(cb, ggml_backend_dev_buffer_type(devB));
        ggml_backend_tensor_set(w2, wq.data(), 0, wq.size());
        ggml_backend_tensor_set_async(A, x2, x.data(), 0, x.size() * sizeof(float));
        CHECK(ggml_backend_graph_compute_async(A, g2) == GGML_STATUS_SUCCESS, "graph_compute_async failed");
        ggml_backend_event_record(ev, A);
        ggml_backend_event_wait(B, ev);
        ggml_backend_tensor_copy_async(A, B, out2, tb);
        ggml_backend_tensor_get_async(B, tb, got.data(), 0, got.size() * sizeof(float));
        ggml_backend_synchronize(B);
        ggml_backend_event_synchronize(ev);
        double num = 0, den = 0;
        for (size_t i = 0; i < ref.size(); ++i) { num += (double) (got[i] - ref[i]) * (got[i] - ref[i]); den += (double) ref[i] * ref[i]; }
        const double nmse = num / den;
        printf("  %-5s N=%-4lld  A computes, B receives: nmse_vs_cpu %.2e : %s\n", ggml_type_name(type), (long long) N, nmse, nmse <= 5e-4 ? "OK" : "FAIL");
        if (nmse <= 5e-4) ++n_ok; else ++fails;
        ggml_backend_buffer_free(bufb); ggml_backend_buffer_free(bufa); ggml_backend_buffer_free(bufc);
        ggml_free(cb); ggml_free(ca); ggml_free(cc);
    }
    ggml_backend_event_free(ev);
    ggml_backend_free(B); ggml_backend_free(A); ggml_backend_free(cpu);
    printf("%d OK, %d FAILED\n", n_ok, fails);
    return fails ? 1 : 0;
}
