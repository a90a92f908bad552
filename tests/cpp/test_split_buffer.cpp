// test_split_buffer.cpp — drives the plugin's row split (get_proc_address("ggml_backend_split_buffer_type")) the way
// llama.cpp does with -sm row (src/llama-model.cpp:316-346): weights in the split buffer type, src1/dst in the root
// device's buffer, one MUL_MAT node computed by the root device's backend; the result is compared with
//   (1) the CPU backend on the same quantized weights (NMSE, the bar of tests/test-backend-ops.cpp:1982-1984), and
//   (2) the same product with the weights in ONE device buffer (the split must not change a single row's value when the
//       kernel choice is the same: checked bit for bit for N <= 8, where every row is an independent wave).
// Uses only ggml's public API.  Built by oracle/Makefile into oracle/_ref/ (it links the reference's libggml*.so).
// Run with GGML_BACKEND_PATH=<libggml-mi355x.so> GGML_MI355X_VIRTUAL_DEVICES=<n> on a one-GPU box.
#include "ggml.h"
#include "ggml-backend.h"
#include "ggml-cpu.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

typedef ggml_backend_buffer_type_t (*split_buft_fn)(int main_device, const float * tensor_split);

static std::vector<float> run(ggml_backend_t backend, ggml_backend_buffer_type_t w_buft, ggml_backend_buffer_type_t c_buft,
                              ggml_type type, int64_t M, int64_t K, int64_t N, const std::vector<uint8_t> & wq, const std::vector<float> & x) {
    ggml_init_params ip = { ggml_tensor_overhead() * 8 + ggml_graph_overhead(), nullptr, true };
    ggml_context * cw = ggml_init(ip), * cc = ggml_init(ip);
    ggml_tensor * w = ggml_new_tensor_2d(cw, type, K, M);
    ggml_set_name(w, "w");
    ggml_backend_buffer_t bw = ggml_backend_alloc_ctx_tensors_from_buft(cw, w_buft);
    ggml_tensor * xt = ggml_new_tensor_2d(cc, GGML_TYPE_F32, K, N);
    ggml_tensor * out = ggml_mul_mat(cc, w, xt);
    ggml_set_name(out, "out");
    ggml_backend_buffer_t bc = ggml_backend_alloc_ctx_tensors_from_buft(cc, c_buft);
    if (!bw || !bc) { fprintf(stderr, "buffer allocation failed\n"); exit(2); }
    ggml_backend_tensor_set(w, wq.data(), 0, wq.size());
    ggml_backend_tensor_set(xt, x.data(), 0, x.size() * sizeof(float));
    ggml_cgraph * g = ggml_new_graph(cc);
    ggml_build_forward_expand(g, out);
    if (ggml_backend_graph_compute(backend, g) != GGML_STATUS_SUCCESS) { fprintf(stderr, "graph_compute failed\n"); exit(2); }
    std::vector<float> r((size_t) M * N);
    ggml_backend_tensor_get(out, r.data(), 0, r.size() * sizeof(float));
    // read the weights back through the split buffer: must be the bytes that went in
    std::vector<uint8_t> back(wq.size());
    ggml_backend_tensor_get(w, back.data(), 0, back.size());
    if (memcmp(back.data(), wq.data(), wq.size()) != 0) { fprintf(stderr, "get_tensor(w) differs from set_tensor(w)\n"); exit(2); }
    ggml_backend_buffer_free(bc);
    ggml_backend_buffer_free(bw);
    ggml_free(cc);
    ggml_free(cw);
    return r;
}

int main() {
    ggml_backend_load_all();
    ggml_backend_reg_t reg = ggml_backend_reg_by_name("MI355X");
    if (!reg) { fprintf(stderr, "MI355X backend not loaded (GGML_BACKEND_PATH?)\n"); return 2; }
    const int ndev = (int) ggml_backend_reg_dev_count(reg);
    auto fn = (split_buft_fn) ggml_backend_reg_get_proc_address(reg, "ggml_backend_split_buffer_type");
    if (!fn) { fprintf(stderr, "ggml_backend_split_buffer_type not exported\n"); return 2; }
    printf("MI355X devices: %d\n", ndev);
    ggml_backend_t cpu = ggml_backend_init_by_type(GGML_BACKEND_DEVICE_TYPE_CPU, nullptr);
    const ggml_type types[] = { GGML_TYPE_Q4_0, GGML_TYPE_Q8_0, GGML_TYPE_Q4_K, GGML_TYPE_Q5_K, GGML_TYPE_Q6_K };
    const float splits[][8] = { { 0 }, { 3, 1, 2, 1, 1, 1, 1, 1 }, { 1, 0, 1, 0, 0, 0, 0, 0 } };     // equal shares; uneven; a device with no rows
    const int64_t shapes[][2] = { { 1000, 512 }, { 4096, 1024 } };                                       // M (ragged vs the 64-row rounding), K
    const int64_t batches[] = { 1, 5, 33, 300 };
    int n_ok = 0, n_fail = 0;
    std::mt19937 rng(42);
    for (int root = 0; root < ndev && root < 2; ++root) {
        ggml_backend_dev_t dev = ggml_backend_reg_dev_get(reg, root);
        ggml_backend_t be = ggml_backend_dev_init(dev, nullptr);
        ggml_backend_buffer_type_t dev_buft = ggml_backend_dev_buffer_type(dev);
        for (const auto & sp : splits) {
            ggml_backend_buffer_type_t sbuft = fn(root, sp);
            if (!sbuft || !ggml_backend_dev_supports_buft(dev, sbuft)) { fprintf(stderr, "split buft unusable on device %d\n", root); return 2; }
            for (ggml_type type : types) for (const auto & sh : shapes) for (int64_t N : batches) {
                const int64_t M = sh[0], K = sh[1];
                std::uniform_real_distribution<float> u(-1.0f, 1.0f);
                std::vector<float> wf((size_t) M * K), x((size_t) N * K);
                for (auto & v : wf) v = u(rng);
                for (auto & v : x) v = u(rng);
                std::vector<uint8_t> wq(ggml_row_size(type, K) * M);
                ggml_quantize_chunk(type, wf.data(), wq.data(), 0, M, K, nullptr);
                const std::vector<float> ref = run(cpu, ggml_backend_get_default_buffer_type(cpu), ggml_backend_get_default_buffer_type(cpu), type, M, K, N, wq, x);
                const std::vector<float> one = run(be, dev_buft, dev_buft, type, M, K, N, wq, x);
                const std::vector<float> spl = run(be, sbuft, dev_buft, type, M, K, N, wq, x);
                double num = 0, den = 0, num1 = 0;
                size_t diff_bits = 0;
                for (size_t i = 0; i < ref.size(); ++i) {
                    num += (double) (spl[i] - ref[i]) * (spl[i] - ref[i]);
                    num1 += (double) (spl[i] - one[i]) * (spl[i] - one[i]);
                    den += (double) ref[i] * ref[i];
                    diff_bits += memcmp(&spl[i], &one[i], 4) != 0;
                }
                const double nmse = num / den, rel1 = std::sqrt(num1 / den);
                const bool ok = nmse <= 5e-4 && rel1 <= 1e-3 && (N > 8 || diff_bits == 0);
                printf("  root %d split {%g,%g,%g..} %-5s M=%lld K=%lld N=%-4lld nmse_vs_cpu %.2e  rel_vs_unsplit %.2e  differing %zu : %s\n",
                       root, sp[0], sp[1], sp[2], ggml_type_name(type), (long long) M, (long long) K, (long long) N, nmse, rel1, diff_bits, ok ? "OK" : "FAIL");
                ok ? ++n_ok : ++n_fail;
            }
        }
        ggml_backend_free(be);
    }
    ggml_backend_free(cpu);
    printf("%d OK, %d FAILED\n", n_ok, n_fail);
    return n_fail ? 1 : 0;
}
