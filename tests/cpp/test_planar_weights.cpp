// test_planar_weights.cpp — SURVEY 8f-2 through the plugin: Q4_0 / Q8_0 / Q6_K weights are re-laid into aligned planes at their
// first MUL_MAT, in place (the AMX buffer type's precedent converts at set_tensor, ggml/src/ggml-cpu/amx/amx.cpp).  What the rest of
// ggml sees must not change: (1) results equal the CPU backend's (NMSE <= 5e-4, tests/test-backend-ops.cpp:1982-1984) and are
// identical from run to run; (2) ggml_backend_tensor_get returns the GGUF bytes that were set, also AFTER the tensor was used;
// (3) a partial ggml_backend_tensor_set into a used tensor (llama.cpp's pipelined loader writes in byte chunks) lands in wire
// layout and the next MUL_MAT sees the patched weights; (4) a device-to-device tensor copy carries wire bytes.
// The first line printed is a checksum of every result: the Python test runs this twice, GGML_MI355X_REPACK=1 and =0, and wants
// the same number (the planar kernels are bit-identical to the wire ones).  Public ggml API only; GGML_BACKEND_PATH=<module>.
#include "ggml.h"
#include "ggml-backend.h"
#include "ggml-cpu.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

static int n_ok = 0, n_fail = 0;
static void check(bool ok, const char * what, ggml_type t, long long n) {
    printf("  %-5s N=%-4lld %-46s : %s\n", ggml_type_name(t), n, what, ok ? "OK" : "FAIL");
    ok ? ++n_ok : ++n_fail;
}

int main() {
    ggml_backend_load_all();
    ggml_backend_reg_t reg = ggml_backend_reg_by_name("MI355X");
    if (!reg) { fprintf(stderr, "MI355X backend not loaded (GGML_BACKEND_PATH?)\n"); return 2; }
    ggml_backend_t gpu = ggml_backend_dev_init(ggml_backend_reg_dev_get(reg, 0), nullptr);
    ggml_backend_t cpu = ggml_backend_init_by_type(GGML_BACKEND_DEVICE_TYPE_CPU, nullptr);
    std::mt19937 rng(5);
    std::uniform_real_distribution<float> u(-1.0f, 1.0f);
    uint64_t checksum = 1469598103934665603ull;
    const ggml_type types[] = { GGML_TYPE_Q4_0, GGML_TYPE_Q8_0, GGML_TYPE_Q6_K, GGML_TYPE_Q4_K };       // Q4_K: never repacked, the control
    const int64_t K = 2048, M = 96;
    for (ggml_type type : types) for (int64_t N : { (int64_t) 1, (int64_t) 40 }) {
        std::vector<float> wf((size_t) M * K), wf2((size_t) M * K), x((size_t) N * K);
        for (auto & v : wf) v = u(rng);
        for (auto & v : wf2) v = u(rng);
        for (auto & v : x) v = u(rng);
        const size_t rb = ggml_row_size(type, K), wbytes = rb * M;
        std::vector<uint8_t> wq(wbytes), wq2(wbytes), back(wbytes);
        ggml_quantize_chunk(type, wf.data(), wq.data(), 0, M, K, nullptr);
        ggml_quantize_chunk(type, wf2.data(), wq2.data(), 0, M, K, nullptr);
        // patched weights: bytes [lo, hi) of the tensor replaced, a range that cuts rows on both sides
        const size_t lo = rb * 10 + 6, hi = rb * 51 + rb / 2;
        std::vector<uint8_t> wpatched = wq;
        memcpy(wpatched.data() + lo, wq2.data() + lo, hi - lo);

        auto run = [&](ggml_backend_t be, const std::vector<uint8_t> & w0, bool patch, std::vector<float> & out, std::vector<float> & out_patched,
                       std::vector<uint8_t> * readback, std::vector<uint8_t> * copyback) {
            ggml_init_params ip = { ggml_tensor_overhead() * 16 + ggml_graph_overhead(), nullptr, true };
            ggml_context * ctx = ggml_init(ip);
            ggml_tensor * w = ggml_new_tensor_2d(ctx, type, K, M);
            ggml_tensor * w_copy = ggml_new_tensor_2d(ctx, type, K, M);
            ggml_tensor * b = ggml_new_tensor_2d(ctx, GGML_TYPE_F32, K, N);
            ggml_tensor * o = ggml_mul_mat(ctx, w, b);
            ggml_cgraph * g = ggml_new_graph(ctx);
            ggml_build_forward_expand(g, o);
            ggml_backend_buffer_t buf = ggml_backend_alloc_ctx_tensors_from_buft(ctx, ggml_backend_get_default_buffer_type(be));
            if (!buf) { fprintf(stderr, "alloc failed\n"); exit(2); }
            ggml_backend_tensor_set(w, w0.data(), 0, wbytes);
            ggml_backend_tensor_set(b, x.data(), 0, x.size() * sizeof(float));
            out.resize((size_t) M * N);
            if (ggml_backend_graph_compute(be, g) != GGML_STATUS_SUCCESS) { fprintf(stderr, "graph_compute failed\n"); exit(2); }
            ggml_backend_tensor_get(o, out.data(), 0, out.size() * sizeof(float));
            if (readback) { readback->assign(wbytes, 0); ggml_backend_tensor_get(w, readback->data(), 0, wbytes); }
            if (copyback) {                                  // used tensor -> another tensor of the same buffer -> host
                ggml_backend_tensor_copy(w, w_copy);
                copyback->assign(wbytes, 0);
                ggml_backend_tensor_get(w_copy, copyback->data(), 0, wbytes);
            }
            if (patch) {
                if (ggml_backend_graph_compute(be, g) != GGML_STATUS_SUCCESS) exit(2);          // used (planar) again before the partial write
                ggml_backend_tensor_set(w, wq2.data() + lo, lo, hi - lo);
                out_patched.resize((size_t) M * N);
                if (ggml_backend_graph_compute(be, g) != GGML_STATUS_SUCCESS) exit(2);
                ggml_backend_tensor_get(o, out_patched.data(), 0, out_patched.size() * sizeof(float));
            }
            ggml_backend_buffer_free(buf);
            ggml_free(ctx);
        };
        auto nmse = [](const std::vector<float> & a, const std::vector<float> & ref) {
            double num = 0, den = 0;
            for (size_t i = 0; i < a.size(); ++i) { const double d = a[i] - ref[i]; num += d * d; den += (double) ref[i] * ref[i]; }
            return num / den;
        };
        std::vector<float> c0, c1, g0, g1, g2, tmp;
        std::vector<uint8_t> rbk, cbk;
        run(cpu, wq, true, c0, c1, nullptr, nullptr);
        run(gpu, wq, true, g0, g1, &rbk, &cbk);
        run(gpu, wpatched, false, g2, tmp, nullptr, nullptr);
        check(nmse(g0, c0) <= 5e-4, "MUL_MAT vs CPU backend", type, N);
        check(rbk == wq, "get_tensor after use returns the bytes that were set", type, N);
        check(cbk == wq, "tensor copy of a used weight carries wire bytes", type, N);
        check(nmse(g1, c1) <= 5e-4, "partial set_tensor into a used weight vs CPU backend", type, N);
        check(g1 == g2, "... and equals a fresh upload of the patched bytes", type, N);
        for (const auto * v : { &g0, &g1 })
            for (float f : *v) { uint32_t b; memcpy(&b, &f, 4); checksum = (checksum ^ b) * 1099511628211ull; }
    }
    ggml_backend_free(gpu);
    ggml_backend_free(cpu);
    printf("checksum %016llx\n", (unsigned long long) checksum);
    printf("%d OK, %d FAILED\n", n_ok, n_fail);
    return n_fail ? 1 : 0;
}
