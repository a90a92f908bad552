// attn_dev.hip — development harness of the token-generation attention launch (csrc/qmm_ops.hip attn_decode_kernel<128, true>:
// rope(q), rope(k) -> K cache, v -> V cache, kq, softmax, kqv, merge heads in one launch): llama3-8b shape (32 heads, 8 kv heads,
// D = 128), one new token into an n_kv = 256 window.  Prints the average duration behind a producer launch (so the new rows come
// from another kernel's stores, as in the graph) and the 100 MHz phase stamps of the workgroups.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -o attn_dev profiles/tools/attn_dev.hip && ./attn_dev [n_kv]
#define ATTN_STAMPS 1
#include "../../ggml-hexagon_amd/csrc/qmm_ops.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

int qmm_internal_chain_flush(qmm_ctx *) { return 0; }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ void producer_kernel(float * q, float * k, float * v, int nq, int nk, float seed) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < nq) q[i] = __sinf(seed + 0.37f * i);
    if (i < nk) { k[i] = __cosf(seed + 0.11f * i); v[i] = __sinf(seed * 0.5f + 0.23f * i); }
}

int main(int argc, char ** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int n_kv = argc > 1 ? atoi(argv[1]) : 256, H = 32, Hk = 8, D = 128, N = 1, j0 = argc > 2 ? atoi(argv[2]) : n_kv / 2, n_ctx = 1024;
    float * q, * kraw, * vraw, * mask, * out;
    _Float16 * kc, * vc;
    int32_t * pos;
    CK(hipMalloc(&q, H * D * 4)); CK(hipMalloc(&kraw, Hk * D * 4)); CK(hipMalloc(&vraw, Hk * D * 4)); CK(hipMalloc(&mask, n_kv * 4 * 64)); CK(hipMalloc(&out, H * D * 4));
    CK(hipMalloc(&kc, (size_t) n_ctx * Hk * D * 2)); CK(hipMalloc(&vc, (size_t) n_ctx * Hk * D * 2)); CK(hipMalloc(&pos, 4));
    std::mt19937 rng(3);
    std::vector<_Float16> hc((size_t) n_ctx * Hk * D);
    for (auto & x : hc) x = (_Float16) (((int) (rng() % 2001) - 1000) / 1000.0f);
    CK(hipMemcpy(kc, hc.data(), hc.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(vc, hc.data(), hc.size() * 2, hipMemcpyHostToDevice));
    std::vector<float> hm(n_kv * 64, 0.0f);
    for (int j = j0 + 1; j < n_kv; ++j) hm[j] = -INFINITY;
    CK(hipMemcpy(mask, hm.data(), hm.size() * 4, hipMemcpyHostToDevice));
    const int32_t p0 = j0;
    CK(hipMemcpy(pos, &p0, 4, hipMemcpyHostToDevice));
    AttnArgs g;
    g.q = (const char *) q; g.k = (const char *) kc; g.v = (const char *) vc; g.mask = (const char *) mask; g.dst = (char *) out;
    g.q_nb1 = (int64_t) H * D * 4; g.q_nb2 = D * 4;
    g.k_nb1 = (int64_t) Hk * D * 2; g.k_nb2 = D * 2;                       // K cache [D, Hk, n_ctx] viewed [D, n_kv, Hk]
    g.v_nb1 = (int64_t) n_ctx * 2; g.v_nb2 = (int64_t) n_ctx * D * 2;      // V^T cache [n_ctx, D * Hk] viewed [n_kv, D, Hk]
    g.m_nb1 = (int64_t) n_kv * 4; g.d_nb1 = (int64_t) H * D * 4;
    g.D = D; g.Dv = D; g.n_kv = n_kv; g.H = H; g.gqa = H / Hk; g.scale = 0.0883883f;
    AttnFresh f;
    f.kraw = (const char *) kraw; f.vraw = (const char *) vraw;
    f.kd = (char *) (kc + (size_t) j0 * Hk * D); f.vd = (char *) (vc + j0);
    f.pos = pos; f.ff = nullptr;
    f.kraw_nbh = D * 4; f.kraw_nbn = (int64_t) Hk * D * 4; f.vraw_nbn = (int64_t) Hk * D * 4;
    f.kd_nbh = D * 2; f.kd_nbn = (int64_t) Hk * D * 2; f.vd_nbc = (int64_t) n_ctx * 2;
    memset(&f.rp, 0, sizeof(f.rp));
    f.rp.n_dims = D; f.rp.theta_scale = powf(500000.0f, -2.0f / D); f.rp.freq_scale = 1.0f; f.rp.attn_factor = 1.0f;
    f.N = N; f.j0 = j0;
    const size_t lds = (size_t) n_kv * 4 + (size_t) D * 4 + (size_t) N * (D + D) * 2;
    int which = 1;                                      // 0: attn_decode_kernel (general), 1: attn_decode_short_kernel
    auto launch = [&](float seed) {
        hipLaunchKernelGGL(producer_kernel, dim3(16), dim3(256), 0, 0, q, kraw, vraw, H * D, Hk * D, seed);
        if (which) hipLaunchKernelGGL((n_kv <= 256 ? attn_decode_short_kernel<128, true, 256> : n_kv <= 512 ? attn_decode_short_kernel<128, true, 512> : attn_decode_short_kernel<128, true, 1024>), dim3(H, N), dim3(1024), lds + (size_t) n_kv * 4, 0, g, f);
        else       hipLaunchKernelGGL((attn_decode_kernel<128, true>), dim3(H, N), dim3(1024), lds, 0, g, f);
    };
    // parity of the two kernels: output row and the stored cache rows
    std::vector<float> o0(H * D), o1(H * D);
    std::vector<_Float16> c0(hc.size()), c1(hc.size()), w0(hc.size()), w1(hc.size());
    which = 0; launch(0.7f); CK(hipDeviceSynchronize());
    CK(hipMemcpy(o0.data(), out, o0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(c0.data(), kc, c0.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(w0.data(), vc, w0.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(kc, hc.data(), hc.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(vc, hc.data(), hc.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(out, 0xff, o0.size() * 4));
    which = 1; launch(0.7f); CK(hipDeviceSynchronize());
    CK(hipMemcpy(o1.data(), out, o1.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(c1.data(), kc, c1.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(w1.data(), vc, w1.size() * 2, hipMemcpyDeviceToHost));
    {
        double mx = 0, rms = 0; size_t kd = 0, vd = 0;
        for (size_t i = 0; i < o0.size(); ++i) { mx = std::max(mx, (double) fabsf(o0[i] - o1[i])); rms += (double) o0[i] * o0[i]; }
        for (size_t i = 0; i < c0.size(); ++i) { kd += memcmp(&c0[i], &c1[i], 2) != 0; vd += memcmp(&w0[i], &w1[i], 2) != 0; }
        rms = sqrt(rms / o0.size());
        // host reference from the stored caches (both kernels stored the same rows when the counts below are 0): q roped in double,
        // rounded to f16; scores, softmax in double; p rounded to f16; V product in double
        std::vector<float> hq(H * D), hkr(Hk * D), hvr(Hk * D);
        CK(hipMemcpy(hq.data(), q, hq.size() * 4, hipMemcpyDeviceToHost));
        double e0 = 0, e1 = 0; int worst_h = -1; double worst_d = 0;
        for (int hh = 0; hh < H; ++hh) {
            const int hk2 = hh / (H / Hk);
            std::vector<double> qr(D), sc2(n_kv), pr2(n_kv);
            for (int pI = 0; pI < D / 2; ++pI) {
                float th = (float) p0; for (int k2 = 0; k2 < pI; ++k2) th *= f.rp.theta_scale;
                const double c = cos((double) th), sn = sin((double) th), a = hq[hh * D + 2 * pI], b = hq[hh * D + 2 * pI + 1];
                qr[2 * pI] = (double) (_Float16) (float) (a * c - b * sn); qr[2 * pI + 1] = (double) (_Float16) (float) (a * sn + b * c);
            }
            double mx2 = -1e30;
            for (int j = 0; j < n_kv; ++j) {
                double sdot = 0; for (int e = 0; e < D; ++e) sdot += (double) c1[((size_t) j * Hk + hk2) * D + e] * qr[e];
                sc2[j] = sdot * g.scale + (double) hm[j]; mx2 = std::max(mx2, sc2[j]);
            }
            double sum2 = 0; for (int j = 0; j < n_kv; ++j) { pr2[j] = exp(sc2[j] - mx2); sum2 += pr2[j]; }
            for (int j = 0; j < n_kv; ++j) pr2[j] = (double) (_Float16) (float) (pr2[j] / sum2);
            for (int d2 = 0; d2 < D; ++d2) {
                double o = 0; for (int j = 0; j < n_kv; ++j) o += pr2[j] * (double) w1[(size_t) (hk2 * D + d2) * n_ctx + j];
                e0 = std::max(e0, fabs(o - o0[hh * D + d2])); e1 = std::max(e1, fabs(o - o1[hh * D + d2]));
                const double dd = fabs((double) o0[hh * D + d2] - o1[hh * D + d2]);
                if (dd > worst_d) { worst_d = dd; worst_h = hh; }
            }
        }
        printf("against a double-precision host restatement: general kernel max |d| / rms %.3e, short kernel %.3e; largest kernel-to-kernel difference in head %d\n", e0 / rms, e1 / rms, worst_h);
        printf("short vs general kernel: max |d| / rms %.3e (rms %.4f), K cache halves differing %zu, V cache halves differing %zu\n", mx / rms, rms, kd, vd);
    }
    for (int i = 0; i < 20; ++i) launch(0.1f * i);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 200;
    for (int round = 0; round < 6; ++round) {
        which = round & 1;
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) launch(0.01f * i);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        float ms2;
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(producer_kernel, dim3(16), dim3(256), 0, 0, q, kraw, vraw, H * D, Hk * D, 0.01f * i);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms2, e0, e1));
        printf("round %d %s: producer + attention %.2f us per pair, producer alone %.2f us -> attention slot %.2f us\n", round, which ? "short  " : "general", ms * 1e3 / reps, ms2 * 1e3 / reps, (ms - ms2) * 1e3 / reps);
    }
    for (which = 0; which < 2; ++which) {
    launch(0.5f); CK(hipDeviceSynchronize());
    unsigned long long st[64][8];
    CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(attn_stamps), sizeof(st)));
    unsigned long long t0 = ~0ull;
    for (int w = 0; w < H; ++w) t0 = std::min(t0, st[w][0]);
    const char * names[8] = { "entry", "prep done", "barrier 1", "scores done", "barrier 2", "softmax done", "end", "loads issued" };
    printf("%s kernel stamps (us from the first workgroup's entry; thread 0 of each workgroup): median / max over %d workgroups\n", which ? "short" : "general", H);
    for (int s = 0; s < (which ? 8 : 7); ++s) {
        std::vector<double> v;
        for (int w = 0; w < H; ++w) v.push_back((double) (st[w][s] - t0) / 100.0);
        std::sort(v.begin(), v.end());
        printf("  %-13s %6.2f / %6.2f\n", names[s], v[H / 2], v[H - 1]);
    }
    }
    std::vector<float> ho(H * D);
    CK(hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost));
    double cs = 0; for (float x : ho) cs += x;
    printf("checksum %.6f\n", cs);
    return 0;
}
