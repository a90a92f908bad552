// r64s_dev.hip — development harness of the hand-placed Q4_K prefill K-step (csrc/qmm_mfma_r64s.hiph): runs mfma_r64_q4k_kernel<8>
// (compiler-scheduled) and mfma_r64s_q4k_kernel on the same random Q4_K rows and prepared activations, requires identical bits, and
// times both interleaved in one process (cdna_hip_programming.md rule 24), on rotating weight sets (no Infinity-Cache reuse).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -o r64s_dev profiles/tools/r64s_dev.hip && ./r64s_dev [M K N ksplit]
#define R64S_STAMPS 1
#include "../../ggml-hexagon_amd/csrc/qmm_mfma_r64s.hiph"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

using namespace qmm;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

static uint16_t f2h(float f) { _Float16 h = (_Float16) f; uint16_t u; memcpy(&u, &h, 2); return u; }

int main(int argc, char ** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int M = argc > 1 ? atoi(argv[1]) : 28672, K = argc > 2 ? atoi(argv[2]) : 4096, N = argc > 3 ? atoi(argv[3]) : 512;
    const int ksplit = argc > 4 ? atoi(argv[4]) : 1, sets = argc > 5 ? atoi(argv[5]) : 6, reps = 5;
    const int64_t rb = (int64_t) K / 256 * 144;
    std::mt19937 rng(1);
    std::vector<uint8_t> w((size_t) M * rb);
    for (auto & b : w) b = (uint8_t) rng();
    for (int64_t r = 0; r < M; ++r)
        for (int b = 0; b < K / 256; ++b) {
            uint16_t d = f2h(0.002f + 0.004f * (rng() % 1000) / 1000.0f), dm = f2h(0.01f + 0.02f * (rng() % 1000) / 1000.0f);
            memcpy(&w[r * rb + b * 144], &d, 2);
            memcpy(&w[r * rb + b * 144 + 2], &dm, 2);
        }
    const int Np = (N + 255) / 256 * 256;
    std::vector<uint16_t> xh((size_t) Np * K);
    for (auto & v : xh) v = f2h(((int) (rng() % 255) - 127) / 127.0f * ((rng() % 16) / 16.0f));
    std::vector<float> sc(Np);
    for (auto & v : sc) v = 0.5f + (rng() % 100) / 100.0f;
    uint8_t * dw; uint16_t * dx; float * dsc, * o1, * o2, * part;
    CK(hipMalloc(&dw, w.size() * sets + 256));
    for (int s = 0; s < sets; ++s) CK(hipMemcpy(dw + w.size() * s, w.data(), w.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&dx, xh.size() * 2)); CK(hipMemcpy(dx, xh.data(), xh.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&dsc, Np * 4)); CK(hipMemcpy(dsc, sc.data(), Np * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&o1, (size_t) N * M * 4)); CK(hipMalloc(&o2, (size_t) N * M * 4));
    CK(hipMalloc(&part, (size_t) (ksplit > 1 ? ksplit : 1) * N * M * 4));
    CK(hipMemset(o1, 0xff, (size_t) N * M * 4)); CK(hipMemset(o2, 0xee, (size_t) N * M * 4));
    RegbMore more; memset(&more, 0, sizeof(more));
    const dim3 grid((M + 255) / 256, Np / 256, ksplit);
    auto run_ref = [&](float * out, int s) {
        hipLaunchKernelGGL(mfma_r64_q4k_kernel<8>, grid, dim3(256), 0, 0, dw + w.size() * s, rb, (int64_t) 0, M, K, dx, K, dsc, nullptr, nullptr, N, out, (int64_t) M, nullptr, ksplit, part, more);
    };
    const int variant = getenv("R64S_VARIANT") ? atoi(getenv("R64S_VARIANT")) : 1;
    auto run_new = [&](float * out, int s) {
        hipLaunchKernelGGL(variant == 2 ? mfma_r64s_q4k_kernel<2> : variant & 1 ? mfma_r64s_q4k_kernel<1> : mfma_r64s_q4k_kernel<0>, grid, dim3(256), 0, 0, dw + w.size() * s, rb, (int64_t) 0, M, K, dx, K, dsc, nullptr, nullptr, N, out, (int64_t) M, nullptr, ksplit, part, more);
    };
    // parity: identical bits (split-K: the partial slabs)
    std::vector<float> a((size_t) N * M), b((size_t) N * M);
    float * src = ksplit > 1 ? part : nullptr;
    const size_t cmp = ksplit > 1 ? (size_t) ksplit * N * M : (size_t) N * M;
    std::vector<float> pa(cmp), pb(cmp);
    run_ref(o1, 0); CK(hipDeviceSynchronize());
    CK(hipMemcpy(pa.data(), src ? src : o1, cmp * 4, hipMemcpyDeviceToHost));
    if (src) CK(hipMemset(part, 0xdd, cmp * 4));
    run_new(o2, 0); CK(hipDeviceSynchronize());
    CK(hipMemcpy(pb.data(), src ? src : o2, cmp * 4, hipMemcpyDeviceToHost));
    size_t diff = 0, nan = 0; double mx = 0;
    for (size_t i = 0; i < cmp; ++i) { if (memcmp(&pa[i], &pb[i], 4)) { if (!diff) fprintf(stderr, "first diff at %zu: %g vs %g\n", i, pa[i], pb[i]); ++diff; } if (pa[i] != pa[i]) ++nan; mx = std::max(mx, (double) fabsf(pa[i])); }
    printf("parity %d x %d x %d ksplit %d: %zu of %zu differ (ref max |v| %.3g, nan %zu) : %s\n", M, K, N, ksplit, diff, cmp, mx, nan, diff == 0 && nan == 0 && mx > 0 ? "IDENTICAL" : "MISMATCH");
    // timing: interleaved rounds, rotating weight sets
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double flop = 2.0 * M * K * N;
    for (int round = 0; round < 3; ++round)
        for (int which = 0; which < 2; ++which) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps * sets; ++i) which ? run_new(o2, i % sets) : run_ref(o1, i % sets);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / (reps * sets);
            printf("round %d %-28s %8.2f us   %7.1f TFLOP/s\n", round, which ? (variant == 2 ? "mfma_r64s_q4k_kernel<2>" : variant & 1 ? "mfma_r64s_q4k_kernel<1>" : "mfma_r64s_q4k_kernel<0>") : "mfma_r64_q4k_kernel<8>", us, flop / us * 1e-6);
        }
    {   // in-kernel stamps of the last r64s launch: cycles and 100 MHz ticks over the K loop of wave 0 of every workgroup
        static unsigned long long h[6][1024];
        CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(r64s_stamp), sizeof(h)));
        const int nwg = std::min<int>(1024, grid.x * grid.y * grid.z);
        std::vector<double> cyc, clk;
        unsigned long long first = ~0ull, last = 0;
        for (int i = 0; i < nwg; ++i) if (h[3][i]) { cyc.push_back((double) h[0][i] / (4.0 * h[3][i])); clk.push_back((double) h[0][i] / h[1][i] * 0.1); first = std::min(first, h[2][i]); last = std::max(last, h[2][i]); }
        std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
        if (!cyc.empty()) printf("stamps (%zu workgroups): cycles per K-step median %.0f (min %.0f max %.0f) = %.1f per MFMA; shader clock median %.2f GHz (min %.2f max %.2f); loop starts spread over %.2f us\n",
                                 cyc.size(), cyc[cyc.size() / 2], cyc.front(), cyc.back(), cyc[cyc.size() / 2] / 64.0, clk[clk.size() / 2], clk.front(), clk.back(), (last - first) * 0.01);
        // phases of thread 0 of every workgroup on the 100 MHz clock: entry -> K loop (prologue), the loop, loop end -> last store issued
        std::vector<double> pro, loop, epi;
        unsigned long long e0 = ~0ull, e1 = 0;
        for (int i = 0; i < nwg; ++i) if (h[3][i]) {
            pro.push_back((h[2][i] - h[4][i]) * 0.01); loop.push_back(h[1][i] * 0.01); epi.push_back((h[5][i] - h[2][i] - h[1][i]) * 0.01);
            e0 = std::min(e0, h[4][i]); e1 = std::max(e1, h[5][i]);
        }
        std::sort(pro.begin(), pro.end()); std::sort(loop.begin(), loop.end()); std::sort(epi.begin(), epi.end());
        if (!pro.empty()) printf("phases (us, median [min .. max]): prologue %.2f [%.2f .. %.2f] | K loop %.2f [%.2f .. %.2f] | epilogue %.2f [%.2f .. %.2f] | first entry -> last end %.2f\n",
                                 pro[pro.size() / 2], pro.front(), pro.back(), loop[loop.size() / 2], loop.front(), loop.back(), epi[epi.size() / 2], epi.front(), epi.back(), (e1 - e0) * 0.01);
    }
    return diff == 0 ? 0 : 1;
}
