// qmm_matvec.cuh — batch <= 8 quantized mat-vec for gfx950, the token-generation kernel.
//
// HBM-bound design (roofline: weight bytes / 8 TB/s):
//   * one launch per MUL_MAT (or per group of MUL_MATs that share src1 and a weight type): the f32
//     activations are quantized to Q8_0 / Q8_K *inside* the kernel, by every workgroup, straight
//     into LDS (q int8, block scales f32, Q8_K bsums int16) — no separate quantize launch, no
//     round trip through HBM.  This is phase 1 of ggml_compute_forward_mul_mat
//     (ggml/src/ggml-cpu/ggml-cpu.c:6807-6842; ggml-hexagon kernels/ggml-dsp.c:1262-1285).
//   * each wave owns whole weight rows; each lane owns `units` of the row (qmm_device.cuh) which it
//     streams from HBM with 16-byte loads, dots against the LDS activations with v_dot4_i32_i8, and
//     scales per block in f32 exactly as the CPU vec_dot does (ggml-cpu-quants.c a9-a13 in
//     SURVEY.md §8a).  The row sum is finished with wave shuffles.
//   * no LDS round trip for the weights: every byte of W is read once, by one lane.
#pragma once

#include "qmm_act.cuh"

namespace qmm {

constexpr int MV_MAX_GROUP = 4;

struct MatvecGroup {
    const uint8_t * w[MV_MAX_GROUP];
    float *         dst[MV_MAX_GROUP];
    int64_t         row_bytes[MV_MAX_GROUP];
    int64_t         ldd[MV_MAX_GROUP];
    int             row_end[MV_MAX_GROUP];   // cumulative row counts
    int             n;
};

template <int T> __host__ __device__ constexpr int act_block() { return Traits<T>::ACT == T_Q8_0 ? 32 : 256; }

template <int T, int NTOK> __host__ __device__ inline size_t matvec_lds_bytes(int K) {
    size_t b = (size_t) NTOK * K + (size_t) NTOK * (K / act_block<T>()) * 4;
    if (Traits<T>::ACT == T_Q8_K) b += (size_t) NTOK * (K / 16) * 2;
    return (b + 15) & ~(size_t) 15;
}

template <int T, int NTOK>
__global__ void __launch_bounds__(1024)
matvec_kernel(const MatvecGroup g, const float * __restrict__ x, const int64_t ldx, const int K, const int act_mode) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int ACT = Traits<T>::ACT;
    int8_t *  aq = reinterpret_cast<int8_t *>(smem);
    float *   ad = reinterpret_cast<float *>(smem + (size_t) NTOK * K);
    int16_t * ab = reinterpret_cast<int16_t *>(ad + (size_t) NTOK * (K / act_block<T>()));

    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE, nwaves = blockDim.x / WAVE;
    const int units = K / Traits<T>::UNIT_W;
    const int total_rows = g.row_end[g.n - 1];

    // first row of this wave: put its weight loads in flight before the activation staging
    int row = blockIdx.x * nwaves + wave;
    Unit<T> pre;
    const uint8_t * wrow = nullptr;
    float * drow = nullptr;
    int64_t ldd = 0;
    auto locate = [&](int r) {
        int i = 0, base = 0;
#pragma unroll
        for (int k = 0; k < MV_MAX_GROUP - 1; ++k)
            if (k < g.n - 1 && r >= g.row_end[k]) { i = k + 1; base = g.row_end[k]; }
        wrow = g.w[i] + (int64_t) (r - base) * g.row_bytes[i];
        drow = g.dst[i] + (r - base);
        ldd  = g.ldd[i];
    };
    const bool have = row < total_rows;
    if (have) {
        locate(row);
        if (lane < units) pre.load(wrow, lane);
    }

    quantize_rows<ACT>(x, ldx, NTOK, K, act_mode, aq, ad, ACT == T_Q8_K ? ab : nullptr, tid, blockDim.x);
    __syncthreads();

    for (; row < total_rows; row += gridDim.x * nwaves) {
        float acc[NTOK];
#pragma unroll
        for (int n = 0; n < NTOK; ++n) acc[n] = 0.0f;

        int u = lane;
        if (u < units) {                      // first chunk: already in registers
#pragma unroll
            for (int n = 0; n < NTOK; ++n)
                acc[n] += pre.dot(u, aq + (size_t) n * K, ad + (size_t) n * (K / act_block<T>()), ab + (size_t) n * (K / 16));
        }
#pragma unroll 2
        for (u += WAVE; u < units; u += WAVE) {
            Unit<T> un;
            un.load(wrow, u);
#pragma unroll
            for (int n = 0; n < NTOK; ++n)
                acc[n] += un.dot(u, aq + (size_t) n * K, ad + (size_t) n * (K / act_block<T>()), ab + (size_t) n * (K / 16));
        }

        // next row's first chunk goes in flight before the reduction of this one
        float * dcur = drow;
        const int64_t ldcur = ldd;
        const int next = row + gridDim.x * nwaves;
        if (next < total_rows) {
            locate(next);
            if (lane < units) pre.load(wrow, lane);
        }

        float out = 0.0f;
#pragma unroll
        for (int n = 0; n < NTOK; ++n) {
            const float s = wave_sum(acc[n]);
            if (lane == n) out = s;
        }
        if (lane < NTOK) dcur[(int64_t) lane * ldcur] = out;
    }
}

// ---------------------------------------------------------------------------------------------
// bit-exact block unpack to f32, one unit per thread (parity surface of Unit<T>::load/to_f32)
template <int T>
__global__ void __launch_bounds__(256)
dequant_kernel(const uint8_t * __restrict__ w, const int64_t row_bytes, const int64_t rows, const int K, float * __restrict__ dst) {
    const int units = K / Traits<T>::UNIT_W;
    const int64_t gid = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= rows * units) return;
    const int64_t r = gid / units;
    const int u = (int) (gid % units);
    Unit<T> un;
    un.load(w + r * row_bytes, u);
    float out[Traits<T>::UNIT_W];
    un.to_f32(u, out);
#pragma unroll
    for (int rr = 0; rr < Unit<T>::RUNS; ++rr) {
        float * o = dst + r * K + Unit<T>::k_run(u, rr);
#pragma unroll
        for (int e = 0; e < Unit<T>::RUN_LEN; e += 4)
            *reinterpret_cast<float4 *>(o + e) = make_float4(out[rr * Unit<T>::RUN_LEN + e], out[rr * Unit<T>::RUN_LEN + e + 1],
                                                              out[rr * Unit<T>::RUN_LEN + e + 2], out[rr * Unit<T>::RUN_LEN + e + 3]);
    }
}

} // namespace qmm
