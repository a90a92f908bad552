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
#include "qmm_mvunit.cuh"

namespace qmm {

constexpr int MV_MAX_GROUP = 4;

struct MatvecGroup {
    const uint8_t * w[MV_MAX_GROUP];
    float *         dst[MV_MAX_GROUP];
    int64_t         row_bytes[MV_MAX_GROUP];
    int64_t         ldd[MV_MAX_GROUP];
    int             row_end[MV_MAX_GROUP];   // cumulative row counts
    int             n;
    int             type[MV_MAX_GROUP];      // per matrix; read by matvec_kmix_kernel only
    // what a caller that sees the whole layer can fold in (qmm_mul_mat_group_ex), all optional:
    const float *   res[MV_MAX_GROUP];       // dst = W x + res (same row stride as dst): the residual add behind wo / ffn_down
    const float *   norm_w;                  // x is rms_norm(x) * norm_w, formed while staging (attn_norm / ffn_norm in front of q/k/v, gate/up)
    float           norm_eps;
    int             swiglu;                  // 1 / 2: two matrices of one type and shape; dst[0] = silu(W_a x) * (W_b x) with a = swiglu - 1
};

// RMS_NORM * w of the NTOK activation rows into LDS (f32), by every workgroup: K floats per row are one or a few float4 per
// thread, so this is two L2 reads and a block reduction in front of the quantizer instead of a launch of its own.
// Same arithmetic as rms_norm_vec_kernel (qmm_ops.hip): per-thread sum of squares, wave butterfly, waves in order.
template <int NTOK>
__device__ __forceinline__ void stage_rms_norm(const float * __restrict__ x, int64_t ldx, int K, const float * __restrict__ w, float eps,
                                               float * xs, float * red, int tid, int nthreads) {
    const int lane = tid & (WAVE - 1), wave = tid / WAVE, nwaves = nthreads / WAVE;
    const bool one_trip = K <= nthreads * 4;             // K = 4096 with 16 waves: the thread's slice stays in registers and
    const int i0 = tid * 4;                              // the weight is requested before the reduction, not behind it
    float4 ww = make_float4(0.f, 0.f, 0.f, 0.f);
    if (one_trip && i0 < K) ww = *reinterpret_cast<const float4 *>(w + i0);
#pragma unroll
    for (int n = 0; n < NTOK; ++n) {
        float sum = 0.0f;
        float4 keep = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = i0; i < K; i += nthreads * 4) {
            const float4 v = *reinterpret_cast<const float4 *>(x + (int64_t) n * ldx + i);
            if (one_trip) keep = v;
            else *reinterpret_cast<float4 *>(xs + (size_t) n * K + i) = v;
            sum += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        sum = wave_sum(sum);
        __syncthreads();
        if (lane == 0) red[wave] = sum;
        __syncthreads();
        float tot = red[0];
        for (int k = 1; k < nwaves; ++k) tot += red[k];
        const float scale = 1.0f / sqrtf(tot / (float) K + eps);
        for (int i = i0; i < K; i += nthreads * 4) {
            float4 v = one_trip ? keep : *reinterpret_cast<float4 *>(xs + (size_t) n * K + i);
            if (!one_trip) ww = *reinterpret_cast<const float4 *>(w + i);
            v.x = v.x * scale * ww.x; v.y = v.y * scale * ww.y; v.z = v.z * scale * ww.z; v.w = v.w * scale * ww.w;
            *reinterpret_cast<float4 *>(xs + (size_t) n * K + i) = v;
        }
    }
    __syncthreads();
}

template <int T> __host__ __device__ constexpr int act_block() { return Traits<T>::ACT == T_Q8_0 ? 32 : 256; }

template <int T, int NTOK> __host__ __device__ inline size_t matvec_lds_bytes(int K) {
    size_t b = (size_t) NTOK * K + (size_t) NTOK * (K / act_block<T>()) * 4;
    if (Traits<T>::ACT == T_Q8_K) b += (size_t) NTOK * (K / MvUnit<T>::BSG) * 2;
    return (b + 15) & ~(size_t) 15;
}

// Work split: rows are dealt to waves round-robin (row = wave_id + i * n_waves), so at any moment the
// chip sweeps one contiguous window of W.  A wave walks a row in slices of 64 units; latency is hidden by
// the 16 waves per CU (measured: explicit per-wave software pipelining is slower than plain TLP here).
// Loads are issued unconditionally (unit index clamped) because hipcc drains vmcnt to 0 at every branch
// that surrounds a global_load.  (Measured and dropped: requesting the first slice of a wave's first row
// before the activation staging.  In-kernel timestamps show the request then completes under the staging,
// but the kernel's span stays 3.8 us for a 9.4 MB matrix: HBM latency plus 9.4 MB at ~4.5 TB/s.  Also dropped: two
// adjacent rows per wave step sharing the activation reads: 66 MB at N = 1 15.3 -> 15.9 us, N = 8 unchanged; four
// slices' loads in flight per wave on K = 14336 rows: 11.5 -> 14.7 us; two Q4_0 blocks per lane and slice: 9.4 MB
// 6.2 -> 5.8 us but the 7B pass 743 -> 724 tok/s; nontemporal weight loads: 66 MB 15.3 -> 18.5 us.  Every variant that
// widens a wave's window of outstanding loads loses on the long streams.  Once more at the end of the round: both slices of a
// K = 4096 row requested together (iters == 2 only, long rows untouched): tg128 747 -> 738.)
// EX: the instantiation that honours g.res / g.norm_w (qmm_mul_mat_group_ex); the plain one compiles them away, so the hot
// path of bench.py is the kernel it was before those fields existed (with them as run-time branches: 9.17 -> 9.50 us per launch)
template <int T, int NTOK, bool EX>
__global__ void __launch_bounds__(1024)
matvec_kernel(const MatvecGroup g, const float * __restrict__ x, const int64_t ldx, const int K, const int act_mode) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int ACT = Traits<T>::ACT;
    int8_t *  aq = reinterpret_cast<int8_t *>(smem);
    float *   ad = reinterpret_cast<float *>(smem + (size_t) NTOK * K);
    int16_t * ab = reinterpret_cast<int16_t *>(ad + (size_t) NTOK * (K / act_block<T>()));

    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / WAVE), nwaves = blockDim.x / WAVE;
    const int units = K / MvUnit<T>::W;
    const int iters = (units + WAVE - 1) / WAVE;
    const int total_rows = g.row_end[g.n - 1];
    const int W = gridDim.x * nwaves, gw = blockIdx.x * nwaves + wave;

    auto locate = [&](int r, const uint8_t *& wrow, float *& drow, const float *& rrow, int64_t & ldd) {
        int i = 0, b0 = 0;
#pragma unroll
        for (int k = 0; k < MV_MAX_GROUP - 1; ++k)
            if (k < g.n - 1 && r >= g.row_end[k]) { i = k + 1; b0 = g.row_end[k]; }
        wrow = g.w[i] + (int64_t) (r - b0) * g.row_bytes[i];
        drow = g.dst[i] + (r - b0);
        rrow = EX && g.res[i] ? g.res[i] + (r - b0) : nullptr;
        ldd  = g.ldd[i];
    };

    const uint8_t * wrow; float * drow; const float * rrow; int64_t ldd;
    locate(min(gw, total_rows - 1), wrow, drow, rrow, ldd);

    const float * xq = x;
    int64_t ldq = ldx;
    if (EX && g.norm_w) {                                 // the f32 rows sit behind the quantized fields (launcher sizes LDS for it)
        __shared__ float red[16];
        float * xs = reinterpret_cast<float *>(smem + matvec_lds_bytes<T, NTOK>(K));
        stage_rms_norm<NTOK>(x, ldx, K, g.norm_w, g.norm_eps, xs, red, tid, blockDim.x);
        xq = xs;
        ldq = K;
    }
    quantize_rows<ACT, MvUnit<T>::BSG, T>(xq, ldq, NTOK, K, act_mode, aq, ad, ACT == T_Q8_K ? ab : nullptr, tid, blockDim.x);
    __syncthreads();

    if (EX && g.swiglu) {
        // ffn_gate and ffn_up as row pairs: a wave computes row r of both and writes silu(gate) * up, the SwiGLU of build_ffn
        // (ggml_silu + ggml_mul, same float operations as unary_kernel): neither product makes a trip through HBM
        const int ga = g.swiglu - 1, ub = 1 - ga, M = g.row_end[0];
        for (int row = gw; row < M; row += W) {
            const uint8_t * wg = g.w[ga] + (int64_t) row * g.row_bytes[ga], * wu = g.w[ub] + (int64_t) row * g.row_bytes[ub];
            // one row at a time, as in the plain loop (two rows' loads in flight per wave lose on long streams: see above)
            float tg[NTOK], out = 0.0f;
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const uint8_t * wr = pass == 0 ? wg : wu;
                float acc[NTOK];
#pragma unroll
                for (int n = 0; n < NTOK; ++n) acc[n] = 0.0f;
                for (int it = 0; it < iters; ++it) {
                    const int u = lane + WAVE * it, uc = min(u, units - 1);
                    MvUnit<T> un;
                    un.load(wr, uc);
#pragma unroll
                    for (int n = 0; n < NTOK; ++n) {
                        const float p = un.dot(uc, aq + (size_t) n * K, ad + (size_t) n * (K / act_block<T>()), ab + (size_t) n * (K / MvUnit<T>::BSG));
                        acc[n] += u < units ? p : 0.0f;
                    }
                }
#pragma unroll
                for (int n = 0; n < NTOK; ++n) {
                    const float t = wave_sum(acc[n]);
                    if (pass == 0) tg[n] = t;
                    else if (lane == n) out = tg[n] / (1.0f + expf(-tg[n])) * t;
                }
            }
            if (lane < NTOK) g.dst[0][row + (int64_t) lane * g.ldd[0]] = out;
        }
        return;
    }

    for (int row = gw; row < total_rows; row += W) {
        float acc[NTOK];
#pragma unroll
        for (int n = 0; n < NTOK; ++n) acc[n] = 0.0f;
        int it = 0;
        for (; it < iters; ++it) {
            const int u = lane + WAVE * it, uc = min(u, units - 1);
            MvUnit<T> un;
            un.load(wrow, uc);
#pragma unroll
            for (int n = 0; n < NTOK; ++n) {
                const float p = un.dot(uc, aq + (size_t) n * K, ad + (size_t) n * (K / act_block<T>()), ab + (size_t) n * (K / MvUnit<T>::BSG));
                acc[n] += u < units ? p : 0.0f;
            }
        }
        float out = 0.0f;
#pragma unroll
        for (int n = 0; n < NTOK; ++n) {
            const float t = wave_sum(acc[n]);
            if (lane == n) out = t;
        }
        if (lane < NTOK) drow[(int64_t) lane * ldd] = EX && rrow ? out + rrow[(int64_t) lane * ldd] : out;
        if (row + W < total_rows) locate(row + W, wrow, drow, rrow, ldd);
    }
}

// ---------------------------------------------------------------------------------------------
// K-quant matrices of DIFFERENT types that share src1 (Q4_K_M's attn_q/attn_k in Q4_K beside attn_v in Q6_K, Q5_K_M's and the
// 70B recipe's Q5_K/Q6_K mixes) in ONE launch: they all dot against the same Q8_K activations, so the staging is done once
// (block sums per 16, which Q6_K needs and Q4_K/Q5_K add up in pairs; Q4_K's LDS order) and each wave picks the dot of the
// matrix its row belongs to.  Saves the separate 3.4 MB launch (4.5 us at batch 1) of every such layer.
__host__ __device__ inline size_t kmix_lds_bytes(int ntok, int K) {
    return ((size_t) ntok * K + (size_t) ntok * (K / 256) * 4 + (size_t) ntok * (K / 16) * 2 + 15) & ~(size_t) 15;
}

template <int NTOK, bool EX>
__global__ void __launch_bounds__(1024)
matvec_kmix_kernel(const MatvecGroup g, const float * __restrict__ x, const int64_t ldx, const int K, const int act_mode) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    int8_t *  aq = reinterpret_cast<int8_t *>(smem);
    float *   ad = reinterpret_cast<float *>(smem + (size_t) NTOK * K);
    int16_t * ab = reinterpret_cast<int16_t *>(ad + (size_t) NTOK * (K / 256));

    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / WAVE), nwaves = blockDim.x / WAVE;
    const int units = K / 64;
    const int iters = (units + WAVE - 1) / WAVE;
    const int total_rows = g.row_end[g.n - 1];
    const int W = gridDim.x * nwaves, gw = blockIdx.x * nwaves + wave;

    const float * xq = x;
    int64_t ldq = ldx;
    if (EX && g.norm_w) {
        __shared__ float red[16];
        float * xs = reinterpret_cast<float *>(smem + kmix_lds_bytes(NTOK, K));
        stage_rms_norm<NTOK>(x, ldx, K, g.norm_w, g.norm_eps, xs, red, tid, blockDim.x);
        xq = xs;
        ldq = K;
    }
    quantize_rows<T_Q8_K, 16, T_Q4_K>(xq, ldq, NTOK, K, act_mode, aq, ad, ab, tid, blockDim.x);
    __syncthreads();

    for (int row = gw; row < total_rows; row += W) {
        int i = 0, b0 = 0;
#pragma unroll
        for (int k = 0; k < MV_MAX_GROUP - 1; ++k)
            if (k < g.n - 1 && row >= g.row_end[k]) { i = k + 1; b0 = g.row_end[k]; }
        const uint8_t * wrow = g.w[i] + (int64_t) (row - b0) * g.row_bytes[i];
        float * drow = g.dst[i] + (row - b0);
        const float * rrow = EX && g.res[i] ? g.res[i] + (row - b0) : nullptr;
        const int64_t ldd = g.ldd[i];
        const int type = __builtin_amdgcn_readfirstlane(g.type[i]);
        float acc[NTOK];
#pragma unroll
        for (int n = 0; n < NTOK; ++n) acc[n] = 0.0f;
        auto walk = [&](auto unit_tag, auto dot_fn) {
            using U = decltype(unit_tag);
            for (int it = 0; it < iters; ++it) {
                const int u = lane + WAVE * it, uc = min(u, units - 1);
                U un;
                un.load(wrow, uc);
#pragma unroll
                for (int n = 0; n < NTOK; ++n) {
                    const float p = dot_fn(un, uc, aq + (size_t) n * K, ad + (size_t) n * (K / 256), ab + (size_t) n * (K / 16));
                    acc[n] += u < units ? p : 0.0f;
                }
            }
        };
        if (type == T_Q4_K)
            walk(MvUnit<T_Q4_K>{}, [](const MvUnit<T_Q4_K> & un, int uc, const int8_t * a, const float * d, const int16_t * b) { return un.template dot<true>(uc, a, d, b); });
        else if (type == T_Q5_K)
            walk(MvUnit<T_Q5_K>{}, [](const MvUnit<T_Q5_K> & un, int uc, const int8_t * a, const float * d, const int16_t * b) { return un.template dot<true>(uc, a, d, b); });
        else
            walk(MvUnit<T_Q6_K>{}, [](const MvUnit<T_Q6_K> & un, int uc, const int8_t * a, const float * d, const int16_t * b) { return un.template dot<T_Q4_K>(uc, a, d, b); });
        float out = 0.0f;
#pragma unroll
        for (int n = 0; n < NTOK; ++n) {
            const float t = wave_sum(acc[n]);
            if (lane == n) out = t;
        }
        if (lane < NTOK) drow[(int64_t) lane * ldd] = EX && rrow ? out + rrow[(int64_t) lane * ldd] : out;
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
