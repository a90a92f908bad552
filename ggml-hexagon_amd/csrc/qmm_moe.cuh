// qmm_moe.cuh — GGML_OP_MUL_MAT_ID (indirect expert mat-mul) on gfx950.
//
// Semantics: ggml_compute_forward_mul_mat_id, ggml/src/ggml-cpu/ggml-cpu.c:6941-7197 (the Hexagon
// reference has no MUL_MAT_ID, ggml-hexagon.cpp:514): for token t and slot s
//     dst[:, s, t] = as[:, :, ids[s, t]] . b[:, s % ne11, t]
// `ids` stays on the device: nothing is read back to the host and nothing synchronizes.
//
//   few (token, slot) pairs  -> matvec_id_kernel: one pair per blockIdx.x, the expert is looked up
//                               in-kernel, activations quantized into LDS, rows streamed like qmm_matvec.
//   many pairs (prefill)     -> moe_sort_kernel groups the pairs by expert (counting sort in LDS),
//                               prep_act_kernel gathers + converts the pairs' src1 rows in expert order,
//                               mfma_kernel runs one grid with blockIdx.z = expert over the sorted rows
//                               and scatters the result rows through dst_off.
#pragma once

#include "qmm_matvec.cuh"
#include "qmm_mfma.cuh"

namespace qmm {

template <int T>
__global__ void __launch_bounds__(1024)
matvec_id_kernel(const uint8_t * __restrict__ as, const int64_t row_bytes, const int64_t expert_bytes, const int K, const int M,
                 const int n_expert, const float * __restrict__ b, const int ne11, const int64_t b_s1, const int64_t b_s2,
                 const int32_t * __restrict__ ids, const int n_used, const int64_t ids_s1,
                 float * __restrict__ dst, const int64_t d_s1, const int64_t d_s2, const int act_mode, int * __restrict__ flag,
                 const uint8_t * __restrict__ as2, float * __restrict__ dst2) {
    // blockIdx.z = 1: the second expert tensor of a pair that shares b and ids (ffn_gate_exps / ffn_up_exps)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int ACT = Traits<T>::ACT;
    int8_t *  aq = reinterpret_cast<int8_t *>(smem);
    float *   ad = reinterpret_cast<float *>(smem + K);
    int16_t * ab = reinterpret_cast<int16_t *>(ad + K / act_block<T>());

    const int p = blockIdx.x, t = p / n_used, s = p % n_used;
    const int e = ids[(int64_t) t * ids_s1 + s];
    if (e < 0 || e >= n_expert) {
        if (threadIdx.x == 0 && blockIdx.y == 0) atomicOr(flag, 1);
        return;
    }
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE, nwaves = blockDim.x / WAVE;
    const int units = K / MvUnit<T>::W;
    const uint8_t * We = (blockIdx.z ? as2 : as) + (int64_t) e * expert_bytes;
    float * out = (blockIdx.z ? dst2 : dst) + (int64_t) t * d_s2 + (int64_t) s * d_s1;

    int row = blockIdx.y * nwaves + wave;
    MvUnit<T> pre;
    if (row < M && lane < units) pre.load(We + (int64_t) row * row_bytes, lane);

    quantize_rows<ACT, MvUnit<T>::BSG, T>(b + (int64_t) t * b_s2 + (int64_t) (s % ne11) * b_s1, 0, 1, K, act_mode, aq, ad,
                       ACT == T_Q8_K ? ab : nullptr, tid, blockDim.x);
    __syncthreads();

    for (; row < M; row += gridDim.y * nwaves) {
        const uint8_t * wrow = We + (int64_t) row * row_bytes;
        float acc = 0.0f;
        int u = lane;
        if (u < units) acc += pre.dot(u, aq, ad, ab);
#pragma unroll 2
        for (u += WAVE; u < units; u += WAVE) {
            MvUnit<T> un;
            un.load(wrow, u);
            acc += un.dot(u, aq, ad, ab);
        }
        const int next = row + gridDim.y * nwaves;
        if (next < M && lane < units) pre.load(We + (int64_t) next * row_bytes, lane);
        acc = wave_sum(acc);
        if (lane == 0) out[row] = acc;
    }
}

// counting sort of the (token, slot) pairs by expert; one workgroup
__global__ void __launch_bounds__(1024)
moe_sort_kernel(const int32_t * __restrict__ ids, const int64_t ids_s1, const int n_used, const int n_tokens, const int n_expert,
                const int ne11, const int64_t b_s1, const int64_t b_s2, const int64_t d_s1, const int64_t d_s2,
                int * __restrict__ seg_start, int * __restrict__ seg_count, int * __restrict__ n_live,
                int64_t * __restrict__ gather, int64_t * __restrict__ dst_off, int * __restrict__ flag) {
    extern __shared__ int sh[];
    int * cnt = sh;                 // [n_expert]
    int * cur = sh + n_expert;      // [n_expert]
    const int P = n_used * n_tokens;
    for (int e = threadIdx.x; e < n_expert; e += blockDim.x) cnt[e] = 0;
    __syncthreads();
    for (int p = threadIdx.x; p < P; p += blockDim.x) {
        const int e = ids[(int64_t) (p / n_used) * ids_s1 + (p % n_used)];
        if (e >= 0 && e < n_expert) atomicAdd(&cnt[e], 1);
        else atomicOr(flag, 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int e = 0; e < n_expert; ++e) {
            seg_start[e] = run;
            seg_count[e] = cnt[e];
            cur[e] = run;
            run += cnt[e];
        }
        *n_live = run;
    }
    __syncthreads();
    for (int p = threadIdx.x; p < P; p += blockDim.x) {
        const int t = p / n_used, s = p % n_used;
        const int e = ids[(int64_t) t * ids_s1 + s];
        if (e < 0 || e >= n_expert) continue;
        const int pos = atomicAdd(&cur[e], 1);
        gather[pos]  = (int64_t) t * b_s2 + (int64_t) (s % ne11) * b_s1;
        dst_off[pos] = (int64_t) t * d_s2 + (int64_t) s * d_s1;
    }
}

template <int T>
inline int launch_matvec_id(qmm_ctx * c, hipStream_t st, const void * as, int64_t rb, int64_t eb, int K, int M, int n_expert,
                            const float * b, int ne11, int64_t b_s1, int64_t b_s2, const int32_t * ids, int n_used, int n_tokens,
                            int64_t ids_s1, float * dst, int64_t d_s1, int64_t d_s2, const void * as2 = nullptr, float * dst2 = nullptr) {
    const size_t lds = matvec_lds_bytes<T, 1>(K);
    auto kern = matvec_id_kernel<T>;
    if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    const int P = n_used * n_tokens, nm = as2 ? 2 : 1;
    const int nw = 8;
    int by = (M + nw - 1) / nw;
    const int cap = (2 * c->cus + P * nm - 1) / (P * nm);
    if (by > cap) by = cap;
    if (by < 1) by = 1;
    hipLaunchKernelGGL(kern, dim3(P, by, nm), dim3(nw * WAVE), lds, st, (const uint8_t *) as, rb, eb, K, M, n_expert, b, ne11, b_s1, b_s2,
                       ids, n_used, ids_s1, dst, d_s1, d_s2, c->act_mode, c->flag, (const uint8_t *) as2, dst2);
    HIP_TRY(hipGetLastError());
    return QMM_OK;
}

inline int moe_mul_mat_id(qmm_ctx * c, hipStream_t st, int type, const void * as, int64_t rb, int64_t eb, int64_t K, int64_t M,
                          int64_t n_expert, const float * b, int64_t ne11, int64_t b_nb1, int64_t b_nb2,
                          const int32_t * ids, int64_t n_used, int64_t n_tokens, int64_t ids_nb1,
                          float * dst, int64_t d_nb1, int64_t d_nb2, const void * as2 = nullptr, float * dst2 = nullptr) {
    // as2 / dst2: a second expert tensor of the same type and shape on the same b and ids (ffn_gate_exps + ffn_up_exps): one
    // mat-vec launch for both, or one sort + one activation prep for both
    const int64_t b_s1 = b_nb1 / 4, b_s2 = b_nb2 / 4, d_s1 = d_nb1 / 4, d_s2 = d_nb2 / 4, ids_s1 = ids_nb1 / 4;
    const int64_t P = n_used * n_tokens;
    if (P <= 16) {
#define QMM_MVID(TT)                                                                                                                  \
    return launch_matvec_id<TT>(c, st, as, rb, eb, (int) K, (int) M, (int) n_expert, b, (int) ne11, b_s1, b_s2, ids, (int) n_used, \
                                (int) n_tokens, ids_s1, dst, d_s1, d_s2, as2, dst2)
        switch (type) {
            case T_Q4_0: QMM_MVID(T_Q4_0);
            case T_Q8_0: QMM_MVID(T_Q8_0);
            case T_Q4_K: QMM_MVID(T_Q4_K);
            case T_Q5_K: QMM_MVID(T_Q5_K);
            default:     QMM_MVID(T_Q6_K);
        }
#undef QMM_MVID
    }
    if (n_expert > 4096) return fail(QMM_EUNSUPPORTED, "qmm_mul_mat_id: n_expert=%lld", (long long) n_expert);
    const int Kp = mfma_kpad(K);
    const int64_t rows = P + 128;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) & ~(size_t) 255; return o; };
    const size_t o_xh = take((size_t) rows * Kp * 2), o_sc = take((size_t) rows * 4), o_g = take((size_t) P * 8), o_d = take((size_t) P * 8),
                 o_ss = take((size_t) n_expert * 4), o_sn = take((size_t) n_expert * 4), o_nl = take(4);
    int rc = ensure_ws(c, off);
    if (rc) return rc;
    uint8_t * ws = (uint8_t *) c->ws;
    uint16_t * xh = (uint16_t *) (ws + o_xh);
    float * scale = (float *) (ws + o_sc);
    int64_t * gather = (int64_t *) (ws + o_g);
    int64_t * dst_off = (int64_t *) (ws + o_d);
    int * seg_start = (int *) (ws + o_ss), * seg_count = (int *) (ws + o_sn), * n_live = (int *) (ws + o_nl);

    hipLaunchKernelGGL(moe_sort_kernel, dim3(1), dim3(1024), (size_t) n_expert * 8, st, ids, ids_s1, (int) n_used, (int) n_tokens,
                       (int) n_expert, (int) ne11, b_s1, b_s2, d_s1, d_s2, seg_start, seg_count, n_live, gather, dst_off, c->flag);
    HIP_TRY(hipGetLastError());
    const bool q8_0 = (type == T_Q4_0 || type == T_Q8_0);
    const int frag = mfma_use_skinny(c, type, P, M, n_expert);
    rc = q8_0 ? launch_prep<T_Q8_0>(c, st, type, b, 0, gather, n_live, (int) P, (int) P, (int) K, Kp, frag, xh, scale)
              : launch_prep<T_Q8_K>(c, st, type, b, 0, gather, n_live, (int) P, (int) P, (int) K, Kp, frag, xh, scale);
    if (rc) return rc;
    MfmaOperand op = { xh, scale, Kp, frag };
    rc = launch_mfma_any(c, st, type, as, rb, eb, (int) n_expert, (int) M, (int) K, op, seg_start, seg_count, (int) P,
                         (int) ((P + 127) / 128), dst, 0, dst_off);
    if (rc || !as2) return rc;
    const int64_t shift = dst2 - dst;                         // same scatter pattern, other destination tensor
    return launch_mfma_any(c, st, type, as2, rb, eb, (int) n_expert, (int) M, (int) K, op, seg_start, seg_count, (int) P,
                           (int) ((P + 127) / 128), dst + shift, 0, dst_off);
}

} // namespace qmm
