// qmm_mfma.cuh — batched (prefill) quantized mat-mul on the gfx950 matrix cores.
//
//   dst[n, m] = sum_k W[m, k] * x[n, k]        N > 8 tokens
//
// This file: the activation side and the host logic shared by all prefill kernels, plus the LDS-tile kernel.
//   prep_act_kernel  f32 activations -> 16-bit MFMA operand  Xh[Npad][Kp]  (+ one f32 scale per token), once per src1
//        QMM_PREC_F16_Q8 (default): the row is quantized to Q8_0 / Q8_K exactly as the CPU backend does
//                          (ggml-cpu.c:6807-6842), then each value q*d is stored as f16 relative to the
//                          row's largest block scale (|value| <= 127, no f16 range problem); the GEMM
//                          therefore multiplies the SAME quantized activations as ggml's vec_dot, and the
//                          only difference left is f16 rounding of the operands (~2^-12 relative).
//                          The row is written in the k-order (PERM) and layout (row-major / fragment-major)
//                          the weight kernel's lane ownership wants.
//        QMM_PREC_BF16   : plain round-to-nearest bf16 of x (no Q8 emulation).
//   weight kernels, QMM_PREC_F16_Q8: qmm_mfma_regb.cuh (register-B tiled kernels, few-token kernel, split-K, groups).
//   mfma_kernel (here), QMM_PREC_BF16 only: each workgroup owns a BN-token x BM-row tile of dst.  Per K-step it
//        - fetches one weight *unit* per thread straight from HBM (16-byte loads, qmm_device.cuh),
//          unpacks it bit-exactly to f32 (Unit<T>::to_f32), rounds to bf16 and stores it into the
//          XOR-swizzled LDS tile Ws[BM][BK] (4 producer waves);
//        - copies the matching Xh tile into Xs[BN][BK];
//        - runs v_mfma_f32_32x32x16_bf16 (4 consumer waves) with tokens on the MFMA row index and weight
//          rows on the column (= lane) index, so that the epilogue writes 128 contiguous bytes of dst per
//          half-wave.
//
// MoE (MUL_MAT_ID) reuses the kernels: blockIdx.z selects the expert, seg_start/seg_count (device
// arrays) give the expert's slice of the expert-sorted token list, dst_off scatters the rows.
#pragma once

#include "qmm_act.cuh"
#include "qmm_host.h"

namespace qmm {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16   bf16x8 __attribute__((ext_vector_type(8)));
typedef float    f32x16 __attribute__((ext_vector_type(16)));

__device__ int g_mfma_dbg = 0;     // development ablation knob (GGML_MI355X_ABLATE), 0 in production

template <int T> struct MfmaBK { static constexpr int value = 64; };
template <> struct MfmaBK<T_Q6_K> { static constexpr int value = 128; };

// physical byte offset of 16-byte slot `s` of row `r` in a [rows][BK] 16-bit tile
template <int BK> __device__ __forceinline__ int tile_off(int r, int s) {
    if (BK == 64) return r * 128 + ((s ^ ((r >> 1) & 7)) << 4);
    else          return r * 256 + ((s ^ (r & 15)) << 4);
}

__device__ __forceinline__ uint32_t pack_f16(float a, float b) {
    const __half2 h = __floats2half2_rn(a, b);
    return *reinterpret_cast<const uint32_t *>(&h);
}
__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    bf2 v;
    v[0] = (__bf16) a;
    v[1] = (__bf16) b;
    return *reinterpret_cast<const uint32_t *>(&v);
}
template <bool F16> __device__ __forceinline__ uint32_t pack16(float a, float b) { return F16 ? pack_f16(a, b) : pack_bf16(a, b); }

// fragment-major operand layout: index (in 16-byte chunks) of the chunk holding positions k..k+7 of token row `row`.
// A K-step is BKP positions = BKP/16 MFMA k-steps; chunk (kk, h) of a row belongs to lane h*32 + row%32 of its 32-token tile.
__device__ __forceinline__ int64_t frag_major_chunk(int row, int k, int Kp, int BKP) {
    const int ks = k / BKP, p = k % BKP, kk = p >> 4, h = (p >> 3) & 1;
    return (((int64_t) (row >> 5) * (Kp / BKP) + ks) * (BKP / 16) + kk) * 64 + h * 32 + (row & 31);
}

// ------------------------------------------------------------------------------------------------
// stage 1: activation rows -> MFMA operand rows.  One workgroup per output row.
//   src row r: x + gather(r)   (gather == nullptr: r*ldx; MoE: element offset of the pair's src1 row)
//   out row r: xh + r*Kp   (16-bit), scale[r]
// PERM: the k-order in which the operand row is stored; MFMA sums over k, so any order shared by both operands is fine.
//   0  natural
//   1  every aligned group of four k as (0, 2, 1, 3): the order the packed-f16 unpacks (unpack_q6k_f16) produce
//   2, 3, 4, 5  the lane-ownership orders of the register-B kernels for Q4_K/Q5_K, Q6_K, Q4_0, Q8_0 (table in qmm_mfma_regb.cuh):
//      position p = kk*16 + h*8 + e of a 64- (Q6_K: 128-) block holds k = f(kk, h) + (0,2,1,3,4,6,5,7)[e]
template <int ACT, bool F16Q8, int PERM>
__global__ void __launch_bounds__(256)
prep_act_kernel(const float * __restrict__ x, const int64_t ldx, const int64_t * __restrict__ gather,
                const int * __restrict__ n_rows_dev, const int n_rows, const int K, const int Kp, const int act_mode,
                const int frag_major, uint16_t * __restrict__ xh, float * __restrict__ scale,
                const float * __restrict__ x2 = nullptr, const int64_t ldx2 = 0) {      // x2: the row is silu(x) * x2 (qmm_mul_mat_swiglu_in)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int r = blockIdx.x;
    const int tid = threadIdx.x;
    const int live_rows = n_rows_dev ? *n_rows_dev : n_rows;
    // where the 8 values of output position k (multiple of 8) of this row go: row-major Xh[r][k], or fragment-major
    // (few-token kernel): the 64 lanes' 16-byte A-fragment chunks of one (32-token tile, K-step, 16-deep k-step) contiguous
    constexpr int BKP = PERM == 3 ? 128 : 64;
    auto out_at = [&](int k) -> uint4 * {
        if (!frag_major) return reinterpret_cast<uint4 *>(xh + (int64_t) r * Kp + k);
        return reinterpret_cast<uint4 *>(xh) + frag_major_chunk(r, k, Kp, BKP);
    };
    if (r >= live_rows) {                                   // padding rows: zeros
        for (int k = tid * 8; k < Kp; k += 256 * 8) *out_at(k) = make_uint4(0, 0, 0, 0);
        if (tid == 0) scale[r] = 0.0f;
        return;
    }
    const float * src = x + (gather ? gather[r] : (int64_t) r * ldx);
    if (!F16Q8) {
        for (int k = tid * 8; k < Kp; k += 256 * 8) {
            uint4 o = make_uint4(0, 0, 0, 0);
            if (k < K) {
                const float4 a = *reinterpret_cast<const float4 *>(src + k);
                const float4 b = *reinterpret_cast<const float4 *>(src + k + 4);
                o = make_uint4(pack_bf16(a.x, a.y), pack_bf16(a.z, a.w), pack_bf16(b.x, b.y), pack_bf16(b.z, b.w));
            }
            *out_at(k) = o;
        }
        if (tid == 0) scale[r] = 1.0f;
        return;
    }
    constexpr int QB = ACT == T_Q8_0 ? 32 : 256;
    int8_t * aq = reinterpret_cast<int8_t *>(smem);
    float *  ad = reinterpret_cast<float *>(smem + K);
    float *  red = ad + K / QB;                              // 4 floats for the block max
    if (x2) quantize_rows<ACT, 16, 0, true>(src, 0, 1, K, act_mode, aq, ad, nullptr, tid, 256, x2 + (int64_t) r * ldx2, 0);
    else    quantize_rows<ACT>(src, 0, 1, K, act_mode, aq, ad, nullptr, tid, 256);
    __syncthreads();
    float mx = 0.0f;
    for (int b = tid; b < K / QB; b += 256) mx = fmaxf(mx, fabsf(ad[b]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, WAVE));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float inv = mx > 0.0f ? 1.0f / mx : 0.0f;
    for (int k = tid * 8; k < Kp; k += 256 * 8) {
        uint4 o = make_uint4(0, 0, 0, 0);
        int ksrc = k;                                        // where this group of 8 output positions comes from
        if (PERM == 2) {
            const int p = k & 63, kk = p >> 4, h = (p >> 3) & 1;
            ksrc = (k & ~63) + (kk & 1) * 8 + 16 * h + 32 * (kk >> 1);
        } else if (PERM == 4) {
            const int p = k & 63, kk = p >> 4, h = (p >> 3) & 1;
            ksrc = (k & ~63) + 32 * h + (kk & 1) * 8 + 16 * (kk >> 1);
        } else if (PERM == 5) {
            const int p = k & 63, kk = p >> 4, h = (p >> 3) & 1;
            ksrc = (k & ~63) + 32 * h + 8 * kk;
        } else if (PERM == 3) {
            const int p = k & 127, kk = p >> 4, h = (p >> 3) & 1;
            ksrc = (k & ~127) + 32 * (kk >> 1) + 16 * h + 8 * (kk & 1);
        }
        if (ksrc < K) {                                      // K need not be a multiple of the permutation block (Q4_0: K % 32)
            const float t = ad[ksrc / QB] * inv;             // |t| <= 1
            const int2 qq = *reinterpret_cast<const int2 *>(aq + ksrc);
            float v[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i]     = (float) (int8_t) ((qq.x >> (8 * i)) & 0xff) * t;
                v[4 + i] = (float) (int8_t) ((qq.y >> (8 * i)) & 0xff) * t;
            }
            o = PERM ? make_uint4(pack_f16(v[0], v[2]), pack_f16(v[1], v[3]), pack_f16(v[4], v[6]), pack_f16(v[5], v[7]))
                      : make_uint4(pack_f16(v[0], v[1]), pack_f16(v[2], v[3]), pack_f16(v[4], v[5]), pack_f16(v[6], v[7]));
        }
        *out_at(k) = o;
    }
    if (tid == 0) scale[r] = mx;
}

// Q4_K -> f16 without leaving the packed domain: a nibble n OR-ed into 0x6400 is the f16 1024 + n; subtracting 1024
// (exact) and one packed FMA with the f16-rounded sub-block scale and offset give w = ds*n - om.  Three roundings of
// <= 2^-11 instead of one, ~2.5 VALU per weight instead of ~5.5.  Output dword pairs hold (k, k+2), (k+1, k+3): PERM4.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t h2_bits(f16x2 v) { return *reinterpret_cast<uint32_t *>(&v); }
__device__ __forceinline__ f16x2 bits_h2(uint32_t u) { return *reinterpret_cast<f16x2 *>(&u); }

__device__ __forceinline__ void unpack_q4k_f16(const Unit<T_Q4_K> & wu, int j, uint4 (&lo)[2], uint4 (&hi)[2]) {
    int s0, m0, s1, m1;
    k4_scale_min(wu.hdr, 2 * j, s0, m0);
    k4_scale_min(wu.hdr, 2 * j + 1, s1, m1);
    const float d = h2f(wu.hdr.x & 0xffff), dmin = h2f(wu.hdr.x >> 16);
    const _Float16 ds0 = (_Float16) (d * (float) s0), ds1 = (_Float16) (d * (float) s1);
    const _Float16 no0 = (_Float16) (-(dmin * (float) m0)), no1 = (_Float16) (-(dmin * (float) m1));
    const f16x2 DS0 = { ds0, ds0 }, DS1 = { ds1, ds1 }, NO0 = { no0, no0 }, NO1 = { no1, no1 };
    const f16x2 BIAS = { (_Float16) -1024.0f, (_Float16) -1024.0f };
    const uint32_t w[4] = { wu.qs.x, wu.qs.y, wu.qs.z, wu.qs.w };
    uint32_t ol[8], oh[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t M = 0x000f000fu, E = 0x64006400u;
        const f16x2 a = bits_h2((w[i] & M) | E) + BIAS;               // bytes 0,2 low nibbles
        const f16x2 b = bits_h2(((w[i] >> 8) & M) | E) + BIAS;        // bytes 1,3 low nibbles
        const f16x2 c = bits_h2(((w[i] >> 4) & M) | E) + BIAS;        // bytes 0,2 high nibbles
        const f16x2 e = bits_h2(((w[i] >> 12) & M) | E) + BIAS;       // bytes 1,3 high nibbles
        ol[2 * i]     = h2_bits(__builtin_elementwise_fma(a, DS0, NO0));
        ol[2 * i + 1] = h2_bits(__builtin_elementwise_fma(b, DS0, NO0));
        oh[2 * i]     = h2_bits(__builtin_elementwise_fma(c, DS1, NO1));
        oh[2 * i + 1] = h2_bits(__builtin_elementwise_fma(e, DS1, NO1));
    }
    lo[0] = make_uint4(ol[0], ol[1], ol[2], ol[3]); lo[1] = make_uint4(ol[4], ol[5], ol[6], ol[7]);
    hi[0] = make_uint4(oh[0], oh[1], oh[2], oh[3]); hi[1] = make_uint4(oh[4], oh[5], oh[6], oh[7]);
}

// Q6_K the same way: the four 6-bit values of a byte column are assembled packed (ql nibble | qh pair << 4), OR-ed into
// 0x6400 as byte pairs (0,2) / (1,3), rebased by -(1024 + 32) (exact) and multiplied by the f16-rounded d * sc.
__device__ __forceinline__ void unpack_q6k_f16(const Unit<T_Q6_K> & wu, int g, uint4 (&out)[4][2]) {
    const float d = h2f(wu.d);
    f16x2 DS[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { const _Float16 t = (_Float16) (d * (float) wu.scale(g, r)); DS[r] = f16x2{ t, t }; }
    const f16x2 BIAS = { (_Float16) -1056.0f, (_Float16) -1056.0f };
    const uint32_t A[4] = { wu.qa.x, wu.qa.y, wu.qa.z, wu.qa.w }, B[4] = { wu.qb.x, wu.qb.y, wu.qb.z, wu.qb.w },
                   H[4] = { wu.qh.x, wu.qh.y, wu.qh.z, wu.qh.w };
    uint32_t o[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t m4 = 0x0f0f0f0fu, m2 = 0x30303030u, P = 0x00ff00ffu, E = 0x64006400u;
        const uint32_t v[4] = { (A[i] & m4) | ((H[i] << 4) & m2), (B[i] & m4) | ((H[i] << 2) & m2),
                                ((A[i] >> 4) & m4) | (H[i] & m2), ((B[i] >> 4) & m4) | ((H[i] >> 2) & m2) };
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const f16x2 a = (bits_h2((v[r] & P) | E) + BIAS) * DS[r];            // l = 4i, 4i+2
            const f16x2 b = (bits_h2(((v[r] >> 8) & P) | E) + BIAS) * DS[r];     // l = 4i+1, 4i+3
            o[r][2 * i] = h2_bits(a);
            o[r][2 * i + 1] = h2_bits(b);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        out[r][0] = make_uint4(o[r][0], o[r][1], o[r][2], o[r][3]);
        out[r][1] = make_uint4(o[r][4], o[r][5], o[r][6], o[r][7]);
    }
}

template <int T, bool F16> struct MfmaPerm4 { static constexpr bool value = false; };
// (Q4_K / Q6_K / Q4_0 in F16 mode run the register-B kernels of qmm_mfma_regb.cuh; this LDS-tile kernel serves Q5_K, Q8_0 and
//  the bf16 mode with the bit-exact to_f32 unpack in natural k order.  The packed unpacks above remain available: switch a type
//  to `true` here together with PERM 1 in launch_prep.)

// ------------------------------------------------------------------------------------------------
// stage 2.  Wave-specialized workgroup of 8 waves (512 threads) per BN x BM tile of dst:
//
//   waves 4-7  PRODUCERS   fetch(ks+3): one weight unit per thread straight from HBM + the matching Xh chunks
//                          stash(ks+1): unpack the unit bit-exactly to f32 (Unit<T>::to_f32), round to f16/bf16,
//                                       write both tiles into LDS stage (ks+1)&1 (XOR-swizzled, conflict-free)
//   waves 0-3  CONSUMERS   compute(ks): ds_read_b128 fragments from LDS stage ks&1 + v_mfma_f32_32x32x16_{f16,bf16}
//
// One barrier per K-step.  The two roles run different loops that meet at the same s_barrier count; on a SIMD the
// producer wave's VALU work and the consumer wave's MFMAs issue to different pipes and overlap (measured: with all
// roles in one wave per SIMD the three phases simply add up).  Loads are unconditional (K-step index clamped): a branch
// around a global_load makes hipcc drain vmcnt to 0.  Fetched data has two K-steps of latency cover.
template <int T, int BM, bool F16>
__global__ void __launch_bounds__(512)
mfma_kernel(const uint8_t * __restrict__ W, const int64_t row_bytes, const int64_t expert_bytes, const int M, const int K,
            const uint16_t * __restrict__ Xh, const int Kp, const float * __restrict__ scale,
            const int * __restrict__ seg_start, const int * __restrict__ seg_count, const int N,
            float * __restrict__ dst, const int64_t ldd, const int64_t * __restrict__ dst_off) {
    constexpr int BN = 128;
    constexpr int BK = MfmaBK<T>::value;
    constexpr int ROWB = BK * 2;
    constexpr int SLOTS = BK / 8;                            // 16-byte slots per tile row
    constexpr int UPS = BK / Traits<T>::UNIT_W;              // weight units per row per K-step (= 2)
    constexpr int W_UNITS = BM * UPS;                        // units per K-step (128 or 256)
    constexpr int X_CHUNKS = BN * SLOTS / 256;               // 16-byte chunks per producer thread per K-step
    constexpr int RT = BM / 64;                              // 32-row MFMA tiles per consumer wave along the weight rows
    constexpr int STAGE = (BM + BN) * ROWB;
    static_assert(UPS == 2, "unit/K-step geometry");

    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];      // 2 stages of [Ws | Xs]

    const int seg0 = seg_start ? seg_start[blockIdx.z] : 0;
    const int segn = seg_count ? seg_count[blockIdx.z] : N;
    const int tok0 = blockIdx.y * BN;                        // within the segment
    if (tok0 >= segn) return;
    const int row0 = blockIdx.x * BM;
    const int nk = Kp / BK;                                  // even: Kp is a multiple of 128
    const int dbg = g_mfma_dbg;

    if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) {
        // ------------------------------------------------------------------ producers
        const int tid = threadIdx.x - 256;
        const uint8_t * Wz = W + (int64_t) blockIdx.z * expert_bytes;
        const bool has_w = tid < W_UNITS;
        const int  wr = tid >> 1;                            // rows past M re-read row M-1 (never stored)
        const uint8_t * wrow = Wz + (int64_t) min(row0 + min(wr, BM - 1), M - 1) * row_bytes;
        const int units_per_row = K / Traits<T>::UNIT_W;
        const uint8_t * xthr = reinterpret_cast<const uint8_t *>(Xh + (int64_t) (seg0 + tok0) * Kp) +
                               (size_t) (tid / SLOTS) * Kp * 2 + (tid % SLOTS) * 16;
        const int xrow_step = (256 / SLOTS) * Kp * 2;        // bytes between this thread's consecutive chunks

        struct Regs { Unit<T> wu; uint4 xc[X_CHUNKS]; };
        auto fetch = [&](Regs & r, int ks_raw) {
            const int ks = min(ks_raw, nk - 1);
            const int u = ks * UPS + (tid & 1);
            r.wu.load(wrow, min(u, units_per_row - 1));
            if (u >= units_per_row) r.wu.kill();              // K tail of a padded K-step: scales -> 0, values -> 0
            const uint8_t * xp = xthr + ks * (BK * 2);
#pragma unroll
            for (int i = 0; i < X_CHUNKS; ++i) r.xc[i] = *reinterpret_cast<const uint4 *>(xp + i * xrow_step);
        };
        auto stash = [&](const Regs & r, int ks_raw, uint8_t * stage) {
            const int ks = min(ks_raw, nk - 1);
            uint8_t * Ws = stage;
            uint8_t * Xs = stage + BM * ROWB;
            const int u = min(ks * UPS + (tid & 1), units_per_row - 1);
            if constexpr (MfmaPerm4<T, F16>::value && T == T_Q6_K) {
                if (has_w) {
                    uint4 o[4][2];
                    unpack_q6k_f16(r.wu, u & 1, o);
                    const int s0 = (Unit<T>::k_run(tid & 1, 0) & (BK - 1)) >> 3;  // run r sits 32 k = 4 slots further each
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        *reinterpret_cast<uint4 *>(Ws + tile_off<BK>(wr, s0 + 4 * rr))     = o[rr][0];
                        *reinterpret_cast<uint4 *>(Ws + tile_off<BK>(wr, s0 + 4 * rr + 1)) = o[rr][1];
                    }
                }
            } else if constexpr (MfmaPerm4<T, F16>::value) {
                if (has_w) {
                    uint4 lo[2], hi[2];
                    unpack_q4k_f16(r.wu, (u & 7) >> 1, lo, hi);
                    const int kk = Unit<T>::k_run(tid & 1, 0) & (BK - 1);         // low-nibble run; the high-nibble run is 32 further
                    *reinterpret_cast<uint4 *>(Ws + tile_off<BK>(wr, (kk >> 3)))     = lo[0];
                    *reinterpret_cast<uint4 *>(Ws + tile_off<BK>(wr, (kk >> 3) + 1)) = lo[1];
                    *reinterpret_cast<uint4 *>(Ws + tile_off<BK>(wr, (kk >> 3) + 4)) = hi[0];
                    *reinterpret_cast<uint4 *>(Ws + tile_off<BK>(wr, (kk >> 3) + 5)) = hi[1];
                }
            } else if (has_w) {
                float v[Traits<T>::UNIT_W];
                r.wu.to_f32(u, v);
#pragma unroll
                for (int rr = 0; rr < Unit<T>::RUNS; ++rr) {
                    const int kk = Unit<T>::k_run(tid & 1, rr) & (BK - 1);   // offset inside the K-step window
#pragma unroll
                    for (int e = 0; e < Unit<T>::RUN_LEN; e += 8) {
                        uint4 o;
                        o.x = pack16<F16>(v[rr * Unit<T>::RUN_LEN + e + 0], v[rr * Unit<T>::RUN_LEN + e + 1]);
                        o.y = pack16<F16>(v[rr * Unit<T>::RUN_LEN + e + 2], v[rr * Unit<T>::RUN_LEN + e + 3]);
                        o.z = pack16<F16>(v[rr * Unit<T>::RUN_LEN + e + 4], v[rr * Unit<T>::RUN_LEN + e + 5]);
                        o.w = pack16<F16>(v[rr * Unit<T>::RUN_LEN + e + 6], v[rr * Unit<T>::RUN_LEN + e + 7]);
                        *reinterpret_cast<uint4 *>(Ws + tile_off<BK>(wr, (kk + e) >> 3)) = o;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < X_CHUNKS; ++i) {
                const int c = tid + 256 * i;
                *reinterpret_cast<uint4 *>(Xs + tile_off<BK>(c / SLOTS, c % SLOTS)) = r.xc[i];
            }
        };

        // register ring: PF K-steps of loads in flight per thread (HBM/L2 latency under load is ~1 us, a K-step ~0.2 us)
#ifndef QMM_MFMA_PF
#define QMM_MFMA_PF 2
#endif
        constexpr int PF = QMM_MFMA_PF;
        Regs ring[PF];
#pragma unroll
        for (int i = 0; i < PF; ++i) fetch(ring[i], i);
        stash(ring[0], 0, lds);
        fetch(ring[0], PF);
        __syncthreads();
        for (int ks = 0; ks < nk; ks += PF) {                 // indices are clamped; iterations past nk are skipped
#pragma unroll
            for (int i = 1; i <= PF; ++i) {
                if (ks + i - 1 < nk) {                        // uniform; matches the consumers' barrier count
                    const int slot = i % PF;
                    if (!(dbg & 1)) stash(ring[slot], ks + i, lds + ((ks + i) & 1) * STAGE);
                    if (!(dbg & 2)) fetch(ring[slot], ks + i + PF);
                    __syncthreads();
                }
            }
        }
        // both LDS stages are free now: stage the tile's per-token scales and dst row offsets for the epilogue
        float *   sc_lds  = reinterpret_cast<float *>(lds);
        int64_t * off_lds = reinterpret_cast<int64_t *>(lds + 1024);
        if (tid < BN) {
            const int t = tok0 + tid;
            const bool live = t < segn;
            sc_lds[tid]  = live ? scale[seg0 + t] : 0.0f;
            off_lds[tid] = live ? (dst_off ? dst_off[seg0 + t] : (int64_t) t * ldd) : 0;
        }
        __syncthreads();
        return;
    }

    // ---------------------------------------------------------------------- consumers
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_t = (wave & 1) * 64;                      // token offset of the wave inside the tile
    const int wave_r = (wave >> 1) * (BM / 2);               // weight-row offset

    f32x16 acc[2][RT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < RT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    auto compute = [&](const uint8_t * stage) {
        const uint8_t * Ws = stage;
        const uint8_t * Xs = stage + BM * ROWB;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            const int slot = kk * 2 + (lane >> 5);
            uint4 a[2], b[RT];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const uint4 *>(Xs + tile_off<BK>(wave_t + 32 * i + (lane & 31), slot));
#pragma unroll
            for (int j = 0; j < RT; ++j) b[j] = *reinterpret_cast<const uint4 *>(Ws + tile_off<BK>(wave_r + 32 * j + (lane & 31), slot));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < RT; ++j) {
                    if (F16)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8 *>(&a[i]),
                                                                          *reinterpret_cast<const f16x8 *>(&b[j]), acc[i][j], 0, 0, 0);
                    else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8 *>(&a[i]),
                                                                           *reinterpret_cast<const bf16x8 *>(&b[j]), acc[i][j], 0, 0, 0);
                }
        }
    };

    __syncthreads();
    for (int ks = 0; ks < nk; ks += 2) {
        if (!(dbg & 4)) compute(lds);
        __syncthreads();
        if (!(dbg & 4)) compute(lds + STAGE);
        __syncthreads();
    }

    // epilogue: D[i = token][j = weight row]; lane -> weight row, registers -> tokens.  Scales / row offsets come from
    // LDS (staged by the producers) so that no dependent global load sits in front of the stores.
    __syncthreads();
    const float *   sc_lds  = reinterpret_cast<const float *>(lds);
    const int64_t * off_lds = reinterpret_cast<const int64_t *>(lds + 1024);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int tl = wave_t + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            if (tok0 + tl < segn) {
                const float sc = sc_lds[tl];
                float * drow = dst + off_lds[tl];
#pragma unroll
                for (int j = 0; j < RT; ++j) {
                    const int m = row0 + wave_r + 32 * j + (lane & 31);
                    if (m < M) drow[m] = acc[i][j][e] * sc;
                }
            }
        }
}

// ------------------------------------------------------------------------------------------------ host

inline int mfma_kpad(int64_t K) { return (int) ((K + 127) / 128 * 128); }
inline int mfma_npad(int64_t N) { return (int) ((N + 127) / 128 * 128); }

struct MfmaOperand {            // a prepared activation matrix in the workspace
    const uint16_t * xh;
    const float *    scale;
    int              Kp;
    int              frag_major;   // layout of xh: 0 = rows of Kp, 1 = fragment-major (frag_major_chunk)
    int              ksplit = 1;   // > 1: K is split over this many workgroups per tile (256 rows x 128 tokens), partial
    float *          part = nullptr;   //  tiles go to part[range][token][row] (mfma_splitk, splitk_reduce_kernel)
};

template <int ACT>
inline int launch_prep(qmm_ctx * c, hipStream_t st, int type, const float * x, int64_t ldx, const int64_t * gather, const int * n_dev,
                       int n_rows, int n_pad, int K, int Kp, int frag_major, uint16_t * xh, float * scale) {
    const size_t lds = (size_t) K + (size_t) (K / 32) * 4 + 64;
    const float * x2 = gather ? nullptr : c->prep_x2;                        // set by qmm_mul_mat_swiglu_in around the call
    if (x2 && c->prec != QMM_PREC_F16_Q8) return fail(QMM_EUNSUPPORTED, "SwiGLU input: only in the default (f16 on Q8 activations) prefill mode");
#define QMM_PREP(PERMv)                                                                                                            \
    hipLaunchKernelGGL((prep_act_kernel<ACT, true, PERMv>), dim3(n_pad), dim3(256), lds, st, x, ldx, gather, n_dev, n_rows, K, Kp,     \
                       c->act_mode, frag_major, xh, scale, x2, c->prep_ldx2)
    if (c->prec == QMM_PREC_F16_Q8 && (type == T_Q4_K || type == T_Q5_K)) QMM_PREP(2);      // register-B lane orders (qmm_mfma_regb.cuh)
    else if (c->prec == QMM_PREC_F16_Q8 && type == T_Q8_0) QMM_PREP(5);
    else if (c->prec == QMM_PREC_F16_Q8 && type == T_Q6_K) QMM_PREP(3);
    else if (c->prec == QMM_PREC_F16_Q8 && type == T_Q4_0) QMM_PREP(4);
    else if (c->prec == QMM_PREC_F16_Q8)
        hipLaunchKernelGGL((prep_act_kernel<ACT, true, 0>), dim3(n_pad), dim3(256), lds, st, x, ldx, gather, n_dev, n_rows, K, Kp,
                           c->act_mode, frag_major, xh, scale, x2, c->prep_ldx2);
    else
        hipLaunchKernelGGL((prep_act_kernel<ACT, false, 0>), dim3(n_pad), dim3(256), 0, st, x, ldx, gather, n_dev, n_rows, K, Kp,
                           c->act_mode, frag_major, xh, scale);
#undef QMM_PREP
    HIP_TRY(hipGetLastError());
    return QMM_OK;
}

template <int T>
inline int launch_mfma(qmm_ctx * c, hipStream_t st, const void * W, int64_t rb, int64_t eb, int n_expert, int M, int K,
                       const MfmaOperand & op, const int * seg_start, const int * seg_count, int N, int n_tiles_y,
                       float * dst, int64_t ldd, const int64_t * dst_off) {
    const bool f16 = c->prec == QMM_PREC_F16_Q8;
    // 64-row tiles when 128-row tiles would leave CUs idle
    const bool small = (int64_t) ((M + 127) / 128) * n_tiles_y * n_expert < c->cus;
    const dim3 block(512);
#define QMM_LAUNCH(BMv, F16v)                                                                                                      \
    do {                                                                                                                           \
        auto kern = mfma_kernel<T, BMv, F16v>;                                                                                     \
        const size_t lds = (size_t) 2 * (BMv + 128) * MfmaBK<T>::value * 2;                                                       \
        if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); \
        hipLaunchKernelGGL(kern, dim3((M + BMv - 1) / BMv, n_tiles_y, n_expert), block, lds, st, (const uint8_t *) W,             \
                           rb, eb, M, K, op.xh, op.Kp, op.scale, seg_start, seg_count, N, dst, ldd, dst_off);                      \
    } while (0)
    if (small) { if (f16) QMM_LAUNCH(64, true); else QMM_LAUNCH(64, false); }
    else       { if (f16) QMM_LAUNCH(128, true); else QMM_LAUNCH(128, false); }
#undef QMM_LAUNCH
    HIP_TRY(hipGetLastError());
    return QMM_OK;
}

bool mfma_regb_supports(const qmm_ctx * c, int type);
// few tokens: the split-K kernel of qmm_mfma_regb.cuh, which wants the operand fragment-major
// (measured on MI355X: it wins up to 64 tokens on any matrix, and up to 128 tokens on matrices of <= 8192 rows, which give
// the tiled kernels too few workgroups to fill the chip; from 129 tokens the tiled kernel with split-K is level or ahead)
inline bool mfma_use_skinny(const qmm_ctx * c, int type, int64_t N, int64_t M, int64_t n_expert = 1) {
    if (!mfma_regb_supports(c, type) || !c->skinny) return false;
    return N <= c->skinny_max_n || (N <= c->skinny_max_n_few && (M + 31) / 32 * n_expert <= c->cus);
}
struct RegbMore;
int  launch_mfma_regb(qmm_ctx * c, hipStream_t st, int type, const void * W, int64_t rb, int64_t eb, int n_expert, int M, int K,
                      const MfmaOperand & op, const int * seg_start, const int * seg_count, int N, int n_tiles_y,
                      float * dst, int64_t ldd, const int64_t * dst_off, const RegbMore * group = nullptr);

inline int launch_mfma_any(qmm_ctx * c, hipStream_t st, int type, const void * W, int64_t rb, int64_t eb, int n_expert, int M, int K,
                           const MfmaOperand & op, const int * seg_start, const int * seg_count, int N, int n_tiles_y,
                           float * dst, int64_t ldd, const int64_t * dst_off) {
    if (mfma_regb_supports(c, type))
        return launch_mfma_regb(c, st, type, W, rb, eb, n_expert, M, K, op, seg_start, seg_count, N, n_tiles_y, dst, ldd, dst_off);
    switch (type) {
        case T_Q4_0: return launch_mfma<T_Q4_0>(c, st, W, rb, eb, n_expert, M, K, op, seg_start, seg_count, N, n_tiles_y, dst, ldd, dst_off);
        case T_Q8_0: return launch_mfma<T_Q8_0>(c, st, W, rb, eb, n_expert, M, K, op, seg_start, seg_count, N, n_tiles_y, dst, ldd, dst_off);
        case T_Q4_K: return launch_mfma<T_Q4_K>(c, st, W, rb, eb, n_expert, M, K, op, seg_start, seg_count, N, n_tiles_y, dst, ldd, dst_off);
        case T_Q5_K: return launch_mfma<T_Q5_K>(c, st, W, rb, eb, n_expert, M, K, op, seg_start, seg_count, N, n_tiles_y, dst, ldd, dst_off);
        default:     return launch_mfma<T_Q6_K>(c, st, W, rb, eb, n_expert, M, K, op, seg_start, seg_count, N, n_tiles_y, dst, ldd, dst_off);
    }
}

// Split-K factor for the tiled register-B kernel on a plain MUL_MAT.  One 256-row x 128-token tile over K = 4096 is 27 us
// of MFMA time on its CU however few tiles there are, and 4096 x 4096 at 512 tokens is only 64 tiles; K is therefore cut
// into ranges (one workgroup each, partial tiles to the workspace, splitk_reduce_kernel adds them) until the launch has
// about as many workgroups as the chip has CUs.  Each extra range costs one more N x M f32 slab to write and read.
inline int mfma_splitk(const qmm_ctx * c, int type, int64_t M, int64_t K, int64_t N) {
    if (!mfma_regb_supports(c, type) || !c->splitk || mfma_use_skinny(c, type, N, M)) return 1;
    const int64_t tiles = (M + 255) / 256 * ((N + 127) / 128);
    if (tiles * 10 >= (int64_t) c->cus * 8) return 1;
    int64_t s = c->cus / tiles;
    if (s > 8) s = 8;
    if (c->splitk > 1 && s > c->splitk) s = c->splitk;       // (GGML_MI355X_SPLITK=n caps the factor; for experiments)
    if (s > K / 512) s = K / 512;
    return s < 2 ? 1 : (int) s;
}
inline size_t mfma_splitk_bytes(int ksplit, int64_t M, int64_t N) { return ksplit > 1 ? (size_t) ksplit * N * M * sizeof(float) : 0; }

// what the prepared operand depends on besides src1 itself: Q8_0 vs Q8_K emulation and the k-order of the unpack
inline int mfma_prep_key(const qmm_ctx * c, int type, int64_t N, int64_t M) {
    if (c->prec != QMM_PREC_F16_Q8) return 0;
    // activation format and k-order follow from the weight type (Q5_K shares Q4_K's); the layout from the kernel choice
    return 2 * (1 + (type == T_Q5_K ? T_Q4_K : type)) + (mfma_use_skinny(c, type, N, M) ? 1 : 0);
}

// plain MUL_MAT, N > 8.  `reuse_prep`: the previous call of a group already prepared the same src1 with the same key.
inline int mfma_mul_mat(qmm_ctx * c, hipStream_t st, int type, const void * W, int64_t rb, int64_t K, int64_t M,
                        const float * x, int64_t N, int64_t ldx, float * dst, int64_t ldd, bool reuse_prep) {
    const int Kp = mfma_kpad(K), Np = mfma_npad(N);
    const size_t xh_bytes = (size_t) Np * Kp * 2;
    const int ksplit = mfma_splitk(c, type, M, K, N);
    const size_t sc_bytes = ((size_t) Np * 4 + 255) & ~(size_t) 255;
    const size_t need = xh_bytes + sc_bytes + mfma_splitk_bytes(ksplit, M, N) + 256;
    int rc = ensure_ws(c, need);
    if (rc) return rc;
    uint16_t * xh = (uint16_t *) c->ws;
    float * scale = (float *) ((uint8_t *) c->ws + xh_bytes);
    const bool q8_0 = (type == T_Q4_0 || type == T_Q8_0);
    const int frag = mfma_use_skinny(c, type, N, M);
    if (!reuse_prep) {           // the caller guarantees: same src1, same activation format and k-order as the previous call
        rc = q8_0 ? launch_prep<T_Q8_0>(c, st, type, x, ldx, nullptr, nullptr, (int) N, Np, (int) K, Kp, frag, xh, scale)
                  : launch_prep<T_Q8_K>(c, st, type, x, ldx, nullptr, nullptr, (int) N, Np, (int) K, Kp, frag, xh, scale);
        if (rc) return rc;
    }
    MfmaOperand op = { xh, scale, Kp, frag };
    op.ksplit = ksplit;
    op.part = reinterpret_cast<float *>((uint8_t *) c->ws + xh_bytes + sc_bytes);
    return launch_mfma_any(c, st, type, W, rb, 0, 1, (int) M, (int) K, op, nullptr, nullptr, (int) N, Np / 128, dst, ldd, nullptr);
}

} // namespace qmm
