// qmm_act.cuh — activation quantizers (f32 -> Q8_0 / Q8_K fields) as device functions that write
// through whatever pointers they are given: LDS inside the mat-vec kernel, global in the standalone
// qmm_quantize_act kernel.  Bit-exact with the reference quantizers:
//   Q8_0  quantize_row_q8_0_ref  ggml/src/ggml-quants.c:194-217        (QMM_ACT_REF)
//         quantize_row_q8_0 AVX2 ggml/src/ggml-cpu/ggml-cpu-quants.c:806-870 (QMM_ACT_X86)
//   Q8_K  quantize_row_q8_K_ref  ggml/src/ggml-quants.c:2479-2516
// Device layout (structure of arrays): q int8 [K], d f32 [K/32 | K/256], bsum int16 [K/16].
#pragma once

#include "qmm_device.cuh"

namespace qmm {

__device__ __forceinline__ uint32_t pack4(int a, int b, int c, int d) {
    return (uint32_t) (a & 0xff) | ((uint32_t) (b & 0xff) << 8) | ((uint32_t) (c & 0xff) << 16) | ((uint32_t) (d & 0xff) << 24);
}

// One Q8_0 block (32 floats) is handled by 8 consecutive lanes, 4 floats each.
// `v` = this lane's 4 values; `sub` = lane index within its group of 8 (only for the store slot).
__device__ __forceinline__ void q8_0_block(const float4 v, int act_mode, uint32_t & packed, float & d_out) {
#pragma clang fp contract(off)
    float amax = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    amax = fmaxf(amax, dpp_mov<DPP_QUAD_X1>(amax));          // the block's 8 lanes: two quads ...
    amax = fmaxf(amax, dpp_mov<DPP_QUAD_X2>(amax));
    amax = fmaxf(amax, dpp_mov<DPP_ROW_HALF_MIRROR>(amax));   // ... mirrored within each group of 8
    const float d = __fdiv_rn(amax, 127.0f);
    d_out = __half2float(__float2half_rn(d));                  // y[i].d = GGML_FP32_TO_FP16(d)
    int q0, q1, q2, q3;
    if (act_mode == 1) {                                       // x86: id = 127/amax, round half to even
        const float id = amax != 0.0f ? __fdiv_rn(127.0f, amax) : 0.0f;
        q0 = (int) rintf(__fmul_rn(v.x, id)); q1 = (int) rintf(__fmul_rn(v.y, id));
        q2 = (int) rintf(__fmul_rn(v.z, id)); q3 = (int) rintf(__fmul_rn(v.w, id));
    } else {                                                   // ref: id = 1/d, roundf (half away)
        const float id = d != 0.0f ? __fdiv_rn(1.0f, d) : 0.0f;
        q0 = (int) roundf(__fmul_rn(v.x, id)); q1 = (int) roundf(__fmul_rn(v.y, id));
        q2 = (int) roundf(__fmul_rn(v.z, id)); q3 = (int) roundf(__fmul_rn(v.w, id));
    }
    packed = pack4(q0, q1, q2, q3);
}

// nearest_int (ggml-quants.c:372-377): two roundings.  hipcc's default -ffp-contract=fast would fuse the
// caller's multiply into this add (even through __fmul_rn/__fadd_rn, which are plain operators in the HIP
// headers) and change ties: contraction is switched off in these functions.
__device__ __forceinline__ int magic_round(float prod) {
#pragma clang fp contract(off)
    const float t = prod + 12582912.0f;
    return (int) (__float_as_uint(t) & 0x007fffffu) - 0x00400000;
}

// One Q8_K block (256 floats) is handled by a whole wave, lane l holding elements 4l..4l+3.
// Returns the packed 4 int8; d_out (all lanes); bsum = sum over the lane's group of 16 (BSG == 16: 4 lanes)
// or 32 (BSG == 32: 8 lanes) activations, valid in every lane of the group.
template <int BSG>
__device__ __forceinline__ void q8_K_block(const float4 v, int lane, uint32_t & packed, float & d_out, int & bsum) {
#pragma clang fp contract(off)      // the product must round to f32 BEFORE the magic add (hipcc contracts __fmul_rn/__fadd_rn too)
    // first element with the largest |x| decides the sign of the scale (strict '>' scan in the reference): the wave max
    // of |x| is taken on the bit patterns (no float-max canonicalisation), then the lowest lane that holds it supplies
    // its own first maximum
    const uint32_t ax = __float_as_uint(v.x) & 0x7fffffffu, ay = __float_as_uint(v.y) & 0x7fffffffu;
    const uint32_t az = __float_as_uint(v.z) & 0x7fffffffu, aw = __float_as_uint(v.w) & 0x7fffffffu;
    uint32_t bbits = ax;
    float bval = v.x;
    if (ay > bbits) { bbits = ay; bval = v.y; }
    if (az > bbits) { bbits = az; bval = v.z; }
    if (aw > bbits) { bbits = aw; bval = v.w; }
    const uint32_t gbits = wave_max_u32(bbits);
    const unsigned long long holders = __ballot(bbits == gbits);
    bval = readlane_f(bval, __builtin_amdgcn_readfirstlane((int) __ffsll((long long) holders) - 1));
    const float best = __uint_as_float(gbits);
    if (best == 0.0f) {            // reference: d = 0, qs = 0 (bsums left as they were; we define 0)
        packed = 0; d_out = 0.0f; bsum = 0;
        return;
    }
    const float iscale = __fdiv_rn(-127.0f, bval);
    const int q0 = min(127, magic_round(__fmul_rn(iscale, v.x)));
    const int q1 = min(127, magic_round(__fmul_rn(iscale, v.y)));
    const int q2 = min(127, magic_round(__fmul_rn(iscale, v.z)));
    const int q3 = min(127, magic_round(__fmul_rn(iscale, v.w)));
    packed = pack4(q0, q1, q2, q3);
    int s = q0 + q1 + q2 + q3;
    s += dpp_mov_i<DPP_QUAD_X1>(s);
    s += dpp_mov_i<DPP_QUAD_X2>(s);
    if (BSG == 32) s += dpp_mov_i<DPP_ROW_HALF_MIRROR>(s);
    bsum  = s;
    d_out = __fdiv_rn(1.0f, iscale);
}

// X2 instantiations: the activations are SwiGLU of two rows, silu(x) * x2 (llama.cpp build_ffn, LLM_FFN_SILU + LLM_FFN_PAR: the
// input of ffn_down), formed while staging; same float operations as the stand-alone SILU and MUL kernels.
template <bool X2>
__device__ __forceinline__ float4 act_fetch(const float * __restrict__ p, const float * __restrict__ p2) {
    float4 v = *reinterpret_cast<const float4 *>(p);
    if (X2) {
        const float4 u = *reinterpret_cast<const float4 *>(p2);
        v.x = v.x / (1.0f + expf(-v.x)) * u.x;
        v.y = v.y / (1.0f + expf(-v.y)) * u.y;
        v.z = v.z / (1.0f + expf(-v.z)) * u.z;
        v.w = v.w / (1.0f + expf(-v.w)) * u.w;
    }
    return v;
}

// Quantize `rows` activation rows of length K into (q, d, bsum).  All threads of the block take part;
// the caller synchronizes afterwards.  ACT = T_Q8_0 or T_Q8_K.  Row r of x starts at x + r*ldx;
// outputs for row r at q + r*K, d + r*(K/blk), bsum + r*(K/16).
template <int ACT, int BSG = 16, int SWZ = 0, bool X2 = false>
__device__ __forceinline__ void quantize_rows(const float * __restrict__ x, int64_t ldx, int rows, int K, int act_mode,
                                              int8_t * q, float * d, int16_t * bsum, int tid, int nthreads,
                                              const float * __restrict__ x2 = nullptr, int64_t ldx2 = 0) {
    if (ACT == T_Q8_0) {
        const int per_row = K / 4;                               // float4 slots per row
        const int total = rows * per_row;                        // multiple of 8
        for (int i = tid; i < ((total + nthreads - 1) / nthreads) * nthreads; i += nthreads) {
            const bool live = i < total;                         // keep the 8-lane groups converged
            const int r = live ? i / per_row : 0, c = live ? i % per_row : 0;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live) v = act_fetch<X2>(x + (int64_t) r * ldx + 4 * c, X2 ? x2 + (int64_t) r * ldx2 + 4 * c : nullptr);
            uint32_t p; float dd;
            q8_0_block(v, act_mode, p, dd);
            if (live) {
                *reinterpret_cast<uint32_t *>(q + (int64_t) r * K + act_pos<SWZ>(4 * c)) = p;
                if ((c & 7) == 0) d[(int64_t) r * (K / 32) + c / 8] = dd;
            }
        }
    } else {
        const int lane = tid & (WAVE - 1), wave = tid / WAVE, nwaves = nthreads / WAVE;
        const int nb = K / 256;
        constexpr int LPG = BSG / 4;                             // lanes per bsum group
        auto emit = [&](const float4 v, int8_t * qr, float * dr, int16_t * br, int b) {
            uint32_t p; float dd; int bs;
            q8_K_block<BSG>(v, lane, p, dd, bs);
            *reinterpret_cast<uint32_t *>(qr + act_pos<SWZ>(b * 256 + 4 * lane)) = p;
            if (lane == 0) dr[b] = dd;
            if (br && (lane & (LPG - 1)) == 0) br[b * (256 / BSG) + lane / LPG] = (int16_t) bs;
        };
        for (int r = 0; r < rows; ++r) {                         // rows is a small compile-time constant at the hot call sites
            const float * xr = x + (int64_t) r * ldx + 4 * lane;
            const float * x2r = X2 ? x2 + (int64_t) r * ldx2 + 4 * lane : nullptr;
            int8_t *  qr = q + (size_t) r * K;
            float *   dr = d + (size_t) r * nb;
            int16_t * br = bsum ? bsum + (size_t) r * (K / BSG) : nullptr;
            if (nb <= nwaves) {                                  // at most one block per wave (K = 4096 with 16 waves)
                if (wave < nb) emit(act_fetch<X2>(xr + wave * 256, X2 ? x2r + wave * 256 : nullptr), qr, dr, br, wave);
                continue;
            }
            // long rows (K = 14336: 56 blocks over 16 waves): four blocks' loads in flight per wave, unconditional + clamped
            for (int b0 = wave; b0 < nb; b0 += 4 * nwaves) {
                float4 v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int off = min(b0 + j * nwaves, nb - 1) * 256;
                    v[j] = act_fetch<X2>(xr + off, X2 ? x2r + off : nullptr);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (b0 + j * nwaves < nb) emit(v[j], qr, dr, br, b0 + j * nwaves);     // wave-uniform, no load inside
            }
        }
    }
}

// standalone kernel (parity tests, and the MFMA path's first stage uses its own variant)
template <int ACT>
__global__ void __launch_bounds__(256) quantize_act_kernel(const float * __restrict__ x, int64_t ldx, int rows, int K, int act_mode,
                                                           int8_t * q, float * d, int16_t * bsum, int rows_per_block) {
    const int r0 = blockIdx.x * rows_per_block;
    const int nr = min(rows_per_block, rows - r0);
    if (nr <= 0) return;
    quantize_rows<ACT>(x + (int64_t) r0 * ldx, ldx, nr, K, act_mode, q + (int64_t) r0 * K,
                       d + (int64_t) r0 * (K / (ACT == T_Q8_0 ? 32 : 256)),
                       bsum ? bsum + (int64_t) r0 * (K / 16) : nullptr, threadIdx.x, blockDim.x);
}

} // namespace qmm
