// qmm_mfma_regb.cuh — prefill kernel, "register-B" form (QMM_PREC_F16_Q8, all five weight types).
//
// Every wave owns 32 weight rows and ALL tokens of the tile.  A lane (row r = lane & 31, half h = lane >> 5) owns a fixed
// slice of the quant bytes of its row's current K-step: it loads them straight from HBM (16-byte loads), unpacks them
// in the packed-f16 domain (0x6400 | n is the f16 1024 + n; an exact rebias and ONE packed multiply or FMA with the
// f16-rounded block scale) and the results ARE the MFMA B fragments of the K-step's 16-deep MFMA k-steps.  No LDS round
// trip, no cross-lane movement, no producer/consumer split for the weights.  Only the activation tile (BN tokens x BK k,
// f16, shared by the workgroup's waves) goes through LDS, double-buffered; prep_act_kernel stores it in the k-order that
// the lane ownership implies (MFMA sums over k: any order shared by both operands is fine):
//
//   type  BK   lane's bytes per K-step                        MFMA k-step kk holds, for half h, k =            prep PERM
//   Q4_K  64   16 of the pair's 32 qs bytes (+ header)        (kk&1)*8 + 16h + 32*(kk>>1) + e'                 2
//   Q4_0  64   block h of the two: d + 16 qs bytes            32h + (kk&1)*8 + 16*(kk>>1) + e'                 4
//   Q5_K  64   as Q4_K + the 16 qh bytes of the same l        as Q4_K                                          2
//   Q8_0  64   block h of the two: d + 32 int8                32h + 8*kk + e'                                  5
//   Q6_K  128  16 ql[l], 16 ql[l+32], 16 qh[l], l = 16h..     32*(kk>>1) + 16h + 8*(kk&1) + e'                 3
//   with e' = (0,2,1,3,4,6,5,7)[e]: the packed unpack emits byte pairs (0,2), (1,3).
//
//   D[token][row] += A[token][k] * B[k][row]:  A = activations (BN/32 fragments per kk from LDS), B = this lane's registers.
// One barrier per K-step; with 8 waves per workgroup (256 rows) two waves share a SIMD and one wave's unpack VALU overlaps
// the other's MFMAs.  Measured alternatives that did NOT help: pinning fragment reads ahead of the MFMAs and interleaving
// the next K-step's unpack under the MFMAs with sched_group_barrier.
#pragma once

#include <type_traits>

#include "qmm_mfma.cuh"
#include "qmm_mvunit.cuh"

namespace qmm {

constexpr uint32_t RB_M4 = 0x000f000fu, RB_M8 = 0x00ff00ffu, RB_E = 0x64006400u;

// (x & mask) | exponent-bits is ONE v_and_or_b32 only if at most one of its two constants is a literal: gfx9 allows a single
// literal / scalar operand per VALU instruction, and with both written as literals hipcc emits v_and + v_or.  Holding the
// exponent pattern in a VGPR (opaque to constant propagation) gets the fused form: 64 fewer VALU ops per Q4_K block and wave.
__device__ __forceinline__ uint32_t vgpr_const(uint32_t v) { asm("" : "+v"(v)); return v; }   // not volatile: may be hoisted and shared

template <int T> struct Regb;

// ---- Q4_K ------------------------------------------------------------------------------------------------------------
template <> struct Regb<T_Q4_K> {
    static constexpr int BK = 64, NFRAG = 4, PERM = 2;
    struct Raw { uint4 qs, hdr; };
    static __device__ __forceinline__ Raw load(const uint8_t * wrow, int ks, int h, int K) {
        const int k2 = min(ks, K / 64 - 1);
        const uint8_t * blk = wrow + (size_t) (k2 >> 2) * 144;
        Raw r;
        r.hdr = ldg<uint4>(blk);
        r.qs  = ldg<uint4>(blk + 16 + 32 * (k2 & 3) + 16 * h);
        if (ks > k2) r.hdr.x = 0;                               // past K: d = dmin = 0 -> zeros
        return r;
    }
    static __device__ __forceinline__ void unpack(const Raw & w, int ks, uint4 (&f)[4]) {
        uint32_t sc, mn;
        k4_pair(w.hdr, ks & 3, sc, mn);
        const float d = h2f(w.hdr.x & 0xffff), dmin = h2f(w.hdr.x >> 16);
        const _Float16 ds0 = (_Float16) (d * (float) (sc & 0xff)), ds1 = (_Float16) (d * (float) (sc >> 8));
        const _Float16 no0 = (_Float16) (-(dmin * (float) (mn & 0xff))), no1 = (_Float16) (-(dmin * (float) (mn >> 8)));
        const f16x2 DS0 = { ds0, ds0 }, DS1 = { ds1, ds1 }, NO0 = { no0, no0 }, NO1 = { no1, no1 };
        const f16x2 BIAS = { (_Float16) -1024.0f, (_Float16) -1024.0f };
        const uint32_t E = vgpr_const(RB_E);
        const uint32_t q[4] = { w.qs.x, w.qs.y, w.qs.z, w.qs.w };
        uint32_t lo[8], hi[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            lo[2 * i]     = h2_bits(__builtin_elementwise_fma(bits_h2((q[i] & RB_M4) | E) + BIAS, DS0, NO0));          // bytes 4i, 4i+2
            lo[2 * i + 1] = h2_bits(__builtin_elementwise_fma(bits_h2(((q[i] >> 8) & RB_M4) | E) + BIAS, DS0, NO0));   // bytes 4i+1, 4i+3
            hi[2 * i]     = h2_bits(__builtin_elementwise_fma(bits_h2(((q[i] >> 4) & RB_M4) | E) + BIAS, DS1, NO1));
            hi[2 * i + 1] = h2_bits(__builtin_elementwise_fma(bits_h2(((q[i] >> 12) & RB_M4) | E) + BIAS, DS1, NO1));
        }
        f[0] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        f[1] = make_uint4(lo[4], lo[5], lo[6], lo[7]);
        f[2] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        f[3] = make_uint4(hi[4], hi[5], hi[6], hi[7]);
    }
};

// ---- Q4_0: value = (n - 8) * d, d is already an f16: one rounding ------------------------------------------------------
template <> struct Regb<T_Q4_0> {
    static constexpr int BK = 64, NFRAG = 4, PERM = 4;
    struct Raw { uint4 qs; uint32_t d; };
    static __device__ __forceinline__ Raw load(const uint8_t * wrow, int ks, int h, int K) {
        const int b = 2 * ks + h, bc = min(b, K / 32 - 1);
        const uint8_t * blk = wrow + (size_t) bc * 18;
        Raw r;
        r.d  = b > bc ? 0u : (uint32_t) ldg<uint16_t>(blk);
        r.qs = ldg<uint4>(blk + 2);
        return r;
    }
    static __device__ __forceinline__ void unpack(const Raw & w, int, uint4 (&f)[4]) {
        const _Float16 d = __builtin_bit_cast(_Float16, (unsigned short) w.d);
        const f16x2 D = { d, d }, BIAS = { (_Float16) -1032.0f, (_Float16) -1032.0f };
        const uint32_t E = vgpr_const(RB_E);
        const uint32_t q[4] = { w.qs.x, w.qs.y, w.qs.z, w.qs.w };
        uint32_t lo[8], hi[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            lo[2 * i]     = h2_bits((bits_h2((q[i] & RB_M4) | E) + BIAS) * D);
            lo[2 * i + 1] = h2_bits((bits_h2(((q[i] >> 8) & RB_M4) | E) + BIAS) * D);
            hi[2 * i]     = h2_bits((bits_h2(((q[i] >> 4) & RB_M4) | E) + BIAS) * D);
            hi[2 * i + 1] = h2_bits((bits_h2(((q[i] >> 12) & RB_M4) | E) + BIAS) * D);
        }
        f[0] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        f[1] = make_uint4(lo[4], lo[5], lo[6], lo[7]);
        f[2] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        f[3] = make_uint4(hi[4], hi[5], hi[6], hi[7]);
    }
};

// ---- Q5_K: Q4_K plus one high bit per weight from qh[l] (bit 2j for the low-nibble run, 2j+1 for the high-nibble run) ------
template <> struct Regb<T_Q5_K> {
    static constexpr int BK = 64, NFRAG = 4, PERM = 2;
    struct Raw { uint4 qs, qh, hdr; };
    static __device__ __forceinline__ Raw load(const uint8_t * wrow, int ks, int h, int K) {
        const int k2 = min(ks, K / 64 - 1);
        const uint8_t * blk = wrow + (size_t) (k2 >> 2) * 176;
        Raw r;
        r.hdr = ldg<uint4>(blk);
        r.qh  = ldg<uint4>(blk + 16 + 16 * h);
        r.qs  = ldg<uint4>(blk + 48 + 32 * (k2 & 3) + 16 * h);
        if (ks > k2) r.hdr.x = 0;
        return r;
    }
    static __device__ __forceinline__ void unpack(const Raw & w, int ks, uint4 (&f)[4]) {
        const int j = ks & 3;
        uint32_t sc, mn;
        k4_pair(w.hdr, j, sc, mn);
        const float d = h2f(w.hdr.x & 0xffff), dmin = h2f(w.hdr.x >> 16);
        const _Float16 ds0 = (_Float16) (d * (float) (sc & 0xff)), ds1 = (_Float16) (d * (float) (sc >> 8));
        const _Float16 no0 = (_Float16) (-(dmin * (float) (mn & 0xff))), no1 = (_Float16) (-(dmin * (float) (mn >> 8)));
        const f16x2 DS0 = { ds0, ds0 }, DS1 = { ds1, ds1 }, NO0 = { no0, no0 }, NO1 = { no1, no1 };
        const f16x2 BIAS = { (_Float16) -1024.0f, (_Float16) -1024.0f };
        const uint32_t E = vgpr_const(RB_E);
        const uint32_t q[4] = { w.qs.x, w.qs.y, w.qs.z, w.qs.w }, g[4] = { w.qh.x, w.qh.y, w.qh.z, w.qh.w };
        const uint32_t ONE = 0x00010001u;
        uint32_t lo[8], hi[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t gl = g[i] >> (2 * j), gh = g[i] >> (2 * j + 1);
            lo[2 * i]     = h2_bits(__builtin_elementwise_fma(bits_h2((q[i] & RB_M4) | ((gl & ONE) << 4) | E) + BIAS, DS0, NO0));
            lo[2 * i + 1] = h2_bits(__builtin_elementwise_fma(bits_h2(((q[i] >> 8) & RB_M4) | (((gl >> 8) & ONE) << 4) | E) + BIAS, DS0, NO0));
            hi[2 * i]     = h2_bits(__builtin_elementwise_fma(bits_h2(((q[i] >> 4) & RB_M4) | ((gh & ONE) << 4) | E) + BIAS, DS1, NO1));
            hi[2 * i + 1] = h2_bits(__builtin_elementwise_fma(bits_h2(((q[i] >> 12) & RB_M4) | (((gh >> 8) & ONE) << 4) | E) + BIAS, DS1, NO1));
        }
        f[0] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        f[1] = make_uint4(lo[4], lo[5], lo[6], lo[7]);
        f[2] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        f[3] = make_uint4(hi[4], hi[5], hi[6], hi[7]);
    }
};

// ---- Q8_0: int8 b -> f16 through (b ^ 0x80) | 0x6400 = 1024 + 128 + b; value = b * d with d an f16: one rounding ---------
template <> struct Regb<T_Q8_0> {
    static constexpr int BK = 64, NFRAG = 4, PERM = 5;
    struct Raw { uint4 q0, q1; uint32_t d; };
    static __device__ __forceinline__ Raw load(const uint8_t * wrow, int ks, int h, int K) {
        const int b = 2 * ks + h, bc = min(b, K / 32 - 1);
        const uint8_t * blk = wrow + (size_t) bc * 34;
        Raw r;
        r.d  = b > bc ? 0u : (uint32_t) ldg<uint16_t>(blk);
        r.q0 = ldg<uint4>(blk + 2);
        r.q1 = ldg<uint4>(blk + 18);
        return r;
    }
    static __device__ __forceinline__ void unpack(const Raw & w, int, uint4 (&f)[4]) {
        const _Float16 d = __builtin_bit_cast(_Float16, (unsigned short) w.d);
        const f16x2 D = { d, d }, BIAS = { (_Float16) -1152.0f, (_Float16) -1152.0f };
        const uint32_t E = vgpr_const(RB_E);
        const uint32_t q[8] = { w.q0.x, w.q0.y, w.q0.z, w.q0.w, w.q1.x, w.q1.y, w.q1.z, w.q1.w };
        uint32_t o[16];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t u = q[i] ^ 0x80808080u;
            o[2 * i]     = h2_bits((bits_h2((u & RB_M8) | E) + BIAS) * D);            // bytes 4i, 4i+2
            o[2 * i + 1] = h2_bits((bits_h2(((u >> 8) & RB_M8) | E) + BIAS) * D);     // bytes 4i+1, 4i+3
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) f[kk] = make_uint4(o[4 * kk], o[4 * kk + 1], o[4 * kk + 2], o[4 * kk + 3]);
    }
};

// ---- Q6_K: K-step = one 128-wide half of a block ------------------------------------------------------------------------
template <> struct Regb<T_Q6_K> {
    static constexpr int BK = 128, NFRAG = 8, PERM = 3;
    struct Raw { uint4 qa, qb, qh; uint2 sc8; uint32_t d; };
    static __device__ __forceinline__ Raw load(const uint8_t * wrow, int ks, int h, int K) {
        const int k2 = min(ks, K / 128 - 1);
        const uint8_t * blk = wrow + (size_t) (k2 >> 1) * 210;
        const int n = k2 & 1;
        Raw r;
        r.qa  = ldg<uint4>(blk + 64 * n + 16 * h);
        r.qb  = ldg<uint4>(blk + 64 * n + 32 + 16 * h);
        r.qh  = ldg<uint4>(blk + 128 + 32 * n + 16 * h);
        r.sc8 = ldg<uint2>(blk + 192 + 8 * n);
        r.d   = ks > k2 ? 0u : (uint32_t) ldg<uint16_t>(blk + 208);
        return r;
    }
    static __device__ __forceinline__ int scale(const Raw & w, int g, int rr) {        // sc[8n + g + 2rr]
        const int i = g + 2 * rr;
        const uint32_t v = i < 4 ? w.sc8.x : w.sc8.y;
        return (int) (int8_t) ((v >> ((i & 3) * 8)) & 0xff);
    }
    // fragments 4*HF .. 4*HF+3 (sub-blocks rr = 2*HF, 2*HF+1 of the lane's four): HF = 0 low nibbles, HF = 1 high nibbles
    template <int HF>
    static __device__ __forceinline__ void unpack_half(const Raw & w, int h, uint4 (&f)[4]) {
        const float d = h2f(w.d);
        const f16x2 BIAS = { (_Float16) -1056.0f, (_Float16) -1056.0f };
        const uint32_t E = vgpr_const(RB_E);
        const uint32_t A[4] = { w.qa.x, w.qa.y, w.qa.z, w.qa.w }, B[4] = { w.qb.x, w.qb.y, w.qb.z, w.qb.w },
                       H[4] = { w.qh.x, w.qh.y, w.qh.z, w.qh.w };
        uint32_t o[2][8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t m4 = 0x0f0f0f0fu, m2 = 0x30303030u;
            const uint32_t v[2] = { HF == 0 ? (A[i] & m4) | ((H[i] << 4) & m2) : ((A[i] >> 4) & m4) | (H[i] & m2),
                                    HF == 0 ? (B[i] & m4) | ((H[i] << 2) & m2) : ((B[i] >> 4) & m4) | ((H[i] >> 2) & m2) };
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const _Float16 t = (_Float16) (d * (float) scale(w, h, 2 * HF + q));
                const f16x2 DS = { t, t };
                o[q][2 * i]     = h2_bits((bits_h2((v[q] & RB_M8) | E) + BIAS) * DS);            // l = 4i, 4i+2
                o[q][2 * i + 1] = h2_bits((bits_h2(((v[q] >> 8) & RB_M8) | E) + BIAS) * DS);     // l = 4i+1, 4i+3
            }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            f[2 * q]     = make_uint4(o[q][0], o[q][1], o[q][2], o[q][3]);
            f[2 * q + 1] = make_uint4(o[q][4], o[q][5], o[q][6], o[q][7]);
        }
    }
    static __device__ __forceinline__ void unpack_h(const Raw & w, int h, uint4 (&f)[8]) {
        uint4 lo[4], hi[4];
        unpack_half<0>(w, h, lo);
        unpack_half<1>(w, h, hi);
#pragma unroll
        for (int i = 0; i < 4; ++i) { f[i] = lo[i]; f[4 + i] = hi[i]; }
    }
};

// ---- split-K across workgroups (plain MUL_MAT on matrices with too few tiles to fill the chip) ----------------------------
// blockIdx.z = which K range.  A workgroup of range s stores its unscaled f32 tile to part[s][token][row] (plain coalesced
// stores, no flags); splitk_reduce_kernel then adds the ranges in order (deterministic), applies the per-token scale and
// writes dst.  (An in-kernel combine by the last workgroup to arrive at a tile counter was measured first: with 128 KB
// tiles its serial read of the other ranges cost more than the split gained; the extra launch is ~2 us and runs chip-wide.)
// Matrices 2..4 of a group that shares src1 and the weight type (attn_q / attn_k / attn_v): one launch walks the row tiles of
// all of them (the first matrix stays in the kernel's plain arguments).  tile_begin = first 256-row tile of the matrix in
// blockIdx.x, col0 = its first column in the split-K partial slabs [range][token][mtot].
struct RegbMore {
    const uint8_t * w[3];
    float *         dst[3];
    int64_t         row_bytes[3];
    int64_t         ldd[3];
    int             m[3];
    int             tile_begin[3];
    int             col0[3];
    int             n;            // how many of the three are used
    int             mtot;         // rows of all matrices together (>= M of the first)
};

__global__ void __launch_bounds__(256)
splitk_reduce_kernel(const float * __restrict__ part, const int ksplit, const int N, const int M, const float * __restrict__ scale,
                     float * __restrict__ dst, const int64_t ldd, const int vec, const RegbMore more) {
    const int mtot = more.n ? more.mtot : M;
    const int64_t range = (int64_t) N * mtot;
    const int64_t i = ((int64_t) blockIdx.x * 256 + threadIdx.x) * (vec ? 4 : 1);
    if (i >= range) return;
    const int t = (int) (i / mtot);
    const int col = (int) (i - (int64_t) t * mtot);
    float * out = dst;
    int64_t ld = ldd;
    int c0 = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k)
        if (k < more.n && col >= more.col0[k]) { out = more.dst[k]; ld = more.ldd[k]; c0 = more.col0[k]; }
    const int m = col - c0;
    const float sc = scale[t];
    if (vec) {                                                 // every M % 4 == 0, ldd % 4 == 0, dst 16-byte aligned
        float4 a = *reinterpret_cast<const float4 *>(part + i);
        for (int s = 1; s < ksplit; ++s) {
            const float4 b = *reinterpret_cast<const float4 *>(part + s * range + i);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        *reinterpret_cast<float4 *>(out + t * ld + m) = make_float4(a.x * sc, a.y * sc, a.z * sc, a.w * sc);
    } else {
        float a = part[i];
        for (int s = 1; s < ksplit; ++s) a += part[s * range + i];
        out[t * ld + m] = a * sc;
    }
}

template <int T, int NW, int BN>   // weight type, waves per workgroup, tokens per tile: tile = 32*NW weight rows x BN tokens
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(1, 2)))
mfma_regb_kernel(const uint8_t * __restrict__ W, const int64_t row_bytes, const int64_t expert_bytes, const int M, const int K,
                 const uint16_t * __restrict__ Xh, const int Kp, const float * __restrict__ scale,
                 const int * __restrict__ seg_start, const int * __restrict__ seg_count, const int N,
                 float * __restrict__ dst, const int64_t ldd, const int64_t * __restrict__ dst_off,
                 const int ksplit, float * __restrict__ part, const RegbMore more) {
    using P = Regb<T>;
    constexpr int BK = P::BK, NFRAG = P::NFRAG, ROWB = BK * 2, SLOTS = BK / 8, NA = BN / 32;
    constexpr int NT = NW * 64;
    constexpr int X_CHUNKS = BN * SLOTS / NT;                 // 16-byte chunks of the activation tile per thread per K-step
    constexpr int STAGE = BN * ROWB;
    static_assert(X_CHUNKS >= 1, "tile too small for the workgroup");

    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];      // 2 stages

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int expert = ksplit > 1 ? 0 : blockIdx.z, split = ksplit > 1 ? blockIdx.z : 0;
    const int seg0 = seg_start ? seg_start[expert] : 0;
    const int segn = seg_count ? seg_count[expert] : N;
    const int tok0 = blockIdx.y * BN;
    if (tok0 >= segn) return;                                 // (never with ksplit > 1: plain MUL_MAT has no empty tiles)
    // which matrix of the group this row tile belongs to (wave-uniform)
    const uint8_t * Wg = W;  float * dstg = dst;  int64_t rbg = row_bytes, lddg = ldd;  int Mg = M, col0 = 0, tile = blockIdx.x;
#pragma unroll
    for (int k = 0; k < 3; ++k)
        if (k < more.n && (int) blockIdx.x >= more.tile_begin[k]) {
            Wg = more.w[k];  dstg = more.dst[k];  rbg = more.row_bytes[k];  lddg = more.ldd[k];  Mg = more.m[k];  col0 = more.col0[k];
            tile = blockIdx.x - more.tile_begin[k];
        }
    const int mtot = more.n ? more.mtot : M;
    const int row0 = tile * (32 * NW) + wave * 32;
    const int r = lane & 31, h = lane >> 5;
    const uint8_t * wrow = Wg + (int64_t) expert * expert_bytes + (int64_t) min(row0 + r, Mg - 1) * rbg;
    const int nk = Kp / BK;                                   // even: Kp is a multiple of 128 (256 for the K-quants)
    const int per = ((nk + ksplit - 1) / ksplit + 1) & ~1;    // this workgroup's K-steps: [ks0, ks1), an even count
    const int ks0 = split * per, ks1 = min(ks0 + per, nk);

    const uint8_t * xthr = reinterpret_cast<const uint8_t *>(Xh + (int64_t) (seg0 + tok0) * Kp) +
                           (size_t) (tid / SLOTS) * Kp * 2 + (tid % SLOTS) * 16;
    const int xrow_step = (NT / SLOTS) * Kp * 2;

    f32x16 acc[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;

    // loads are unconditional; K-step indices are clamped at the end (re-reads valid bytes; scales zeroed past K)
    struct XRegs { uint4 c[X_CHUNKS]; };
    struct Frags { uint4 f[NFRAG]; };
    auto load_x = [&](int ks) {
        XRegs x;
        const uint8_t * xp = xthr + min(ks, nk - 1) * (BK * 2);
#pragma unroll
        for (int i = 0; i < X_CHUNKS; ++i) x.c[i] = *reinterpret_cast<const uint4 *>(xp + i * xrow_step);
        return x;
    };
    auto store_x = [&](const XRegs x, uint8_t * stage) {
#pragma unroll
        for (int i = 0; i < X_CHUNKS; ++i) {
            const int c = tid + NT * i;
            *reinterpret_cast<uint4 *>(stage + tile_off<BK>(c / SLOTS, c % SLOTS)) = x.c[i];
        }
    };
    struct AFr { uint4 a[NA]; };
    auto read_a = [&](const uint8_t * stage, int slot) {
        AFr f;
#pragma unroll
        for (int i = 0; i < NA; ++i) f.a[i] = *reinterpret_cast<const uint4 *>(stage + tile_off<BK>(32 * i + r, slot));
        return f;
    };
    auto mfma_a = [&](const uint4 b, const AFr f) {
        const f16x8 bb = *reinterpret_cast<const f16x8 *>(&b);
#pragma unroll
        for (int i = 0; i < NA; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8 *>(&f.a[i]), bb, acc[i], 0, 0, 0);
    };
    auto unpack = [&](const typename P::Raw & w, int ks) {
        Frags fr;
        if constexpr (T == T_Q6_K) P::unpack_h(w, h, fr.f);
        else                       P::unpack(w, ks, fr.f);
        return fr;
    };

    // software pipeline: weight bytes two K-steps ahead in registers (even/odd slots), activation chunks two K-steps ahead
    // in registers and one K-step ahead in LDS
    typename P::Raw w_e = P::load(wrow, ks0, h, K), w_o = P::load(wrow, ks0 + 1, h, K);
    XRegs x_e = load_x(ks0), x_o = load_x(ks0 + 1);
    store_x(x_e, lds);
    x_e = load_x(ks0 + 2);

    if constexpr (NFRAG == 4) {
        // The consumer and the pipeline of mfma_regb_q4k_kernel below (described there): B fragments one K-step ahead, the
        // upper half of the workgroup's waves multiplies first and unpacks second, A-fragment reads one k-step ahead of
        // the MFMAs.
        auto compute = [&](const Frags fr, const uint8_t * stage) {
            const AFr a0 = read_a(stage, 0 + h);
            const AFr a1 = read_a(stage, 2 + h);
            mfma_a(fr.f[0], a0);
            const AFr a2 = read_a(stage, 4 + h);
            mfma_a(fr.f[1], a1);
            const AFr a3 = read_a(stage, 6 + h);
            mfma_a(fr.f[2], a2);
            mfma_a(fr.f[3], a3);
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * NA, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, NA, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, NA, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, NA, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, NA, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * NA, 0);
        };
        Frags fr = unpack(w_e, ks0);
        w_e = P::load(wrow, ks0 + 2, h, K);
        __syncthreads();
        auto run = [&](auto unpack_first) {
            constexpr bool UF = decltype(unpack_first)::value;
            for (int ks = ks0; ks < ks1; ks += 2) {
                Frags frn;
                if (UF) frn = unpack(w_o, ks + 1);
                store_x(x_o, lds + STAGE);  x_o = load_x(ks + 3);
                compute(fr, lds);
                if (!UF) frn = unpack(w_o, ks + 1);
                w_o = P::load(wrow, ks + 3, h, K);
                __syncthreads();
                fr = frn;
                if (UF) frn = unpack(w_e, ks + 2);
                store_x(x_e, lds);          x_e = load_x(ks + 4);
                compute(fr, lds + STAGE);
                if (!UF) frn = unpack(w_e, ks + 2);
                w_e = P::load(wrow, ks + 4, h, K);
                __syncthreads();
                fr = frn;
            }
        };
        if (NW >= 8 && wave >= NW / 2) run(std::false_type{});
        else                           run(std::true_type{});
    } else {
        // Q6_K (eight fragments per K-step): the pipeline above does not fit the register file; unpack, then multiply.
        // (Measured and dropped: walking Q6_K in 64-wide half K-steps through the pipelined branch, which the PERM 3
        // activation order allows; the lane's 58 bytes are then requested once per half: 4096x14336x512 121 -> 163 us.)
        auto compute = [&](const Frags fr, const uint8_t * stage) {
#pragma unroll
            for (int kk = 0; kk < NFRAG; ++kk) mfma_a(fr.f[kk], read_a(stage, 2 * kk + h));
        };
        __syncthreads();
        for (int ks = ks0; ks < ks1; ks += 2) {
            Frags fr = unpack(w_e, ks);
            w_e = P::load(wrow, ks + 2, h, K);
            store_x(x_o, lds + STAGE);  x_o = load_x(ks + 3);
            compute(fr, lds);
            __syncthreads();
            fr = unpack(w_o, ks + 1);
            w_o = P::load(wrow, ks + 3, h, K);
            store_x(x_e, lds);          x_e = load_x(ks + 4);
            compute(fr, lds + STAGE);
            __syncthreads();
        }
    }

    // epilogue: per-token scales / dst row offsets staged through LDS (all reads of the tiles are behind the last barrier).
    // A split-K range stores its unscaled tile to part[split][token][row] instead (splitk_reduce_kernel finishes the job).
    float *   sc_lds  = reinterpret_cast<float *>(lds);
    int64_t * off_lds = reinterpret_cast<int64_t *>(lds + 1024);
    float *   out     = ksplit > 1 ? part + (int64_t) split * N * mtot + col0 : dstg;
    if (tid < BN) {
        const int t = tok0 + tid;
        const bool live = t < segn;
        sc_lds[tid]  = live ? (ksplit > 1 ? 1.0f : scale[seg0 + t]) : 0.0f;
        off_lds[tid] = live ? (ksplit > 1 ? (int64_t) t * mtot : dst_off ? dst_off[seg0 + t] : (int64_t) t * lddg) : 0;
    }
    __syncthreads();
    const int m = row0 + r;
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int tl = 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (tok0 + tl < segn && m < Mg) out[off_lds[tl] + m] = acc[i][e] * sc_lds[tl];
        }
}

// ---- Q4_K, specialised: one K-quant block = four K-steps with compile-time pair index, header loaded once per block ----
// (10-20 % faster than running Q4_K through the generic kernel above, which reloads and re-decodes the header every K-step)
// the lane's 16 bytes -> four B fragments (8 f16 each).  Element order inside a fragment: bytes (0,2,1,3,4,6,5,7).
struct RegbFrag { uint4 f0, f1, f2, f3; };

__device__ __forceinline__ RegbFrag regb_unpack_q4k(const uint4 qs, const uint4 hdr, int j) {
    int s0, m0, s1, m1;
    k4_scale_min(hdr, 2 * j, s0, m0);
    k4_scale_min(hdr, 2 * j + 1, s1, m1);
    const float d = h2f(hdr.x & 0xffff), dmin = h2f(hdr.x >> 16);
    const _Float16 ds0 = (_Float16) (d * (float) s0), ds1 = (_Float16) (d * (float) s1);
    const _Float16 no0 = (_Float16) (-(dmin * (float) m0)), no1 = (_Float16) (-(dmin * (float) m1));
    const f16x2 DS0 = { ds0, ds0 }, DS1 = { ds1, ds1 }, NO0 = { no0, no0 }, NO1 = { no1, no1 };
    const f16x2 BIAS = { (_Float16) -1024.0f, (_Float16) -1024.0f };
    const uint32_t w[4] = { qs.x, qs.y, qs.z, qs.w };
    // low nibbles: (b & 0xf) | 0x6400 = 1024 + n; high nibbles in place: (b & 0xf0) | 0x5400 = 64 + n (ulp of 64.0 is 1/16):
    // one shift per dword instead of three, same values
    const f16x2 BIAS_H = { (_Float16) -64.0f, (_Float16) -64.0f };
    const uint32_t E = vgpr_const(0x64006400u), EH = vgpr_const(0x54005400u);
    uint32_t lo[8], hi[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t M = 0x000f000fu, MH = 0x00f000f0u;
        const uint32_t w8 = w[i] >> 8;
        lo[2 * i]     = h2_bits(__builtin_elementwise_fma(bits_h2((w[i] & M) | E) + BIAS, DS0, NO0));         // bytes 4i, 4i+2
        lo[2 * i + 1] = h2_bits(__builtin_elementwise_fma(bits_h2((w8 & M) | E) + BIAS, DS0, NO0));           // bytes 4i+1, 4i+3
        hi[2 * i]     = h2_bits(__builtin_elementwise_fma(bits_h2((w[i] & MH) | EH) + BIAS_H, DS1, NO1));
        hi[2 * i + 1] = h2_bits(__builtin_elementwise_fma(bits_h2((w8 & MH) | EH) + BIAS_H, DS1, NO1));
    }
    RegbFrag fr;
    fr.f0 = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    fr.f1 = make_uint4(lo[4], lo[5], lo[6], lo[7]);
    fr.f2 = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    fr.f3 = make_uint4(hi[4], hi[5], hi[6], hi[7]);
    return fr;
}

template <int NW, int BN>   // waves per workgroup, tokens per tile: tile = 32*NW weight rows x BN tokens
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(1, 2)))
mfma_regb_q4k_kernel(const uint8_t * __restrict__ W, const int64_t row_bytes, const int64_t expert_bytes, const int M, const int K,
                     const uint16_t * __restrict__ Xh, const int Kp, const float * __restrict__ scale,
                     const int * __restrict__ seg_start, const int * __restrict__ seg_count, const int N,
                     float * __restrict__ dst, const int64_t ldd, const int64_t * __restrict__ dst_off,
                     const int ksplit, float * __restrict__ part, const RegbMore more) {
    constexpr int BK = 64, ROWB = BK * 2, SLOTS = 8, NA = BN / 32;
    constexpr int NT = NW * 64;
    constexpr int X_CHUNKS = BN * SLOTS / NT;                 // 16-byte chunks of the activation tile per thread per K-step
    constexpr int STAGE = BN * ROWB;                          // 16 KB

    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int expert = ksplit > 1 ? 0 : blockIdx.z, split = ksplit > 1 ? blockIdx.z : 0;
    const int seg0 = seg_start ? seg_start[expert] : 0;
    const int segn = seg_count ? seg_count[expert] : N;
    const int tok0 = blockIdx.y * BN;
    if (tok0 >= segn) return;                                 // (never with ksplit > 1: plain MUL_MAT has no empty tiles)
    // which matrix of the group this row tile belongs to (wave-uniform)
    const uint8_t * Wg = W;  float * dstg = dst;  int64_t rbg = row_bytes, lddg = ldd;  int Mg = M, col0 = 0, tile = blockIdx.x;
#pragma unroll
    for (int k = 0; k < 3; ++k)
        if (k < more.n && (int) blockIdx.x >= more.tile_begin[k]) {
            Wg = more.w[k];  dstg = more.dst[k];  rbg = more.row_bytes[k];  lddg = more.ldd[k];  Mg = more.m[k];  col0 = more.col0[k];
            tile = blockIdx.x - more.tile_begin[k];
        }
    const int mtot = more.n ? more.mtot : M;
    const int row0 = tile * (32 * NW) + wave * 32;
    const int r = lane & 31, h = lane >> 5;
    const uint8_t * wrow = Wg + (int64_t) expert * expert_bytes + (int64_t) min(row0 + r, Mg - 1) * rbg;
    const int nk = Kp / BK;                                   // K % 256 == 0 for Q4_K: nk is a multiple of 4
    const int nblk = K / 256;
    const int bper = (nblk + ksplit - 1) / ksplit;            // this workgroup's Q4_K blocks: [kb0, kb1)
    const int kb0 = split * bper, kb1 = min(kb0 + bper, nblk);

    const uint8_t * xthr = reinterpret_cast<const uint8_t *>(Xh + (int64_t) (seg0 + tok0) * Kp) +
                           (size_t) (tid / SLOTS) * Kp * 2 + (tid % SLOTS) * 16;
    const int xrow_step = (NT / SLOTS) * Kp * 2;

    f32x16 acc[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;

    // loads are unconditional; indices are clamped at the end of K (re-reads valid bytes, results unused)
    auto load_qs  = [&](int ks) { const int k2 = min(ks, nk - 1); return ldg<uint4>(wrow + (size_t) (k2 >> 2) * 144 + 16 + 32 * (k2 & 3) + 16 * h); };
    auto load_hdr = [&](int b)  { return ldg<uint4>(wrow + (size_t) min(b, nblk - 1) * 144); };
    struct XRegs { uint4 c[X_CHUNKS]; };
    auto load_x = [&](int ks) {
        XRegs x;
        const uint8_t * xp = xthr + min(ks, nk - 1) * (BK * 2);
#pragma unroll
        for (int i = 0; i < X_CHUNKS; ++i) x.c[i] = *reinterpret_cast<const uint4 *>(xp + i * xrow_step);
        return x;
    };
    auto store_x = [&](const XRegs x, uint8_t * stage) {
#pragma unroll
        for (int i = 0; i < X_CHUNKS; ++i) {
            const int c = tid + NT * i;
            *reinterpret_cast<uint4 *>(stage + tile_off<BK>(c / SLOTS, c % SLOTS)) = x.c[i];
        }
    };
    struct AFr { uint4 a[NA]; };
    auto read_a = [&](const uint8_t * stage, int slot) {
        AFr f;
#pragma unroll
        for (int i = 0; i < NA; ++i) f.a[i] = *reinterpret_cast<const uint4 *>(stage + tile_off<BK>(32 * i + r, slot));
        return f;
    };
    auto mfma_a = [&](const uint4 b, const AFr f) {
        const f16x8 bb = *reinterpret_cast<const f16x8 *>(&b);
#pragma unroll
        for (int i = 0; i < NA; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8 *>(&f.a[i]), bb, acc[i], 0, 0, 0);
    };
    // the A fragments of k-step kk+1 are read before the MFMAs of k-step kk are issued (two register sets), and the order is
    // pinned: left to itself the scheduler issues the next reads only behind the last MFMA of a group, and the matrix core
    // then idles for an LDS latency per group
    auto compute = [&](const RegbFrag fr, const uint8_t * stage) {
        const AFr a0 = read_a(stage, 0 + h);
        const AFr a1 = read_a(stage, 2 + h);
        mfma_a(fr.f0, a0);
        const AFr a2 = read_a(stage, 4 + h);
        mfma_a(fr.f1, a1);
        const AFr a3 = read_a(stage, 6 + h);
        mfma_a(fr.f2, a2);
        mfma_a(fr.f3, a3);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * NA, 0);   // 2 x NA ds reads
        __builtin_amdgcn_sched_group_barrier(0x008, NA, 0);       // NA MFMAs
        __builtin_amdgcn_sched_group_barrier(0x100, NA, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, NA, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, NA, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * NA, 0);
    };

    // software pipeline: weight bytes two K-steps ahead in registers (even/odd slots), header one block ahead, activation
    // chunks two K-steps ahead in registers and one K-step ahead in LDS, B fragments one K-step ahead: between two barriers a
    // wave unpacks the NEXT K-step's fragments (VALU) and runs this K-step's MFMAs.  The two waves that share a SIMD do these
    // in opposite order (waves < NW/2 unpack first, the others multiply first), so that behind every barrier one of them feeds
    // the VALU while the other feeds the matrix core, instead of both unpacking and then both multiplying.
    uint4 q_e = load_qs(4 * kb0), q_o = load_qs(4 * kb0 + 1);
    uint4 hdr = load_hdr(kb0), hdr_n = load_hdr(kb0 + 1);
    XRegs x_e = load_x(4 * kb0), x_o = load_x(4 * kb0 + 1);
    store_x(x_e, lds);
    x_e = load_x(4 * kb0 + 2);
    RegbFrag fr = regb_unpack_q4k(q_e, hdr, 0);
    q_e = load_qs(4 * kb0 + 2);
    __syncthreads();

    auto run = [&](auto unpack_first) {
        constexpr bool UF = decltype(unpack_first)::value;
        for (int kb = kb0; kb < kb1; ++kb) {                  // one Q4_K block (256 k) = four K-steps, j = 0..3
            const int ks = 4 * kb;
            RegbFrag frn;
            // j = 0: multiply step ks (stage 0), unpack step ks+1 (odd slot)
            if (UF) frn = regb_unpack_q4k(q_o, hdr, 1);
            store_x(x_o, lds + STAGE);  x_o = load_x(ks + 3);
            compute(fr, lds);
            if (!UF) frn = regb_unpack_q4k(q_o, hdr, 1);
            q_o = load_qs(ks + 3);
            __syncthreads();
            fr = frn;
            // j = 1
            if (UF) frn = regb_unpack_q4k(q_e, hdr, 2);
            store_x(x_e, lds);          x_e = load_x(ks + 4);
            compute(fr, lds + STAGE);
            if (!UF) frn = regb_unpack_q4k(q_e, hdr, 2);
            q_e = load_qs(ks + 4);
            __syncthreads();
            fr = frn;
            // j = 2
            if (UF) frn = regb_unpack_q4k(q_o, hdr, 3);
            store_x(x_o, lds + STAGE);  x_o = load_x(ks + 5);
            compute(fr, lds);
            if (!UF) frn = regb_unpack_q4k(q_o, hdr, 3);
            q_o = load_qs(ks + 5);
            __syncthreads();
            fr = frn;
            // j = 3: the next K-step is sub-block 0 of the next block
            if (UF) frn = regb_unpack_q4k(q_e, hdr_n, 0);
            store_x(x_e, lds);          x_e = load_x(ks + 6);
            compute(fr, lds + STAGE);
            if (!UF) frn = regb_unpack_q4k(q_e, hdr_n, 0);
            q_e = load_qs(ks + 6);
            hdr = hdr_n;
            hdr_n = load_hdr(kb + 2);
            __syncthreads();
            fr = frn;
        }
    };
    if (NW >= 8 && wave >= NW / 2) run(std::false_type{});
    else                           run(std::true_type{});

    // epilogue: per-token scales / dst row offsets staged through LDS (all reads of the tiles are behind the last barrier).
    // A split-K range stores its unscaled tile to part[split][token][row] instead (splitk_reduce_kernel finishes the job).
    float *   sc_lds  = reinterpret_cast<float *>(lds);
    int64_t * off_lds = reinterpret_cast<int64_t *>(lds + 1024);
    float *   out     = ksplit > 1 ? part + (int64_t) split * N * mtot + col0 : dstg;
    if (tid < BN) {
        const int t = tok0 + tid;
        const bool live = t < segn;
        sc_lds[tid]  = live ? (ksplit > 1 ? 1.0f : scale[seg0 + t]) : 0.0f;
        off_lds[tid] = live ? (ksplit > 1 ? (int64_t) t * mtot : dst_off ? dst_off[seg0 + t] : (int64_t) t * lddg) : 0;
    }
    __syncthreads();
    const int m = row0 + r;
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int tl = 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (tok0 + tl < segn && m < Mg) out[off_lds[tl] + m] = acc[i][e] * sc_lds[tl];
        }
}

// ---- few tokens (N <= 64): 32 weight rows x 32*NA tokens per workgroup, K split over the workgroup's NWS waves ------------
// With one or two token tiles the kernels above put < 1 workgroup on most CUs and every wave walks the whole of K behind a
// two-K-step prefetch: HBM latency, not bandwidth, sets the time (31 us for a 9 MB matrix).  Here every (32 rows, K/NWS)
// piece is its own wave.  The waves keep the lane -> bytes ownership of Regb<T> and read their A fragments (32 tokens x
// 16 k, L2-resident) straight from the prepared activation matrix, which prep_act_kernel lays out fragment-major for this
// kernel (one coalesced 1 KB load per fragment): no LDS tile and no barrier in the K loop.  What bounds the kernel is the
// unpack VALU (the weights are unpacked for 32 or 64 tokens only) and, on matrices with fewer row groups than CUs, the
// number of CUs pulling on HBM.  The NWS partial tiles meet in LDS; every thread sums its share of the outputs in wave
// order (deterministic) and stores it.
#ifdef QMM_SKINNY_TRACE
__device__ uint64_t * g_trace;
#define QMM_TR(slot) do { if (lane == 0) g_trace[(blockIdx.x * NWS + wave) * 4 + slot] = wall_clock64(); } while (0)
#else
#define QMM_TR(slot)
#endif
template <int T, int NWS, int NA>
__global__ void __launch_bounds__(NWS * 64) __attribute__((amdgpu_waves_per_eu(NA == 1 ? 4 : 2)))
mfma_skinny_kernel(const uint8_t * __restrict__ W, const int64_t row_bytes, const int64_t expert_bytes, const int M, const int K,
                   const uint16_t * __restrict__ Xh, const int Kp, const float * __restrict__ scale,
                   const int * __restrict__ seg_start, const int * __restrict__ seg_count, const int N,
                   float * __restrict__ dst, const int64_t ldd, const int64_t * __restrict__ dst_off) {
    using P = Regb<T>;
    constexpr int BK = P::BK, NFRAG = P::NFRAG, BN = 32 * NA, NT = NWS * 64;
    constexpr int OUTS = NA * 1024 / NT;                       // outputs per thread in the epilogue
    static_assert(NWS * NA * 4096 <= 65536, "partial tiles must fit the static LDS limit");

    __shared__ float red[NWS * NA * 1024];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    QMM_TR(0);
    const int seg0 = seg_start ? seg_start[blockIdx.z] : 0;
    const int segn = seg_count ? seg_count[blockIdx.z] : N;
    // token tiles of one row group are neighbours in launch order (the second tile finds the weights in L2)
    const int lin  = blockIdx.x + gridDim.x * blockIdx.y;
    const int tok0 = (lin % gridDim.y) * BN;
    if (tok0 >= segn) return;
    const int row0 = (lin / gridDim.y) * 32;
    const int r = lane & 31, h = lane >> 5;
    const uint8_t * wrow = W + (int64_t) blockIdx.z * expert_bytes + (int64_t) min(row0 + r, M - 1) * row_bytes;
    const int nk = Kp / BK;
    int per = (nk + NWS - 1) / NWS;                            // K-steps per wave
    if (T == T_Q4_K) per = (per + 3) & ~3;                     // whole Q4_K blocks: compile-time sub-block index
    const int kb = wave * per, ke = min(kb + per, nk);

    // a K-step is SUB substeps of four 16-deep MFMA k-steps (Q6_K: 2, the others 1); A fragments are handled per substep
    constexpr int SUB = NFRAG / 4;
    struct XFr { uint4 a[NA * 4]; };
    struct Frags { uint4 f[4]; };
    // fragment-major operand (frag_major_chunk): the lane's chunk of k-step kk of K-step ks is chunk (ks*NFRAG + kk)*64 of its
    // token tile, i.e. substep s = ks*SUB + sub holds chunks (4s + q)*64, q = 0..3
    const uint4 * xrow[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int t = seg0 + min(tok0 + 32 * i + r, segn - 1);
        xrow[i] = reinterpret_cast<const uint4 *>(Xh) + (int64_t) (t >> 5) * nk * (NFRAG * 64) + h * 32 + (t & 31);
    }
    auto load_x = [&](int s) {
        XFr x;
        const int s2 = min(s, nk * SUB - 1);
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) x.a[i * 4 + q] = xrow[i][(s2 * 4 + q) * 64];
        return x;
    };

    f32x16 acc[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;
    auto mfmas = [&](const Frags fr, const XFr x) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f16x8 bb = *reinterpret_cast<const f16x8 *>(&fr.f[q]);
#pragma unroll
            for (int i = 0; i < NA; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8 *>(&x.a[i * 4 + q]), bb, acc[i], 0, 0, 0);
        }
    };

    // Register rings, statically indexed (the chunk loops are unrolled): weight bytes D K-steps ahead (HBM latency), A
    // fragments one substep ahead (L2 latency).  vmcnt retires in order, so each step issues its fragment loads BEFORE its
    // weight load: the wait for a substep's fragments (issued one substep ago) then never waits for the younger weight
    // loads.  Every wave runs the same number of steps; steps past the wave's range get zero scales.
    XFr xq[2];
    xq[0] = load_x(kb * SUB);
    if constexpr (T == T_Q4_K) {
        const int nblk = K / 256, b0 = kb >> 2, be = ke >> 2;
        auto ld_q = [&](int b, int j) { return ldg<uint4>(wrow + (size_t) min(b, nblk - 1) * 144 + 16 + 32 * j + 16 * h); };
        auto ld_h = [&](int b) { uint4 v = ldg<uint4>(wrow + (size_t) min(b, nblk - 1) * 144); if (b >= be) v.x = 0; return v; };
        uint4 q[4], hdr = ld_h(b0), hdr_n = hdr;
#pragma unroll
        for (int j = 0; j < 4; ++j) q[j] = ld_q(b0, j);
        for (int b = b0; b < b0 + (per >> 2); ++b) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xq[(j + 1) & 1] = load_x(4 * b + j + 1);
                const uint4 cur = q[j];
                q[j] = ld_q(b + 1, j);
                if (j == 0) hdr_n = ld_h(b + 1);
                const RegbFrag rf = regb_unpack_q4k(cur, hdr, j);
                Frags fr;
                fr.f[0] = rf.f0; fr.f[1] = rf.f1; fr.f[2] = rf.f2; fr.f[3] = rf.f3;
                mfmas(fr, xq[j & 1]);
            }
            hdr = hdr_n;
        }
    } else {
        constexpr int D = T == T_Q6_K ? 2 : 4;
        typename P::Raw wq[D];
#pragma unroll
        for (int d = 0; d < D; ++d) wq[d] = P::load(wrow, kb + d < ke ? kb + d : nk, h, K);
        const int chunks = (per + D - 1) / D;
        for (int c = 0; c < chunks; ++c) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int ks = kb + c * D + d;
                const typename P::Raw cur = wq[d];
#pragma unroll
                for (int sub = 0; sub < SUB; ++sub) {
                    const int par = (d * SUB + sub) & 1;
                    xq[par ^ 1] = load_x(ks * SUB + sub + 1);
                    if (sub == 0) wq[d] = P::load(wrow, ks + D < ke ? ks + D : nk, h, K);
                    Frags fr;
                    if constexpr (T == T_Q6_K) { if (sub == 0) P::template unpack_half<0>(cur, h, fr.f); else P::template unpack_half<1>(cur, h, fr.f); }
                    else                       P::unpack(cur, ks, fr.f);
                    mfmas(fr, xq[par]);
                }
            }
        }
    }

    QMM_TR(1);
    // the NWS partial tiles -> LDS; thread `tid` owns outputs tid, tid + NT, ...: output o = (tile i, acc element e, lane l)
    {
        float * o = red + wave * (NA * 1024) + lane;
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[(i * 16 + e) * 64] = acc[i][e];
    }
    float   sc[OUTS];
    int64_t off[OUTS];
    bool    live[OUTS];
#pragma unroll
    for (int j = 0; j < OUTS; ++j) {                          // issued before the barrier: their latency hides behind it
        const int o = tid + j * NT, l = o & 63, e = (o >> 6) & 15, i = o >> 10;
        const int t = tok0 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * (l >> 5);
        live[j] = t < segn && row0 + (l & 31) < M;
        const int tc = seg0 + min(t, segn - 1);
        sc[j]  = scale[tc];
        off[j] = (dst_off ? dst_off[tc] : (int64_t) (tc - seg0) * ldd) + row0 + (l & 31);
    }
    __syncthreads();
    QMM_TR(2);
#pragma unroll
    for (int j = 0; j < OUTS; ++j) {
        const int o = tid + j * NT;
        float sum = red[o];
#pragma unroll
        for (int w = 1; w < NWS; ++w) sum += red[w * (NA * 1024) + o];
        if (live[j]) dst[off[j]] = sum * sc[j];
    }
    QMM_TR(3);
}

inline bool mfma_regb_supports(const qmm_ctx * c, int type) {
    return c->prec == QMM_PREC_F16_Q8 && (type == T_Q4_K || type == T_Q5_K || type == T_Q6_K || type == T_Q4_0 || type == T_Q8_0);
}

template <int T>
inline int launch_mfma_regb_t(qmm_ctx * c, hipStream_t st, const void * W, int64_t rb, int64_t eb, int n_expert, int M, int K,
                              const MfmaOperand & op, const int * seg_start, const int * seg_count, int N, int n_tiles_y,
                              float * dst, int64_t ldd, const int64_t * dst_off, const RegbMore * group = nullptr) {
    RegbMore more;
    memset(&more, 0, sizeof(more));
    if (group) more = *group;                                 // (callers group only where the 256-row tile is the choice)
    const int tiles_x = more.n ? more.tile_begin[more.n - 1] + (more.m[more.n - 1] + 255) / 256 : (M + 255) / 256;
    // tile choice = chip fill: 256 rows x 128 tokens (8 waves, two per SIMD) when that gives (almost) every CU a workgroup,
    // else 128 x 128, else 128 x 64 / 128 x 32 (the weight unpack is then repeated 2x / 4x, on CUs that would otherwise idle).
    // n_tiles_y counts 128-token tiles of the (worst-case) token range.
    const int ksplit = n_expert == 1 ? op.ksplit : 1;
    const int64_t wg_256 = (int64_t) tiles_x * n_tiles_y * n_expert;
    const int64_t wg_128 = (int64_t) ((M + 127) / 128) * n_tiles_y * n_expert;
#define QMM_REGB(NWv, BNv, ROWS, TY)                                                                                                   \
    do {                                                                                                                               \
        auto kern = T == T_Q4_K ? mfma_regb_q4k_kernel<NWv, BNv> : mfma_regb_kernel<T, NWv, BNv>;                                      \
        const size_t lds = (size_t) 2 * BNv * Regb<T>::BK * 2 < 2048 ? 2048 : (size_t) 2 * BNv * Regb<T>::BK * 2;                     \
        hipLaunchKernelGGL(kern, dim3(more.n ? tiles_x : (M + ROWS - 1) / ROWS, TY, n_expert * ksplit), dim3(NWv * 64), lds, st,       \
                           (const uint8_t *) W, rb, eb, M, K, op.xh, op.Kp, op.scale, seg_start, seg_count, N, dst, ldd, dst_off,      \
                           ksplit, op.part, more);                                                                                     \
    } while (0)
#define QMM_SKINNY(NWSv, NAv)                                                                                                          \
    hipLaunchKernelGGL((mfma_skinny_kernel<T, NWSv, NAv>), dim3((M + 31) / 32, (N + 32 * NAv - 1) / (32 * NAv), n_expert),             \
                       dim3(NWSv * 64), 0, st, (const uint8_t *) W, rb, eb, M, K, op.xh, op.Kp, op.scale, seg_start, seg_count, N,     \
                       dst, ldd, dst_off)
    if (op.frag_major && N <= 32) QMM_SKINNY(8, 1);           // (16 waves per group measured no better, Q6_K worse)
    else if (op.frag_major)       QMM_SKINNY(8, 2);
    else if (more.n || ksplit > 1 || wg_256 * 10 >= (int64_t) c->cus * 8) QMM_REGB(8, 128, 256, n_tiles_y);
    else if (wg_128 >= c->cus && T != T_Q6_K) QMM_REGB(4, 128, 128, n_tiles_y);   // (Q6_K: this shape spills, 4.5x slower)
    else if (2 * wg_128 >= c->cus / 2)       QMM_REGB(4, 64, 128, 2 * n_tiles_y);
    else                                     QMM_REGB(4, 32, 128, 4 * n_tiles_y);
#undef QMM_REGB
#undef QMM_SKINNY
    if (ksplit > 1 && !op.frag_major) {
        int vec = M % 4 == 0 && ldd % 4 == 0 && ((uintptr_t) dst & 15) == 0;
        for (int k = 0; k < more.n; ++k) vec = vec && more.m[k] % 4 == 0 && more.ldd[k] % 4 == 0 && ((uintptr_t) more.dst[k] & 15) == 0;
        const int mtot = more.n ? more.mtot : M;
        const int64_t items = ((int64_t) N * mtot + (vec ? 3 : 0)) / (vec ? 4 : 1);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned) ((items + 255) / 256)), dim3(256), 0, st, op.part, ksplit, N, M,
                           op.scale, dst, ldd, vec, more);
    }
    HIP_TRY(hipGetLastError());
    return QMM_OK;
}

inline int launch_mfma_regb(qmm_ctx * c, hipStream_t st, int type, const void * W, int64_t rb, int64_t eb, int n_expert, int M, int K,
                            const MfmaOperand & op, const int * seg_start, const int * seg_count, int N, int n_tiles_y,
                            float * dst, int64_t ldd, const int64_t * dst_off, const RegbMore * group) {
    switch (type) {
        case T_Q4_K: return launch_mfma_regb_t<T_Q4_K>(c, st, W, rb, eb, n_expert, M, K, op, seg_start, seg_count, N, n_tiles_y, dst, ldd, dst_off, group);
        case T_Q5_K: return launch_mfma_regb_t<T_Q5_K>(c, st, W, rb, eb, n_expert, M, K, op, seg_start, seg_count, N, n_tiles_y, dst, ldd, dst_off, group);
        case T_Q8_0: return launch_mfma_regb_t<T_Q8_0>(c, st, W, rb, eb, n_expert, M, K, op, seg_start, seg_count, N, n_tiles_y, dst, ldd, dst_off, group);
        case T_Q6_K: return launch_mfma_regb_t<T_Q6_K>(c, st, W, rb, eb, n_expert, M, K, op, seg_start, seg_count, N, n_tiles_y, dst, ldd, dst_off, group);
        default:     return launch_mfma_regb_t<T_Q4_0>(c, st, W, rb, eb, n_expert, M, K, op, seg_start, seg_count, N, n_tiles_y, dst, ldd, dst_off, group);
    }
}

// Plain MUL_MATs that share src1 and the weight type (attn_q / attn_k / attn_v at prefill): one launch of the tiled kernel
// over the row tiles of all of them and one reduce, instead of a launch (and, with split-K, a reduce) per matrix; the 1024-row
// matrices alone are 16 tiles each.  Falls back to one call per matrix where the few-token kernel or a smaller tile applies.
inline int mfma_mul_mat_group(qmm_ctx * c, hipStream_t st, int type, const qmm_weight * ws, int n, int64_t K, const float * x, int64_t N,
                              int64_t ldx, bool reuse_prep) {
    int64_t mtot = 0, tiles = 0;
    for (int i = 0; i < n; ++i) { mtot += ws[i].M; tiles += (ws[i].M + 255) / 256; }
    tiles *= (N + 127) / 128;
    int ksplit = 1;
    if (c->splitk && tiles * 10 < (int64_t) c->cus * 8) {
        int64_t s = c->cus / tiles;
        if (s > 8) s = 8;
        if (c->splitk > 1 && s > c->splitk) s = c->splitk;
        if (s > K / 512) s = K / 512;
        ksplit = s < 2 ? 1 : (int) s;
    }
    const bool ok = n >= 2 && n <= 4 && c->mm_group && mfma_regb_supports(c, type) && N > c->skinny_max_n_few &&
                    (ksplit > 1 || tiles * 10 >= (int64_t) c->cus * 8);
    if (!ok) {
        for (int i = 0; i < n; ++i) {
            const int rc = mfma_mul_mat(c, st, type, ws[i].w, ws[i].w_row_bytes, K, ws[i].M, x, N, ldx, ws[i].dst, ws[i].ldd, reuse_prep || i > 0);
            if (rc) return rc;
        }
        return QMM_OK;
    }
    const int Kp = mfma_kpad(K), Np = mfma_npad(N);
    const size_t xh_bytes = (size_t) Np * Kp * 2;
    const size_t sc_bytes = ((size_t) Np * 4 + 255) & ~(size_t) 255;
    const size_t need = xh_bytes + sc_bytes + (ksplit > 1 ? (size_t) ksplit * N * mtot * sizeof(float) : 0) + 256;
    int rc = ensure_ws(c, need);
    if (rc) return rc;
    uint16_t * xh = (uint16_t *) c->ws;
    float * scale = (float *) ((uint8_t *) c->ws + xh_bytes);
    if (!reuse_prep) {
        const bool q8_0 = (type == T_Q4_0 || type == T_Q8_0);
        rc = q8_0 ? launch_prep<T_Q8_0>(c, st, type, x, ldx, nullptr, nullptr, (int) N, Np, (int) K, Kp, 0, xh, scale)
                  : launch_prep<T_Q8_K>(c, st, type, x, ldx, nullptr, nullptr, (int) N, Np, (int) K, Kp, 0, xh, scale);
        if (rc) return rc;
    }
    MfmaOperand op = { xh, scale, Kp, 0 };
    op.ksplit = ksplit;
    op.part = reinterpret_cast<float *>((uint8_t *) c->ws + xh_bytes + sc_bytes);
    RegbMore more;
    memset(&more, 0, sizeof(more));
    int tile = (int) ((ws[0].M + 255) / 256), col = (int) ws[0].M;
    for (int i = 1; i < n; ++i) {
        more.w[i - 1] = (const uint8_t *) ws[i].w;  more.dst[i - 1] = ws[i].dst;  more.row_bytes[i - 1] = ws[i].w_row_bytes;
        more.ldd[i - 1] = ws[i].ldd;  more.m[i - 1] = (int) ws[i].M;  more.tile_begin[i - 1] = tile;  more.col0[i - 1] = col;
        tile += (int) ((ws[i].M + 255) / 256);
        col += (int) ws[i].M;
    }
    more.n = n - 1;
    more.mtot = (int) mtot;
    return launch_mfma_regb(c, st, type, ws[0].w, ws[0].w_row_bytes, 0, 1, (int) ws[0].M, (int) K, op, nullptr, nullptr, (int) N, Np / 128,
                            ws[0].dst, ws[0].ldd, nullptr, &more);
}

} // namespace qmm
