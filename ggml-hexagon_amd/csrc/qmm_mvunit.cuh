// qmm_mvunit.cuh — the mat-vec kernel's lane units: wider than the generic Unit<T> of qmm_device.cuh and
// tuned for VALU count, because at batch 1 the v_dot4 pipeline has barely more headroom than HBM.
//
//   type   mat-vec unit                                   weights/unit   loads per unit
//   Q4_K   sub-block pair j of a block: 32 B of qs        64             2 x 16 B + header 16 B (4 lanes share it)
//   Q5_K   the same + all 32 qh bytes                     64             4 x 16 B + header
//   Q6_K   Unit<T_Q6_K> (16 l-values x 4 quarters)        64             3 x 16 B + 8 B scales + d
//   Q4_0 / Q8_0  Unit<T> (one 32-weight block)            32
//
// Activation-side block sums: Q4_K/Q5_K need the sum of the 32 activations under each sub-block
// (their `mins` term), Q6_K the sum under each 16 (to fold the -32 bias of its 6-bit values into the
// integer dot: dot(q-32, a) = dot(q, a) - 32*sum(a)).  BSG is the group size the kernel stages.
#pragma once

#include "qmm_device.cuh"

namespace qmm {

template <int T> struct MvUnit : Unit<T> {             // Q4_0 / Q8_0: the generic block unit
    static constexpr int W = Traits<T>::UNIT_W;
    static constexpr int BSG = 16;          // unused
    __device__ __forceinline__ float dot(int u, const int8_t * aq, const float * ad, const int16_t * ab) const {
        // the block's two 16-byte slots are swapped in LDS for odd groups of 8 blocks (act_pos<2|8>)
        return Unit<T>::dot_at(aq + act_pos<T>(u * 32), aq + act_pos<T>(u * 32 + 16), ad[u]);
    }
};

__device__ __forceinline__ int mul24(int a, int b) { return __mul24(a, b); }

// 6-bit (scale, min) pairs 2j and 2j+1 of a K-quant header, packed as sc = s0 | s1 << 8, mn likewise
__device__ __forceinline__ void k4_pair(const uint4 & hdr, int j, uint32_t & sc, uint32_t & mn) {
    const int sh = (j & 1) * 16;
    const uint32_t p0 = (hdr.y >> sh) & 0xffffu, p1 = (hdr.z >> sh) & 0xffffu, p2 = (hdr.w >> sh) & 0xffffu;
    const uint32_t sc_lo = p0 & 0x3f3fu, mn_lo = p1 & 0x3f3fu;
    const uint32_t sc_hi = (p2 & 0x0f0fu) | ((p0 >> 2) & 0x3030u);
    const uint32_t mn_hi = ((p2 >> 4) & 0x0f0fu) | ((p1 >> 2) & 0x3030u);
    sc = j < 2 ? sc_lo : sc_hi;
    mn = j < 2 ? mn_lo : mn_hi;
}

template <> struct MvUnit<T_Q4_K> {
    static constexpr int W = 64, BSG = 32;
    uint4 q0, q1, hdr;
    __device__ __forceinline__ void load(const uint8_t * row, int u) {
        const uint8_t * blk = row + (size_t) (u >> 2) * 144;
        hdr = ldg<uint4>(blk);
        q0  = ldg<uint4>(blk + 16 + 32 * (u & 3));
        q1  = ldg<uint4>(blk + 32 + 32 * (u & 3));
    }
    // aq int8 [K]; ad f32 [K/256]; ab int16 [K/32] (sub-block sums; BS16: [K/16], sums of 16, as the mixed-type kernel stages them)
    template <bool BS16 = false>
    __device__ __forceinline__ float dot(int u, const int8_t * aq, const float * ad, const int16_t * ab) const {
        const int b = u >> 2, j = u & 3;
        const int8_t * ap = aq + b * 256 + 64 * j;           // logical slot s sits at ((s + b) & 3): act_pos<12>
        const int4 a0 = *reinterpret_cast<const int4 *>(ap + (((0 + b) & 3) << 4));
        const int4 a1 = *reinterpret_cast<const int4 *>(ap + (((1 + b) & 3) << 4));
        const int4 a2 = *reinterpret_cast<const int4 *>(ap + (((2 + b) & 3) << 4));
        const int4 a3 = *reinterpret_cast<const int4 *>(ap + (((3 + b) & 3) << 4));
        const uint32_t m = 0x0f0f0f0fu;
        int lo = 0, hi = 0;
        lo = dot4((int) (q0.x & m), a0.x, lo); hi = dot4((int) ((q0.x >> 4) & m), a2.x, hi);
        lo = dot4((int) (q0.y & m), a0.y, lo); hi = dot4((int) ((q0.y >> 4) & m), a2.y, hi);
        lo = dot4((int) (q0.z & m), a0.z, lo); hi = dot4((int) ((q0.z >> 4) & m), a2.z, hi);
        lo = dot4((int) (q0.w & m), a0.w, lo); hi = dot4((int) ((q0.w >> 4) & m), a2.w, hi);
        lo = dot4((int) (q1.x & m), a1.x, lo); hi = dot4((int) ((q1.x >> 4) & m), a3.x, hi);
        lo = dot4((int) (q1.y & m), a1.y, lo); hi = dot4((int) ((q1.y >> 4) & m), a3.y, hi);
        lo = dot4((int) (q1.z & m), a1.z, lo); hi = dot4((int) ((q1.z >> 4) & m), a3.z, hi);
        lo = dot4((int) (q1.w & m), a1.w, lo); hi = dot4((int) ((q1.w >> 4) & m), a3.w, hi);
        uint32_t sc, mn;
        k4_pair(hdr, j, sc, mn);
        int bs_lo, bs_hi;                                    // sums of the activations under the low / high nibble sub-block
        if (BS16) {
            const uint2 b4 = *reinterpret_cast<const uint2 *>(ab + b * 16 + 4 * j);
            bs_lo = (int) (int16_t) (b4.x & 0xffff) + ((int) b4.x >> 16);
            bs_hi = (int) (int16_t) (b4.y & 0xffff) + ((int) b4.y >> 16);
        } else {
            const uint32_t bs = *reinterpret_cast<const uint32_t *>(ab + b * 8 + 2 * j);
            bs_lo = (int) (int16_t) (bs & 0xffff);
            bs_hi = (int) bs >> 16;
        }
        const int isum = mul24((int) (sc & 0xff), lo) + mul24((int) (sc >> 8), hi);
        const int msum = mul24((int) (mn & 0xff), bs_lo) + mul24((int) (mn >> 8), bs_hi);
        const float yd = ad[b];
        return (h2f(hdr.x & 0xffff) * yd) * (float) isum - (h2f(hdr.x >> 16) * yd) * (float) msum;
    }
};

template <> struct MvUnit<T_Q5_K> {
    static constexpr int W = 64, BSG = 32;
    uint4 q0, q1, h0, h1, hdr;
    __device__ __forceinline__ void load(const uint8_t * row, int u) {
        const uint8_t * blk = row + (size_t) (u >> 2) * 176;
        hdr = ldg<uint4>(blk);
        h0  = ldg<uint4>(blk + 16);
        h1  = ldg<uint4>(blk + 32);
        q0  = ldg<uint4>(blk + 48 + 32 * (u & 3));
        q1  = ldg<uint4>(blk + 64 + 32 * (u & 3));
    }
    template <bool BS16 = false>
    __device__ __forceinline__ float dot(int u, const int8_t * aq, const float * ad, const int16_t * ab) const {
        const int b = u >> 2, j = u & 3;
        const int8_t * ap = aq + b * 256 + 64 * j;           // logical slot s sits at ((s + b) & 3): act_pos<12>
        const int4 a0 = *reinterpret_cast<const int4 *>(ap + (((0 + b) & 3) << 4));
        const int4 a1 = *reinterpret_cast<const int4 *>(ap + (((1 + b) & 3) << 4));
        const int4 a2 = *reinterpret_cast<const int4 *>(ap + (((2 + b) & 3) << 4));
        const int4 a3 = *reinterpret_cast<const int4 *>(ap + (((3 + b) & 3) << 4));
        const uint32_t m = 0x0f0f0f0fu, one = 0x01010101u;
        const int sl = 2 * j, sh = 2 * j + 1;
        int lo = 0, hi = 0;
#define QMM_Q5(qw, hw, al, ah)                                                                 \
        lo = dot4((int) ((qw & m) | (((hw >> sl) & one) << 4)), al, lo);                        \
        hi = dot4((int) (((qw >> 4) & m) | (((hw >> sh) & one) << 4)), ah, hi);
        QMM_Q5(q0.x, h0.x, a0.x, a2.x) QMM_Q5(q0.y, h0.y, a0.y, a2.y) QMM_Q5(q0.z, h0.z, a0.z, a2.z) QMM_Q5(q0.w, h0.w, a0.w, a2.w)
        QMM_Q5(q1.x, h1.x, a1.x, a3.x) QMM_Q5(q1.y, h1.y, a1.y, a3.y) QMM_Q5(q1.z, h1.z, a1.z, a3.z) QMM_Q5(q1.w, h1.w, a1.w, a3.w)
#undef QMM_Q5
        uint32_t sc, mn;
        k4_pair(hdr, j, sc, mn);
        int bs_lo, bs_hi;                                    // sums of the activations under the low / high nibble sub-block
        if (BS16) {
            const uint2 b4 = *reinterpret_cast<const uint2 *>(ab + b * 16 + 4 * j);
            bs_lo = (int) (int16_t) (b4.x & 0xffff) + ((int) b4.x >> 16);
            bs_hi = (int) (int16_t) (b4.y & 0xffff) + ((int) b4.y >> 16);
        } else {
            const uint32_t bs = *reinterpret_cast<const uint32_t *>(ab + b * 8 + 2 * j);
            bs_lo = (int) (int16_t) (bs & 0xffff);
            bs_hi = (int) bs >> 16;
        }
        const int isum = mul24((int) (sc & 0xff), lo) + mul24((int) (sc >> 8), hi);
        const int msum = mul24((int) (mn & 0xff), bs_lo) + mul24((int) (mn >> 8), bs_hi);
        const float yd = ad[b];
        return (h2f(hdr.x & 0xffff) * yd) * (float) isum - (h2f(hdr.x >> 16) * yd) * (float) msum;
    }
};

template <> struct MvUnit<T_Q6_K> : Unit<T_Q6_K> {
    static constexpr int W = 64, BSG = 16;
    // ab int16 [K/16]; SWZ: the LDS order the activations were staged in (the mixed-type kernel stages Q4_K's)
    template <int SWZ = T_Q6_K>
    __device__ __forceinline__ float dot(int u, const int8_t * aq, const float * ad, const int16_t * ab) const {
        const int b = u >> 2, n = (u >> 1) & 1, g = u & 1;
        const int k0 = b * 256 + 128 * n + 16 * g;
        int acc[4] = { 0, 0, 0, 0 };
        int4 a[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = *reinterpret_cast<const int4 *>(aq + act_pos<SWZ>(k0 + 32 * r));
        const uint32_t m4 = 0x0f0f0f0fu, m2 = 0x30303030u;
#define QMM_Q6(i, A, B, H, c)                                                              \
        acc[0] = dot4((int) ((A & m4) | ((H << 4) & m2)), a[0].c, acc[0]);                  \
        acc[1] = dot4((int) ((B & m4) | ((H << 2) & m2)), a[1].c, acc[1]);                  \
        acc[2] = dot4((int) (((A >> 4) & m4) | (H & m2)), a[2].c, acc[2]);                  \
        acc[3] = dot4((int) (((B >> 4) & m4) | ((H >> 2) & m2)), a[3].c, acc[3]);
        QMM_Q6(0, qa.x, qb.x, qh.x, x) QMM_Q6(1, qa.y, qb.y, qh.y, y) QMM_Q6(2, qa.z, qb.z, qh.z, z) QMM_Q6(3, qa.w, qb.w, qh.w, w)
#undef QMM_Q6
        // 6-bit values are stored +32: dot(q - 32, a) = dot(q, a) - 32 * sum(a over the 16)
        const int16_t * bsp = ab + b * 16 + 8 * n + g;
        int isum = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) isum += mul24(scale(g, r), acc[r] - 32 * (int) bsp[2 * r]);
        return (h2f(d) * ad[b]) * (float) isum;
    }
};

} // namespace qmm
