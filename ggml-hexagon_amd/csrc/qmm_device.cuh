// qmm_device.cuh — block formats, lane "units" and their decoders for gfx950 (wave64).
//
// A *unit* is the piece of a quantized weight row that ONE lane owns in every kernel of this
// library: it is what the lane fetches from HBM with its own (possibly unaligned) vector loads and
// what it unpacks.  The same load/unpack code feeds
//   - qmm_dequantize   (f32 out, bit-exact spec: ggml/src/ggml-quants.c:255-273, 349-363,
//                       1280-1302, 1482-1508, 1690-1722),
//   - the mat-vec kernel (int8 dot against Q8 activations held in LDS), and
//   - the MFMA kernel   (f32 -> f16/bf16 into the LDS A tile),
// so the bit-exact unpack test covers all three.
//
//   type   block  bytes  unit                                   weights/unit  units/block
//   Q4_0     32     18   the block                                   32           1
//   Q8_0     32     34   the block                                   32           1
//   Q4_K    256    144   16 B of qs (sub-block pair j, half h)       32           8
//   Q5_K    256    176   16 B of qs + 16 B of qh                     32           8
//   Q6_K    256    210   16 B ql[l], 16 B ql[l+32], 16 B qh          64           4
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

namespace qmm {

enum : int { T_Q4_0 = 2, T_Q8_0 = 8, T_Q4_K = 12, T_Q5_K = 13, T_Q6_K = 14, T_Q8_K = 15 };

constexpr int WAVE = 64;

__device__ __forceinline__ float h2f(uint32_t h) { return __half2float(__ushort_as_half((unsigned short)h)); }

// unaligned vector loads: gfx950 under HSA runs with unaligned global access enabled, and hipcc
// lowers these memcpys to global_load_dwordx2/x4 at the pointer's real alignment.
template <typename V> __device__ __forceinline__ V ldg(const uint8_t * p) {
    V v;
    __builtin_memcpy(&v, p, sizeof(V));
    return v;
}
__device__ __forceinline__ uint4 ldg16_aligned(const uint8_t * p) { return *reinterpret_cast<const uint4 *>(p); }

__device__ __forceinline__ int dot4(int a, int b, int c) { return __builtin_amdgcn_sdot4(a, b, c, false); }

// bytes in [0,63] minus a constant c <= 0x3f per byte, as packed int8 (no inter-byte borrow)
__device__ __forceinline__ uint32_t sub4(uint32_t v, uint32_t c4) { return ((v | 0x80808080u) - c4) ^ 0x80808080u; }

// LDS placement of the int8 activations.  A ds_read_b128 is served in groups of 16 lanes over 64 banks
// (MI355X_MICROARCH.md, LDS): lanes whose 16-byte slots are a multiple of 256 B apart collide.  Each
// unit shape reads its activations at a fixed stride across lanes, so the logical 16-byte slot of
// element k is moved to a per-type position that makes every read instruction conflict-free.
// SWZ: 0 = identity (standalone quantizer), otherwise the weight type.
template <int SWZ> __device__ __forceinline__ int act_pos(int k) {
    if (SWZ == 12 || SWZ == 13) {                 // Q4_K / Q5_K: lane = (block b, pair j) reads slots 0..3 of its 64 B
        const int b = k >> 8, s = (k >> 4) & 3;
        return (k & ~0x30) | (((s + b) & 3) << 4);
    } else if (SWZ == 14) {                       // Q6_K: lane = (b, half n, g) reads slots 8n + g + 2r
        const int b = k >> 8, s = (k >> 4) & 15;
        return (k & ~0xf0) | (((s + 2 * (b & 3)) & 15) << 4);
    } else if (SWZ == 2 || SWZ == 8) {            // Q4_0 / Q8_0: lane = block reads its two slots
        return k ^ (((k >> 8) & 1) << 4);
    }
    return k;
}

template <int T> struct Traits;
template <> struct Traits<T_Q4_0> { static constexpr int BLCK = 32,  TSIZE = 18,  UNIT_W = 32, UPB = 1, ACT = T_Q8_0; };
template <> struct Traits<T_Q8_0> { static constexpr int BLCK = 32,  TSIZE = 34,  UNIT_W = 32, UPB = 1, ACT = T_Q8_0; };
template <> struct Traits<T_Q4_K> { static constexpr int BLCK = 256, TSIZE = 144, UNIT_W = 32, UPB = 8, ACT = T_Q8_K; };
template <> struct Traits<T_Q5_K> { static constexpr int BLCK = 256, TSIZE = 176, UNIT_W = 32, UPB = 8, ACT = T_Q8_K; };
template <> struct Traits<T_Q6_K> { static constexpr int BLCK = 256, TSIZE = 210, UNIT_W = 64, UPB = 4, ACT = T_Q8_K; };

// ---------------------------------------------------------------------------------------------
// K-quant packed 6-bit (scale, min) pair `idx` (0..7) out of the 12 bytes held in hdr.y/z/w
// (get_scale_min_k4, ggml-quants.c:631-638)
__device__ __forceinline__ uint32_t k4_byte(const uint4 & hdr, int i) {
    const uint32_t w = i < 4 ? hdr.y : (i < 8 ? hdr.z : hdr.w);
    return (w >> ((i & 3) * 8)) & 0xffu;
}
__device__ __forceinline__ void k4_scale_min(const uint4 & hdr, int idx, int & sc, int & mn) {
    if (idx < 4) {
        sc = (int)(k4_byte(hdr, idx) & 63u);
        mn = (int)(k4_byte(hdr, idx + 4) & 63u);
    } else {
        const uint32_t hi = k4_byte(hdr, idx + 4);
        sc = (int)((hi & 15u) | ((k4_byte(hdr, idx - 4) >> 6) << 4));
        mn = (int)((hi >> 4)  | ((k4_byte(hdr, idx)     >> 6) << 4));
    }
}

// ---------------------------------------------------------------------------------------------
// Units.  Every Unit<T> has
//   load(row, u)                     fetch unit u of the row (row = first byte of the weight row)
//   k_run(u, r) / RUNS / RUN_LEN     where its weights sit along K: RUNS runs of RUN_LEN consecutive k
//   to_f32(u, out[UNIT_W])           bit-exact dequantized values, run-major
//   dot(u, aq, ad, abs, K, nblk)     integer dot against one token's Q8 activations in LDS -> f32 partial

template <int T> struct Unit;

// ---- Q4_0 : f16 d | 16 bytes, low nibbles = weights 0..15, high nibbles = 16..31 ---------------
template <> struct Unit<T_Q4_0> {
    static constexpr int RUNS = 2, RUN_LEN = 16;
    uint4    qs;
    uint32_t d;
    __device__ __forceinline__ void load(const uint8_t * row, int u) {
        const uint8_t * blk = row + (size_t) u * 18;
        d  = ldg<uint16_t>(blk);
        qs = ldg<uint4>(blk + 2);
    }
    static __device__ __forceinline__ int k_run(int u, int r) { return u * 32 + r * 16; }
    __device__ __forceinline__ void kill() { d = 0; }
    __device__ __forceinline__ void to_f32(int, float * out) const {
        const float df = h2f(d);
        const uint32_t w[4] = { qs.x, qs.y, qs.z, qs.w };
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int byte = (w[i] >> (8 * b)) & 0xff;
                out[4 * i + b]      = (float) ((byte & 15) - 8) * df;
                out[16 + 4 * i + b] = (float) ((byte >> 4) - 8) * df;
            }
    }
    // aq: int8 [K] of this token, ad: f32 [K/32]
    __device__ __forceinline__ float dot(int u, const int8_t * aq, const float * ad, const int16_t *) const {
        return dot_at(aq + u * 32, aq + u * 32 + 16, ad[u]);
    }
    __device__ __forceinline__ float dot_at(const int8_t * p0, const int8_t * p1, float dy) const {
        const int4 a0 = *reinterpret_cast<const int4 *>(p0);
        const int4 a1 = *reinterpret_cast<const int4 *>(p1);
        const uint32_t m = 0x0f0f0f0fu, c8 = 0x08080808u;
        int s = 0;
        s = dot4((int) sub4(qs.x & m, c8), a0.x, s); s = dot4((int) sub4((qs.x >> 4) & m, c8), a1.x, s);
        s = dot4((int) sub4(qs.y & m, c8), a0.y, s); s = dot4((int) sub4((qs.y >> 4) & m, c8), a1.y, s);
        s = dot4((int) sub4(qs.z & m, c8), a0.z, s); s = dot4((int) sub4((qs.z >> 4) & m, c8), a1.z, s);
        s = dot4((int) sub4(qs.w & m, c8), a0.w, s); s = dot4((int) sub4((qs.w >> 4) & m, c8), a1.w, s);
        return (float) s * h2f(d) * dy;               // sumi*dx*dy, ggml-cpu-quants.c:2604-2605
    }
};

// ---- Q8_0 : f16 d | 32 int8 ---------------------------------------------------------------------
template <> struct Unit<T_Q8_0> {
    static constexpr int RUNS = 1, RUN_LEN = 32;
    uint4    q0, q1;
    uint32_t d;
    __device__ __forceinline__ void load(const uint8_t * row, int u) {
        const uint8_t * blk = row + (size_t) u * 34;
        d  = ldg<uint16_t>(blk);
        q0 = ldg<uint4>(blk + 2);
        q1 = ldg<uint4>(blk + 18);
    }
    static __device__ __forceinline__ int k_run(int u, int) { return u * 32; }
    __device__ __forceinline__ void kill() { d = 0; }
    __device__ __forceinline__ void to_f32(int, float * out) const {
        const float df = h2f(d);
        const uint32_t w[8] = { q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w };
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int b = 0; b < 4; ++b) out[4 * i + b] = (float) (int8_t) ((w[i] >> (8 * b)) & 0xff) * df;
    }
    __device__ __forceinline__ float dot(int u, const int8_t * aq, const float * ad, const int16_t *) const {
        return dot_at(aq + u * 32, aq + u * 32 + 16, ad[u]);
    }
    __device__ __forceinline__ float dot_at(const int8_t * p0, const int8_t * p1, float dy) const {
        const int4 a0 = *reinterpret_cast<const int4 *>(p0);
        const int4 a1 = *reinterpret_cast<const int4 *>(p1);
        int s = 0;
        s = dot4((int) q0.x, a0.x, s); s = dot4((int) q0.y, a0.y, s); s = dot4((int) q0.z, a0.z, s); s = dot4((int) q0.w, a0.w, s);
        s = dot4((int) q1.x, a1.x, s); s = dot4((int) q1.y, a1.y, s); s = dot4((int) q1.z, a1.z, s); s = dot4((int) q1.w, a1.w, s);
        return (float) s * (h2f(d) * dy);             // sumi*(dx*dy), ggml-cpu-quants.c:4011
    }
};

// ---- Q4_K : f16 d, f16 dmin, 12 B scales | 128 B qs --------------------------------------------
// unit u: block b = u/8, q = u%8, pair j = q/2, half h = q%2.  Its 16 qs bytes hold, in the low
// nibbles, weights 64j+16h .. +16 (sub-block 2j) and in the high nibbles 64j+32+16h .. +16 (2j+1).
template <> struct Unit<T_Q4_K> {
    static constexpr int RUNS = 2, RUN_LEN = 16;
    uint4 qs, hdr;
    __device__ __forceinline__ void load(const uint8_t * row, int u) {
        const uint8_t * blk = row + (size_t) (u >> 3) * 144;
        hdr = ldg<uint4>(blk);                         // shared by the block's 8 lanes: one L1 line
        qs  = ldg<uint4>(blk + 16 + 16 * (u & 7));
    }
    static __device__ __forceinline__ int k_run(int u, int r) {
        const int q = u & 7;
        return (u >> 3) * 256 + 64 * (q >> 1) + 16 * (q & 1) + 32 * r;
    }
    __device__ __forceinline__ void kill() { hdr.x = 0; }          // d = dmin = 0
    __device__ __forceinline__ void to_f32(int u, float * out) const {
        const int j = (u & 7) >> 1;
        int s0, m0, s1, m1;
        k4_scale_min(hdr, 2 * j, s0, m0);
        k4_scale_min(hdr, 2 * j + 1, s1, m1);
        const float d = h2f(hdr.x & 0xffff), dmin = h2f(hdr.x >> 16);
        const float d0 = d * (float) s0, o0 = dmin * (float) m0, d1 = d * (float) s1, o1 = dmin * (float) m1;
        const uint32_t w[4] = { qs.x, qs.y, qs.z, qs.w };
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int byte = (w[i] >> (8 * b)) & 0xff;
                out[4 * i + b]      = d0 * (float) (byte & 15) - o0;
                out[16 + 4 * i + b] = d1 * (float) (byte >> 4) - o1;
            }
    }
    // aq int8 [K]; ad f32 [K/256]; bsum int16 [K/16]
    __device__ __forceinline__ float dot(int u, const int8_t * aq, const float * ad, const int16_t * bsum) const {
        const int b = u >> 3, q = u & 7, j = q >> 1, h = q & 1;
        const int k0 = b * 256 + 64 * j + 16 * h;
        const int4 a0 = *reinterpret_cast<const int4 *>(aq + k0);
        const int4 a1 = *reinterpret_cast<const int4 *>(aq + k0 + 32);
        const uint32_t m = 0x0f0f0f0fu;
        int lo = 0, hi = 0;
        lo = dot4((int) (qs.x & m), a0.x, lo); hi = dot4((int) ((qs.x >> 4) & m), a1.x, hi);
        lo = dot4((int) (qs.y & m), a0.y, lo); hi = dot4((int) ((qs.y >> 4) & m), a1.y, hi);
        lo = dot4((int) (qs.z & m), a0.z, lo); hi = dot4((int) ((qs.z >> 4) & m), a1.z, hi);
        lo = dot4((int) (qs.w & m), a0.w, lo); hi = dot4((int) ((qs.w >> 4) & m), a1.w, hi);
        int s0, m0, s1, m1;
        k4_scale_min(hdr, 2 * j, s0, m0);
        k4_scale_min(hdr, 2 * j + 1, s1, m1);
        const int isum = s0 * lo + s1 * hi;                                            // aux32, exact
        const int msum = m0 * (int) bsum[b * 16 + 4 * j + h] + m1 * (int) bsum[b * 16 + 4 * j + 2 + h];
        const float yd = ad[b];
        return (h2f(hdr.x & 0xffff) * yd) * (float) isum - (h2f(hdr.x >> 16) * yd) * (float) msum;
    }
};

// ---- Q5_K : header as Q4_K | 32 B qh | 128 B qs -------------------------------------------------
template <> struct Unit<T_Q5_K> {
    static constexpr int RUNS = 2, RUN_LEN = 16;
    uint4 qs, qh, hdr;
    __device__ __forceinline__ void load(const uint8_t * row, int u) {
        const uint8_t * blk = row + (size_t) (u >> 3) * 176;
        hdr = ldg<uint4>(blk);
        qh  = ldg<uint4>(blk + 16 + 16 * (u & 1));     // qh[l], l = 16h .. 16h+15
        qs  = ldg<uint4>(blk + 48 + 16 * (u & 7));
    }
    static __device__ __forceinline__ int k_run(int u, int r) { return Unit<T_Q4_K>::k_run(u, r); }
    __device__ __forceinline__ void kill() { hdr.x = 0; }
    // 5-bit values of dword i: low-nibble weights and high-nibble weights
    __device__ __forceinline__ void q5(int i, int j, uint32_t & lo, uint32_t & hi) const {
        const uint32_t w = i == 0 ? qs.x : i == 1 ? qs.y : i == 2 ? qs.z : qs.w;
        const uint32_t g = i == 0 ? qh.x : i == 1 ? qh.y : i == 2 ? qh.z : qh.w;
        lo = (w & 0x0f0f0f0fu)        | (((g >> (2 * j))     & 0x01010101u) << 4);
        hi = ((w >> 4) & 0x0f0f0f0fu) | (((g >> (2 * j + 1)) & 0x01010101u) << 4);
    }
    __device__ __forceinline__ void to_f32(int u, float * out) const {
        const int j = (u & 7) >> 1;
        int s0, m0, s1, m1;
        k4_scale_min(hdr, 2 * j, s0, m0);
        k4_scale_min(hdr, 2 * j + 1, s1, m1);
        const float d = h2f(hdr.x & 0xffff), dmin = h2f(hdr.x >> 16);
        const float d0 = d * (float) s0, o0 = dmin * (float) m0, d1 = d * (float) s1, o1 = dmin * (float) m1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t lo, hi;
            q5(i, j, lo, hi);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                out[4 * i + b]      = d0 * (float) ((lo >> (8 * b)) & 0xff) - o0;
                out[16 + 4 * i + b] = d1 * (float) ((hi >> (8 * b)) & 0xff) - o1;
            }
        }
    }
    __device__ __forceinline__ float dot(int u, const int8_t * aq, const float * ad, const int16_t * bsum) const {
        const int b = u >> 3, q = u & 7, j = q >> 1, h = q & 1;
        const int k0 = b * 256 + 64 * j + 16 * h;
        const int4 a0 = *reinterpret_cast<const int4 *>(aq + k0);
        const int4 a1 = *reinterpret_cast<const int4 *>(aq + k0 + 32);
        const int av0[4] = { a0.x, a0.y, a0.z, a0.w }, av1[4] = { a1.x, a1.y, a1.z, a1.w };
        int lo = 0, hi = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t l5, h5;
            q5(i, j, l5, h5);
            lo = dot4((int) l5, av0[i], lo);
            hi = dot4((int) h5, av1[i], hi);
        }
        int s0, m0, s1, m1;
        k4_scale_min(hdr, 2 * j, s0, m0);
        k4_scale_min(hdr, 2 * j + 1, s1, m1);
        const int isum = s0 * lo + s1 * hi;
        const int msum = m0 * (int) bsum[b * 16 + 4 * j + h] + m1 * (int) bsum[b * 16 + 4 * j + 2 + h];
        const float yd = ad[b];
        return (h2f(hdr.x & 0xffff) * yd) * (float) isum - (h2f(hdr.x >> 16) * yd) * (float) msum;
    }
};

// ---- Q6_K : 128 B ql | 64 B qh | 16 int8 scales | f16 d -----------------------------------------
// unit u: block b = u/4, half n = (u/2)%2, g = u%2 -> l = 16g .. 16g+15 of that half.  It owns
// ql[64n+l], ql[64n+32+l], qh[32n+l] and therefore weights 128n + l + {0,32,64,96}: 4 runs of 16.
template <> struct Unit<T_Q6_K> {
    static constexpr int RUNS = 4, RUN_LEN = 16;
    uint4    qa, qb, qh;
    uint2    sc8;       // the half's 8 scale bytes
    uint32_t d;
    __device__ __forceinline__ void load(const uint8_t * row, int u) {
        const uint8_t * blk = row + (size_t) (u >> 2) * 210;
        const int n = (u >> 1) & 1, g = u & 1;
        qa  = ldg<uint4>(blk + 64 * n + 16 * g);
        qb  = ldg<uint4>(blk + 64 * n + 32 + 16 * g);
        qh  = ldg<uint4>(blk + 128 + 32 * n + 16 * g);
        sc8 = ldg<uint2>(blk + 192 + 8 * n);
        d   = ldg<uint16_t>(blk + 208);
    }
    static __device__ __forceinline__ int k_run(int u, int r) {
        return (u >> 2) * 256 + 128 * ((u >> 1) & 1) + 16 * (u & 1) + 32 * r;
    }
    __device__ __forceinline__ void kill() { d = 0; }
    __device__ __forceinline__ int scale(int g, int r) const {      // sc[8n + g + 2r]
        const int i = g + 2 * r;
        const uint32_t w = i < 4 ? sc8.x : sc8.y;
        return (int) (int8_t) ((w >> ((i & 3) * 8)) & 0xff);
    }
    // packed 6-bit values (still biased by 32) of dword i for the four runs
    __device__ __forceinline__ void q6(int i, uint32_t (&v)[4]) const {
        const uint32_t a = i == 0 ? qa.x : i == 1 ? qa.y : i == 2 ? qa.z : qa.w;
        const uint32_t b = i == 0 ? qb.x : i == 1 ? qb.y : i == 2 ? qb.z : qb.w;
        const uint32_t h = i == 0 ? qh.x : i == 1 ? qh.y : i == 2 ? qh.z : qh.w;
        const uint32_t m4 = 0x0f0f0f0fu, m2 = 0x03030303u;
        v[0] = (a & m4)        | ((h & m2) << 4);
        v[1] = (b & m4)        | (((h >> 2) & m2) << 4);
        v[2] = ((a >> 4) & m4) | (((h >> 4) & m2) << 4);
        v[3] = ((b >> 4) & m4) | (((h >> 6) & m2) << 4);
    }
    __device__ __forceinline__ void to_f32(int u, float * out) const {
        const int g = u & 1;
        const float df = h2f(d);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t v[4];
            q6(i, v);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float ds = df * (float) scale(g, r);           // d*sc first, as the reference
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    out[16 * r + 4 * i + b] = ds * (float) ((int) ((v[r] >> (8 * b)) & 0xff) - 32);
            }
        }
    }
    __device__ __forceinline__ float dot(int u, const int8_t * aq, const float * ad, const int16_t *) const {
        const int b = u >> 2, g = u & 1;
        const int k0 = k_run(u, 0);
        int acc[4] = { 0, 0, 0, 0 };
        int4 a[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = *reinterpret_cast<const int4 *>(aq + k0 + 32 * r);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t v[4];
            q6(i, v);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int av = i == 0 ? a[r].x : i == 1 ? a[r].y : i == 2 ? a[r].z : a[r].w;
                acc[r] = dot4((int) sub4(v[r], 0x20202020u), av, acc[r]);
            }
        }
        int isum = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) isum += scale(g, r) * acc[r];
        return (h2f(d) * ad[b]) * (float) isum;
    }
};

// ---------------------------------------------------------------------------------------------
// wave-wide reductions on the DPP path (no LDS permute traffic): quad swaps, half-row mirror, row mirror
// leave every lane of a 16-lane row with the row total; the four row totals are combined through SGPRs.
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL> __device__ __forceinline__ int dpp_mov_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }

constexpr int DPP_QUAD_X1 = 0xB1, DPP_QUAD_X2 = 0x4E, DPP_ROW_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140;

__device__ __forceinline__ float readlane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

__device__ __forceinline__ float wave_sum(float v) {          // result in every lane (wave-uniform)
    v += dpp_mov<DPP_QUAD_X1>(v);
    v += dpp_mov<DPP_QUAD_X2>(v);
    v += dpp_mov<DPP_ROW_HALF_MIRROR>(v);
    v += dpp_mov<DPP_ROW_MIRROR>(v);
    return (readlane_f(v, 0) + readlane_f(v, 16)) + (readlane_f(v, 32) + readlane_f(v, 48));
}
// max of non-negative floats through their bit patterns (monotonic as unsigned ints): no NaN canonicalisation ops
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    v = max(v, (uint32_t) dpp_mov_i<DPP_QUAD_X1>((int) v));
    v = max(v, (uint32_t) dpp_mov_i<DPP_QUAD_X2>((int) v));
    v = max(v, (uint32_t) dpp_mov_i<DPP_ROW_HALF_MIRROR>((int) v));
    v = max(v, (uint32_t) dpp_mov_i<DPP_ROW_MIRROR>((int) v));
    const uint32_t a = (uint32_t) __builtin_amdgcn_readlane((int) v, 0), b = (uint32_t) __builtin_amdgcn_readlane((int) v, 16);
    const uint32_t c = (uint32_t) __builtin_amdgcn_readlane((int) v, 32), d = (uint32_t) __builtin_amdgcn_readlane((int) v, 48);
    return max(max(a, b), max(c, d));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_mov<DPP_QUAD_X1>(v));
    v = fmaxf(v, dpp_mov<DPP_QUAD_X2>(v));
    v = fmaxf(v, dpp_mov<DPP_ROW_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_mov<DPP_ROW_MIRROR>(v));
    return fmaxf(fmaxf(readlane_f(v, 0), readlane_f(v, 16)), fmaxf(readlane_f(v, 32), readlane_f(v, 48)));
}

} // namespace qmm
