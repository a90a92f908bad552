// nb2d + two remedies for what nb2d found (global loads cost the 64-rows-per-wave loop 40 %): FEAT bit3 = weight bytes in
// fragment-major order (a wave's 16 bytes per lane are ONE contiguous KB per (32 rows, K-step) instead of 64 row-strided
// sectors), bit4 = the activation tile straight into LDS (global_load_lds_dwordx4: no VGPR round trip, no ds_write), the lane's
// global address chosen so that the linear LDS destination is the swizzled tile.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ int tile_off(int r, int s) { return r * 128 + ((s ^ ((r >> 1) & 7)) << 4); }
__device__ __forceinline__ uint32_t h2b(f16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ f16x2 b2h(uint32_t v) { return __builtin_bit_cast(f16x2, v); }
__device__ __forceinline__ uint32_t vc(uint32_t v) { asm("" : "+v"(v)); return v; }
struct Sc { f16x2 ds0, no0, ds1, no1; };
__device__ __forceinline__ Sc scales(const uint4 hdr, int j) {
    const uint32_t b0 = (hdr.y >> (8 * (j & 3))) & 63, b1 = (hdr.z >> (8 * (j & 3))) & 63, b2 = (hdr.w >> (8 * (j & 3))) & 63, b3 = (hdr.y >> (8 * ((j + 1) & 3))) & 63;
    const float d = (float) __builtin_bit_cast(_Float16, (unsigned short) (hdr.x & 0xffff)), dm = (float) __builtin_bit_cast(_Float16, (unsigned short) (hdr.x >> 16));
    const _Float16 a = (_Float16) (d * (float) b0), b = (_Float16) (-(dm * (float) b1)), c = (_Float16) (d * (float) b2), e = (_Float16) (-(dm * (float) b3));
    return Sc{ { a, a }, { b, b }, { c, c }, { e, e } };
}
template <int KK> __device__ __forceinline__ uint4 unpack_frag(const uint4 raw, const Sc sc) {
    const f16x2 BL = { (_Float16) -1024.f, (_Float16) -1024.f }, BH = { (_Float16) -64.f, (_Float16) -64.f };
    const uint32_t w0 = (KK & 1) ? raw.z : raw.x, w1 = (KK & 1) ? raw.w : raw.y;
    constexpr bool HI = KK >= 2;
    const uint32_t M = HI ? 0x00f000f0u : 0x000f000fu, E = vc(HI ? 0x54005400u : 0x64006400u);
    const f16x2 B = HI ? BH : BL, DS = HI ? sc.ds1 : sc.ds0, NO = HI ? sc.no1 : sc.no0;
    uint4 f;
    f.x = h2b(__builtin_elementwise_fma(b2h((w0 & M) | E) + B, DS, NO));
    f.y = h2b(__builtin_elementwise_fma(b2h(((w0 >> 8) & M) | E) + B, DS, NO));
    f.z = h2b(__builtin_elementwise_fma(b2h((w1 & M) | E) + B, DS, NO));
    f.w = h2b(__builtin_elementwise_fma(b2h(((w1 >> 8) & M) | E) + B, DS, NO));
    return f;
}

template <int FEAT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
k(float * out, const uint8_t * W, const uint8_t * X, int nsteps) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[32768];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    for (int i = tid; i < 32768 / 4; i += blockDim.x) ((uint32_t *) lds)[i] = 0x3c003c00u + (i & 7);
    __syncthreads();
    f32x16 acc[2][4];
    for (int j = 0; j < 2; ++j) for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;
    const int64_t rb = 2304;                                   // K = 4096 Q4_K row
    const uint8_t * w0 = W + ((int64_t) (blockIdx.x % 112) * 256 + wave * 64 + r) * rb + 16 * h, * w1 = w0 + 32 * rb;
    const uint8_t * wh0 = w0 - 16 * h, * wh1 = w1 - 16 * h;
    const uint8_t * xp = X + (size_t) (tid / 8) * 8192 + (tid % 8) * 16;
    // LDS DMA: wave `wave`, load i fills the KB of token rows (4 i + wave) * 8 .. + 7; lane l lands at 16 l of it = (row l / 8, physical slot l % 8)
    const int xrow = wave * 8 + (lane >> 3);
    const uint8_t * xd = X + (size_t) xrow * 8192 + (((lane & 7) ^ ((xrow >> 1) & 7)) << 4);
    auto ldq = [&](int ks) { const size_t o = (FEAT & 8) ? (size_t) ks * 1024 : (size_t) (ks >> 2) * 144 + 16 + 32 * (ks & 3); return o; };
    if (FEAT & 8) {   // (row tile of 32, K-step) -> 1 KB, lane-major; a row tile's K-steps are consecutive (64 KB per tile at K = 4096)
        w0 = W + ((int64_t) (blockIdx.x % 112) * 8 + wave * 2) * 73728 + lane * 16;
        w1 = w0 + 73728;
    }
    uint4 qe0 = *(const uint4 *) (w0 + ldq(0)), qe1 = *(const uint4 *) (w1 + ldq(0)), qo0 = *(const uint4 *) (w0 + ldq(1)), qo1 = *(const uint4 *) (w1 + ldq(1));
    uint4 hd0 = *(const uint4 *) (wh0), hd1 = *(const uint4 *) (wh1);
    uint4 xs[4];
    for (int i = 0; i < 4; ++i) xs[i] = *(const uint4 *) (xp + (size_t) i * 32 * 8192);
    Sc s0 = scales(hd0, 0), s1 = scales(hd1, 0);
    uint4 f0 = unpack_frag<0>(qe0, s0), f1 = unpack_frag<0>(qe1, s1);
#define STEP(KK, KN, C0, C1, N0, N1)                                                                                  \
        {                                                                                                               \
            uint4 a[4];                                                                                                 \
            for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const uint4 *>(stage + tile_off(32 * i + r, 2 * KK + h)); \
            const uint4 n0 = unpack_frag<KN>(KN ? C0 : N0, s0), n1 = unpack_frag<KN>(KN ? C1 : N1, s1);                 \
            for (int i = 0; i < 4; ++i) {                                                                               \
                acc[0][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8 *>(&a[i]), *reinterpret_cast<const f16x8 *>(&f0), acc[0][i], 0, 0, 0); \
                acc[1][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8 *>(&a[i]), *reinterpret_cast<const f16x8 *>(&f1), acc[1][i], 0, 0, 0); \
            }                                                                                                           \
            f0 = n0; f1 = n1;                                                                                           \
        }
#define KSTEP(C0, C1, N0, N1, ks)                                                                                      \
        {                                                                                                               \
            const uint8_t * stage = lds + ((ks) & 1) * 16384;                                                           \
            uint8_t * other = lds + (((ks) + 1) & 1) * 16384;                                                           \
            if (FEAT & 32) {   /* hand-issued DMA: the compiler neither sees nor waits for it; one counted wait in front of the barrier */ \
                const uint32_t l0 = (uint32_t) (uintptr_t) (other + wave * 1024);                                       \
                const uint8_t * g0 = xd + (((ks) + 1) & 63) * 128;                                                      \
                asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\t"                     \
                             "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"             \
                             "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"             \
                             "s_add_u32 m0, m0, 0x1000\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off"                  \
                             :: "v"(g0), "v"(g0 + (size_t) 32 * 8192), "v"(g0 + (size_t) 64 * 8192), "v"(g0 + (size_t) 96 * 8192), \
                                "s"(__builtin_amdgcn_readfirstlane(l0)) : "memory", "m0", "scc");                       \
            } else if (FEAT & 16) {                                                                                     \
                for (int i = 0; i < 4; ++i)                                                                             \
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (xd + (size_t) i * 32 * 8192 + (((ks) + 1) & 63) * 128), \
                                                     (__attribute__((address_space(3))) void *) (other + (i * 4 + wave) * 1024), 16, 0, 0); \
            } else {                                                                                                    \
            for (int i = 0; i < 4; ++i) { const int c = tid + 256 * i; *reinterpret_cast<uint4 *>(other + tile_off(c / 8, c % 8)) = xs[i]; } \
            if (FEAT & 2) for (int i = 0; i < 4; ++i) xs[i] = *(const uint4 *) (xp + (size_t) i * 32 * 8192 + (((ks) + 2) & 63) * 128); } \
            STEP(0, 1, C0, C1, N0, N1) STEP(1, 2, C0, C1, N0, N1) STEP(2, 3, C0, C1, N0, N1)                            \
            if (FEAT & 1) { C0 = *(const uint4 *) (w0 + ldq(((ks) + 2) & 63)); C1 = *(const uint4 *) (w1 + ldq(((ks) + 2) & 63)); } else { C0.x += ks; } \
            if (FEAT & 4) { s0 = scales(hd0, ((ks) + 1) & 3); s1 = scales(hd1, ((ks) + 1) & 3); }                       \
            STEP(3, 0, C0, C1, N0, N1)                                                                                  \
            for (int g = 0; g < 4; ++g) { __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                            \
                for (int m = 0; m < 8; ++m) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 4, 0); } } \
            if (FEAT & 32) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                           \
            __syncthreads();                                                                                            \
        }
    for (int ks = 0; ks < nsteps; ks += 2) {
        KSTEP(qe0, qe1, qo0, qo1, ks)
        KSTEP(qo0, qo1, qe0, qe1, ks + 1)
        if ((FEAT & 1) && (ks & 3) == 2) { hd0 = *(const uint4 *) (wh0 + (size_t) (((ks >> 2) + 1) & 15) * 144); hd1 = *(const uint4 *) (wh1 + (size_t) (((ks >> 2) + 1) & 15) * 144); }
    }
    float s = 0.f;
    for (int j = 0; j < 2; ++j) for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[j][i][e];
    if (s == 1.2345f) out[0] = s;
}

int main() {
    float * out; hipMalloc(&out, 4);
    uint8_t * W; hipMalloc(&W, (size_t) 28672 * 2304 + 4096); hipMemset(W, 0x35, (size_t) 28672 * 2304 + 4096);
    uint8_t * X; hipMalloc(&X, (size_t) 512 * 8192); hipMemset(X, 0x3c, (size_t) 512 * 8192);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int nsteps = 64 * 16;
    auto run = [&](const char * name, auto kern) {
        hipLaunchKernelGGL(kern, dim3(448), dim3(256), 0, 0, out, W, X, 64);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(448), dim3(256), 0, 0, out, W, X, nsteps);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-56s %8.3f ms   %.1f cycles@2.4GHz per MFMA per SIMD\n", name, ms, ms * 1e-3 * 2.4e9 / ((double) nsteps * 32 * 448 * 4 / 1024));
    };
    run("base (register weights, static X, fixed scales)", k<0>);
    run("+ weight bytes from HBM, row-major", k<1>);
    run("+ weight bytes from HBM, fragment-major", k<9>);
    run("+ activation chunks through VGPRs", k<2>);
    run("+ activation tile by LDS DMA", k<16>);
    run("row-major weights + VGPR activations + scales (nb2d)", k<7>);
    run("fragment-major weights + VGPR activations + scales", k<15>);
    run("row-major weights + LDS DMA + scales", k<21>);
    run("fragment-major weights + LDS DMA + scales", k<29>);
    run("hand-issued LDS DMA alone", k<32>);
    run("row-major weights + hand-issued LDS DMA + scales", k<37>);
    run("fragment-major weights + hand-issued LDS DMA + scales", k<45>);
    return 0;
}
