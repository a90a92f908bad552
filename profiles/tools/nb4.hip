// nb4.hip — would a Q8-EXACT prefill K loop run within 1.2x of the f16 one?
//
// The CPU's Q4_K x Q8_K dot is integer per 32-weight sub-block: sum_j sc_j * (sum_k q_k a_k) - dmin * sum_j m_j * bsum_j, scaled once per
// 256-superblock by d * d_act (ggml-cpu-quants.c:7535-7591).  On the matrix core that is v_mfma_i32_32x32x32_i8 (2 x the f16 rate per
// k) with the 6-bit sub-scale split in two so that weight * scale stays a signed byte: sc = 8 * sc_hi + sc_lo, q * sc_hi and q * sc_lo
// <= 105: two MFMAs per 32 k, i.e. the f16 MFMA time.  What it adds is vector work and registers:
//   * an i32 accumulator per (hi, lo) beside the running f32 one: a wave's tile halves (64 rows x 128 tokens, not x 256);
//   * per superblock and accumulator element: (hi << 3) + lo, int -> float, times d_w * d_a, add: 4 VALU on 16 elements per
//     (row set, token tile) = 512 per wave and superblock, against 128 + 16 MFMAs;
//   * the mins term as two more MFMAs per (row set, tile) on the activations' block sums split in two bytes, + its rescale.
// This program measures exactly that instruction mix (operands resident, A fragments from LDS, placement by sched_barrier fences as
// in csrc/qmm_mfma_r64s.hiph), beside the f16 loop's mix on the same footing; cycles by s_memtime, every CU busy.
//   hipcc --offload-arch=gfx950 -O3 -o nb4 profiles/tools/nb4.hip && ./nb4
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define FENCE() __builtin_amdgcn_sched_barrier(0)
__device__ unsigned long long g_cycles[2][1024];

// ---- the f16 loop's mix: per 64-deep K-step and wave 64 MFMAs (2 row sets x 8 token tiles x 4 k-steps), 32 ds_read_b128, 8 B dwords x 4
// k-steps unpacked (and_or, pk_add, pk_fma), 8 scale values
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) k_f16(float * out, int nsuper) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[65536];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 65536 / 4; i += 256) ((uint32_t *) lds)[i] = 0x3c003c00u + (i & 7);
    __syncthreads();
    f32x16 acc[2][8];
    for (int j = 0; j < 2; ++j) for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;
    u32x4 q0 = { 0x12345678u + lane, 0x9abcdef0u, 0x0fedcba9u, 0x87654321u }, q1 = q0 + 1u;
    u32x4 F0 = { 0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u }, F1 = F0, G0 = F0, G1 = F0;
    u32x4 A0[8], A1[8];
    for (int i = 0; i < 8; ++i) A0[i] = A1[i] = *(const u32x4 *) (lds + i * 4096 + lane * 16);
    f16x2 ds = { (_Float16) 0.01f, (_Float16) 0.01f }, no = { (_Float16) -0.02f, (_Float16) -0.02f };
    uint32_t hdr = 0x01020304u + lane;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int sb = 0; sb < nsuper; ++sb) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                u32x4 & Ac = (kk & 1) ? A1[0] : A0[0];
                (void) Ac;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    u32x4 * Acur = (kk & 1) ? A1 : A0, * Anxt = (kk & 1) ? A0 : A1;
                    u32x4 & Fc0 = (kk & 1) ? G0 : F0, & Fc1 = (kk & 1) ? G1 : F1, & Fn0 = (kk & 1) ? F0 : G0, & Fn1 = (kk & 1) ? F1 : G1;
                    acc[0][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, Acur[i]), __builtin_bit_cast(f16x8, Fc0), acc[0][i], 0, 0, 0);
                    Anxt[i] = *(const u32x4 *) (lds + ((ks + kk) & 1) * 32768 + i * 4096 + lane * 16);
                    if (kk == 0) {                                  // one scale value of the next K-step
                        const uint32_t v = (hdr >> (i & 3) * 8) & 63u;
                        const _Float16 s = (_Float16) ((float) v * 0.01f);
                        ds[0] += s;
                    }
                    FENCE();
                    acc[1][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, Acur[i]), __builtin_bit_cast(f16x8, Fc1), acc[1][i], 0, 0, 0);
                    {
                        const uint32_t w = i < 4 ? q0[i & 3] : q1[i & 3];
                        const uint32_t src = (i & 1) ? w >> 8 : w;
                        const f16x2 t = __builtin_bit_cast(f16x2, (src & 0x000f000fu) | 0x64006400u) + f16x2{ (_Float16) -1024.f, (_Float16) -1024.f };
                        const uint32_t o = __builtin_bit_cast(uint32_t, __builtin_elementwise_fma(t, ds, no));
                        if (i < 4) Fn0[i & 3] = o; else Fn1[i & 3] = o;
                    }
                    FENCE();
                }
            }
            q0 += 0x01010101u; q1 += 0x01010101u;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int j = 0; j < 2; ++j) for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) s += acc[j][i][e];
    if (s == 1.2345f) out[0] = s;
    if (tid == 0 && blockIdx.x < 1024) g_cycles[0][blockIdx.x] = t1 - t0;
}

// ---- the exact loop's mix.  Three accumulator sets (hi, lo, running f32) fit the 512 registers of a wave only for ONE row set x 4
// token tiles (32 rows x 128 tokens: 192 accumulator registers; two row sets spilled 145 VGPRs in this very benchmark).  Per
// superblock and wave: 8 sub-blocks x 4 tiles x (hi, lo) = 64 MFMAs + 8 for the mins; 32 ds_read_b128 (one int8 A fragment per
// (tile, sub-block), shared by hi and lo); unpack 7 VALU per dword of nibbles, 16 dwords; rescale of 4 x 16 elements
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) k_i8(float * out, int nsuper) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[65536];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 65536 / 4; i += 256) ((uint32_t *) lds)[i] = 0x01020304u + (i & 7);
    __syncthreads();
    f32x16 accf[4];
    i32x16 hi[4], lo[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) { accf[i][e] = 0.f; hi[i][e] = 0; lo[i][e] = 0; }
    u32x4 q0 = { 0x12345678u + lane, 0x9abcdef0u, 0x0fedcba9u, 0x87654321u };
    i32x4 Bh[2], Bl[2], Nh[2], Nl[2];                           // [sub-block of the pair]: q * sc_hi, q * sc_lo as int8 x 16; N = the next pair's
    for (int s = 0; s < 2; ++s) { Bh[s] = i32x4{ 0x01010101, 0x02020202, 0x01010101, 0x02020202 }; Bl[s] = Bh[s]; Nh[s] = Bh[s]; Nl[s] = Bh[s]; }
    i32x4 A[2][4];
    for (int s = 0; s < 2; ++s) for (int i = 0; i < 4; ++i) A[s][i] = *(const i32x4 *) (lds + (s * 4 + i) * 4096 + lane * 16);
    uint32_t sch = 3 + (lane & 3), scl = 5 + (lane & 1);
    float dw0 = 0.01f, dm0 = 0.003f;
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int sb = 0; sb < nsuper; ++sb) {
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {                            // sub-block pair: 64 k = 16 MFMAs
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    hi[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[s][i], Bh[s], hi[i], 0, 0, 0);
                    {                                               // gap 1: half a dword of the next pair's B operands (3-4 VALU)
                        const uint32_t w = q0[i & 3];
                        const uint32_t n4 = s ? (w >> 4) & 0x0f0f0f0fu : w & 0x0f0f0f0fu;
                        const u16x2 mh = { (unsigned short) sch, (unsigned short) sch }, ml = { (unsigned short) scl, (unsigned short) scl };
                        Nh[s][i & 3] = (int) __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, n4) * mh);
                        Nl[s][i & 3] = (int) __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, n4) * ml);
                    }
                    FENCE();
                    lo[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[s][i], Bl[s], lo[i], 0, 0, 0);
                    A[s][i] = *(const i32x4 *) (lds + ((pr + 1) & 1) * 32768 + (s * 4 + i) * 4096 + lane * 16);     // gap 2: the next pair's A fragment
                    if (pr >= 2) {
                        // ... and, in the last two pairs of a superblock, the rescale of the PREVIOUS superblock's sums (kept in a second
                        // i32 set in a real kernel; here the same registers): 4 elements per gap x 16 gaps = the 64 elements of the tile set
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int t = ((pr - 2) * 8 + s * 4 + i) >> 2, el = (((pr - 2) * 8 + s * 4 + i) & 3) * 4 + e;
                            const float da = *(const float *) (lds + 60000 + ((el + lane) & 63) * 4);
                            const int v0 = (hi[t][el] << 3) + lo[t][el];
                            accf[t][el] += (float) v0 * (dw0 * da) - dm0 * da * (float) (hi[t][(el + 1) & 15] & 0xffff);
                        }
                    }
                    FENCE();
                }
            q0 += 0x01010101u;
#pragma unroll
            for (int s = 0; s < 2; ++s) { Bh[s] = Nh[s]; Bl[s] = Nl[s]; }
        }
        // mins: two MFMAs per tile on the block sums split in two bytes (K = 8 of the 32 used)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            hi[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[0][i], Bh[0], hi[i], 0, 0, 0);  FENCE();
            lo[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[1][i], Bl[0], lo[i], 0, 0, 0);  FENCE();
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += accf[i][e] + (float) hi[i][e] + (float) lo[i][e];
    if (s == 1.2345f) out[0] = s;
    if (tid == 0 && blockIdx.x < 1024) g_cycles[1][blockIdx.x] = t1 - t0;
}

int main() {
    float * out; hipMalloc(&out, 4);
    const int nsuper = 256, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 2; ++which)
        for (int round = 0; round < 3; ++round) {
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(k_f16, dim3(blocks), dim3(256), 0, 0, out, nsuper);
            else            hipLaunchKernelGGL(k_i8, dim3(blocks), dim3(256), 0, 0, out, nsuper);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            static unsigned long long h[2][1024];
            hipMemcpyFromSymbol(h, HIP_SYMBOL(g_cycles), sizeof(h));
            std::vector<double> c;
            for (int i = 0; i < blocks; ++i) c.push_back((double) h[which][i] / nsuper);
            std::sort(c.begin(), c.end());
            // MACs per superblock and wave: f16 tile 64 rows x 256 tokens x 256 k; exact tile 32 x 128 x 256
            const double macs = which == 0 ? 64.0 * 256 * 256 : 32.0 * 128 * 256;
            printf("%-34s round %d: %8.3f ms, %7.0f cycles per superblock and wave (median), %.2f cycles per 1024 MACs, %.1f TFLOP/s\n",
                   which == 0 ? "f16 mix (64 x 256 tile per wave)" : "int8-exact mix (32 x 128 tile)", round, ms, c[c.size() / 2], c[c.size() / 2] / (macs / 1024.0),
                   2.0 * macs * nsuper * 4 * blocks / (ms * 1e-3) * 1e-12);
        }
    return 0;
}
