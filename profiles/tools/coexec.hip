// can the VALU of one wave run under the MFMAs of another wave on the same SIMD?  8 waves per block: waves 0-3 (one per
// SIMD) multiply (ds_read_b128 + v_mfma_f32_32x32x16_f16), waves 4-7 run packed-f16 FMAs; time each alone and both.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ int tile_off(int r, int s) { return r * 128 + ((s ^ ((r >> 1) & 7)) << 4); }

template <bool READ>
__global__ void __launch_bounds__(512) k(float * out, int iters, int mode) {   // mode bit0: MFMA waves work, bit1: VALU waves work
    __shared__ __attribute__((aligned(16))) uint8_t lds[32768];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    for (int i = tid; i < 32768 / 4; i += blockDim.x) ((uint32_t *) lds)[i] = 0x3c003c00u + (i & 7);
    __syncthreads();
    float s = 0.f;
    if (wave < 4) {
        if (!(mode & 1)) return;
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        uint4 b = make_uint4(0x3c003c00u, 0x3c003c01u, 0x3c003c00u, 0x3c003c00u);
        uint4 a[4];
        for (int i = 0; i < 4; ++i) a[i] = make_uint4(0x3c003c00u, 0x3c013c00u, 0x3c003c00u + i, 0x3c003c00u);
        for (int it = 0; it < iters; ++it) {
            const uint8_t * stage = lds + (it & 1) * 16384;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                if (READ) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const uint4 *>(stage + tile_off(32 * i + r, 2 * kk + h));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8 *>(&a[i]), *reinterpret_cast<const f16x8 *>(&b), acc[i], 0, 0, 0);
            }
        }
        for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    } else {
        if (!(mode & 2)) return;
        f16x2 v[8];
        for (int i = 0; i < 8; ++i) v[i] = f16x2{ (_Float16) (1.0f + i), (_Float16) (0.5f * lane) };
        const f16x2 m = { (_Float16) 1.0009765625f, (_Float16) 0.99951171875f }, c = { (_Float16) 0.001f, (_Float16) -0.001f };
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 15; ++rep)                // 15 x 8 = 120 packed FMAs per iteration (one K-step's unpack)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_elementwise_fma(v[i], m, c);
        }
        for (int i = 0; i < 8; ++i) s += (float) v[i][0] + (float) v[i][1];
    }
    if (s == 1.2345f) out[0] = s;
}

int main() {
    float * out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096;
    auto run = [&](const char * name, auto kern, int mode) {
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, out, 64, mode);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, out, iters, mode);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %8.3f ms   %.0f cycles per iteration @2.4GHz (16 MFMAs = 512 ideal; 120 VALU = 480 ideal)\n", name, ms, ms * 1e-3 * 2.4e9 / iters);
    };
    run("MFMA waves alone, with LDS reads", k<true>, 1);
    run("MFMA waves alone, no reads", k<false>, 1);
    run("VALU waves alone", k<true>, 2);
    run("both, with LDS reads", k<true>, 3);
    run("both, no reads", k<false>, 3);
    return 0;
}
