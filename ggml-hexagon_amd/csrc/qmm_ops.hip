// qmm_ops.hip — the per-layer glue ops around the quantized MUL_MAT path (include/ggml_mi355x_ops.h).
// gfx950 only, no CPU path.  Every kernel is HBM-bound elementwise / row work except the small F16 attention matmuls,
// which run on v_mfma_f32_32x32x16_f16.  Semantics follow the ggml CPU backend; each kernel cites the function it restates.
// Parity: the reference's tests/test-backend-ops.cpp (built unmodified into oracle/_ref) against the CPU backend.

#include "qmm_host.h"
#include "../../include/ggml_mi355x_ops.h"
#include "qmm_device.hiph"

#include <hip/hip_fp16.h>
#include <cmath>

using namespace qmm;

namespace {

enum : int { G_F32 = 0, G_F16 = 1, G_I32 = 26 };

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
constexpr int DOT_T = 128;           // threads per dst element of mul_mat_dot_block_kernel / per expert of moe_router_logits_kernel
typedef float    f32x16v __attribute__((ext_vector_type(16)));

struct Shape {           // extents and byte strides of one operand
    int64_t ne[4];
    int64_t nb[4];
};
Shape shape_of(const qmm_tensor * t) {
    Shape s;
    for (int i = 0; i < 4; ++i) { s.ne[i] = t->ne[i]; s.nb[i] = t->nb[i]; }
    return s;
}
int64_t nelements(const qmm_tensor * t) { return t->ne[0] * t->ne[1] * t->ne[2] * t->ne[3]; }
int64_t nrows(const qmm_tensor * t) { return t->ne[1] * t->ne[2] * t->ne[3]; }
int     esize(int type) { return type == G_F32 || type == G_I32 ? 4 : type == G_F16 ? 2 : 0; }
bool    same_shape(const qmm_tensor * a, const qmm_tensor * b) {
    return a->ne[0] == b->ne[0] && a->ne[1] == b->ne[1] && a->ne[2] == b->ne[2] && a->ne[3] == b->ne[3];
}
bool contiguous(const qmm_tensor * t) {
    const int es = esize(t->type);
    if (!es) return false;
    int64_t nb = es;
    for (int i = 0; i < 4; ++i) {
        if (t->ne[i] != 1 && t->nb[i] != nb) return false;
        nb *= t->ne[i];
    }
    return true;
}
// rows are dense runs of elements (nb[0] == element size); rows themselves may sit anywhere
bool dense_rows(const qmm_tensor * t) { return esize(t->type) && (t->nb[0] == esize(t->type) || t->ne[0] == 1); }
bool fits_u32(const qmm_tensor * t) { return nelements(t) < ((int64_t) 1 << 31); }
bool aligned_to(const qmm_tensor * t, int a) {
    return (uintptr_t) t->data % a == 0 && t->nb[1] % a == 0 && t->nb[2] % a == 0 && t->nb[3] % a == 0;
}

// wave_sum / wave_max: the DPP reductions of qmm_device.hiph (no LDS-permute traffic; the __shfl_xor butterflies this file began
// with cost ~0.3 us per reduction: 16 us of the 37 us of attn_prefill_kernel's last token tile were its per-row soft-max)
// sum over an aligned group of 8 lanes, result in all 8
__device__ __forceinline__ float sum8(float v) {
    v += dpp_mov<DPP_QUAD_X1>(v);
    v += dpp_mov<DPP_QUAD_X2>(v);
    v += dpp_mov<DPP_ROW_HALF_MIRROR>(v);
    return v;
}
// block-wide reductions (up to 16 waves) through 16 floats of LDS
template <bool MAX> __device__ __forceinline__ float block_reduce(float v, float * red) {
    v = MAX ? wave_max(v) : wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const int nw = blockDim.x >> 6;
    float r = red[0];
    for (int i = 1; i < nw; ++i) r = MAX ? fmaxf(r, red[i]) : r + red[i];
    return r;
}

// 1 / sqrt(mean of squares + eps) of a row from the threads' partial sums, for the RMS-norm kernels: block_reduce's partial sums and order
// of additions; with 16 waves (four to a SIMD) the scalar tail (the wave sums out of LDS, two IEEE divisions, a square root: ~150
// dependent instructions) is run by wave 0 alone and handed over through LDS: run by every wave it cost ~1 us of the ~4.7 us launch
// (round 3, the same finding as in the mat-vec staging: profiles/tools/ex_times.py).  `red` holds 17 floats.
template <int NT> __device__ __forceinline__ float rms_scale_block(float sum, float * red, const float ne0, const float eps) {
    sum = wave_sum(sum);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (NT <= 512) {
        float r = red[0];
        for (int i = 1; i < NT / 64; ++i) r += red[i];
        return 1.0f / sqrtf(r / ne0 + eps);
    }
    if (threadIdx.x < 64) {
        float r = red[0];
        for (int i = 1; i < NT / 64; ++i) r += red[i];
        const float sc = 1.0f / sqrtf(r / ne0 + eps);
        if (threadIdx.x == 0) red[16] = sc;
    }
    __syncthreads();
    return red[16];
}

// row index -> byte offset of the row in a (possibly broadcast) operand
__device__ __forceinline__ void row_coords(uint32_t row, uint32_t ne1, uint32_t ne2, uint32_t & i1, uint32_t & i2, uint32_t & i3) {
    i1 = row % ne1;
    const uint32_t t = row / ne1;
    i2 = t % ne2;
    i3 = t / ne2;
}

// ------------------------------------------------------------------------------------------------ binary ops
// ggml_compute_forward_add/sub/mul/div (ggml-cpu/binary-ops.cpp; the DSP copy of add: kernels/ggml-dsp.c:991-1066):
// dst[i] = src0[i] op src1[i mod ne1x], src1 broadcast over every dimension it is smaller in.
template <int OP> __device__ __forceinline__ float bin(float a, float b) {
    return OP == QMM_OP_ADD ? a + b : OP == QMM_OP_SUB ? a - b : OP == QMM_OP_MUL ? a * b : a / b;
}

// one block per group of rows; vector path when every row is 16-byte aligned and src1 has full rows
template <int OP, bool VEC>
__global__ void __launch_bounds__(256)
binary_kernel(const char * __restrict__ a, const char * __restrict__ b, char * __restrict__ d, const Shape sa, const Shape sb,
              const Shape sd, const uint32_t rows) {
    const uint32_t ne0 = (uint32_t) sd.ne[0];
    const uint32_t per_row = VEC ? ne0 / 4 : ne0;
    const uint32_t rpb = per_row >= 256 ? 1 : 256 / per_row;          // rows per block for short rows
    const uint32_t tpr = per_row >= 256 ? 256 : per_row;              // threads per row
    const uint32_t lr = threadIdx.x / tpr;
    if (lr >= rpb) return;
    const uint32_t row = blockIdx.x * rpb + lr;
    if (row >= rows) return;
    uint32_t i1, i2, i3;
    row_coords(row, (uint32_t) sd.ne[1], (uint32_t) sd.ne[2], i1, i2, i3);
    const char * pa = a + i1 * sa.nb[1] + i2 * sa.nb[2] + i3 * sa.nb[3];
    const char * pb = b + (i1 % (uint32_t) sb.ne[1]) * sb.nb[1] + (i2 % (uint32_t) sb.ne[2]) * sb.nb[2] + (i3 % (uint32_t) sb.ne[3]) * sb.nb[3];
    char *       pd = d + i1 * sd.nb[1] + i2 * sd.nb[2] + i3 * sd.nb[3];
    for (uint32_t i = threadIdx.x % tpr; i < per_row; i += tpr) {
        if (VEC) {
            const float4 x = ((const float4 *) pa)[i], y = ((const float4 *) pb)[i];
            ((float4 *) pd)[i] = make_float4(bin<OP>(x.x, y.x), bin<OP>(x.y, y.y), bin<OP>(x.z, y.z), bin<OP>(x.w, y.w));
        } else {
            const float y = *(const float *) (pb + (i % (uint32_t) sb.ne[0]) * sb.nb[0]);
            ((float *) pd)[i] = bin<OP>(((const float *) pa)[i], y);
        }
    }
}

// ------------------------------------------------------------------------------------------------ unary / scale
template <int OP> __device__ __forceinline__ float una(float x, float p) {
    switch (OP) {
        case QMM_OP_SCALE:      return x * p;                                            // ggml_vec_scale_f32
        case QMM_OP_SILU:       return x / (1.0f + expf(-x));                            // ggml_silu_f32 (ggml-cpu.c)
        case QMM_OP_GELU:       return 0.5f * x * (1.0f + tanhf(0.79788456080286535587989211986876f * x * (1.0f + 0.044715f * x * x)));
        case QMM_OP_GELU_QUICK: return x * (1.0f / (1.0f + expf(-1.702f * x)));
        case QMM_OP_RELU:       return x > 0.0f ? x : 0.0f;
        case QMM_OP_TANH:       return tanhf(x);
        case QMM_OP_SIGMOID:    return 1.0f / (1.0f + expf(-x));
        case QMM_OP_NEG:        return -x;
        case QMM_OP_EXP:        return expf(x);
        default:                return x;
    }
}
// contiguous f32; MUL2: dst = f(a) * b (SwiGLU: silu(gate) * up, the two nodes build_ffn emits back to back)
template <int OP, bool MUL2>
__global__ void __launch_bounds__(256)
unary_kernel(const float * __restrict__ a, const float * __restrict__ b, float * __restrict__ d, const uint32_t n, const float p) {
    const uint32_t i4 = (blockIdx.x * 256u + threadIdx.x) * 4u;
    if (i4 + 3 < n && (((uintptr_t) a | (uintptr_t) d | (uintptr_t) b) & 15) == 0) {
        const float4 x = *(const float4 *) (a + i4);
        float4 r = make_float4(una<OP>(x.x, p), una<OP>(x.y, p), una<OP>(x.z, p), una<OP>(x.w, p));
        if (MUL2) {
            const float4 y = *(const float4 *) (b + i4);
            r.x *= y.x; r.y *= y.y; r.z *= y.z; r.w *= y.w;
        }
        *(float4 *) (d + i4) = r;
    } else {
        for (uint32_t i = i4; i < n && i < i4 + 4; ++i) d[i] = una<OP>(a[i], p) * (MUL2 ? b[i] : 1.0f);
    }
}

// ------------------------------------------------------------------------------------------------ RMS_NORM
// ggml_compute_forward_rms_norm_f32 (ggml-cpu.c:6254-6300): mean of squares over the row, y = x / sqrt(mean + eps).
// One block per row; the row stays in registers (<= 8 values per thread) or is re-read from L2.
// MUL: y *= w[i0] — the norm weight ggml_mul of build_norm, fused when the plugin sees the pair.
template <bool MUL>
__global__ void __launch_bounds__(256)
rms_norm_kernel(const char * __restrict__ x, const float * __restrict__ w, char * __restrict__ y, const Shape sx, const Shape sy, const float eps) {
    __shared__ float red[4];
    uint32_t i1, i2, i3;
    row_coords(blockIdx.x, (uint32_t) sx.ne[1], (uint32_t) sx.ne[2], i1, i2, i3);
    const float * px = (const float *) (x + i1 * sx.nb[1] + i2 * sx.nb[2] + i3 * sx.nb[3]);
    float *       py = (float *) (y + i1 * sy.nb[1] + i2 * sy.nb[2] + i3 * sy.nb[3]);
    const uint32_t n = (uint32_t) sx.ne[0];
    float keep[8];
    float sum = 0.0f;
    const bool in_regs = n <= 256 * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t i = threadIdx.x + j * 256;
        keep[j] = in_regs && i < n ? px[i] : 0.0f;
        sum += keep[j] * keep[j];
    }
    if (!in_regs)
        for (uint32_t i = threadIdx.x; i < n; i += 256) sum += px[i] * px[i];
    sum = block_reduce<false>(sum, red);
    const float scale = 1.0f / sqrtf(sum / (float) n + eps);
    if (in_regs) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t i = threadIdx.x + j * 256;
            if (i < n) py[i] = MUL ? keep[j] * scale * w[i] : keep[j] * scale;
        }
    } else {
        for (uint32_t i = threadIdx.x; i < n; i += 256) py[i] = MUL ? px[i] * scale * w[i] : px[i] * scale;
    }
}

// ggml_compute_forward_norm_f32 (ggml-cpu.c:6183-6232), the LayerNorm of GPT-2 / Falcon / Phi-style models: mean over the row,
// then the variance of the centred values, y = (x - mean) / sqrt(var + eps).  One block per row, two passes over L2.
__global__ void __launch_bounds__(256)
norm_kernel(const char * __restrict__ x, char * __restrict__ y, const Shape sx, const Shape sy, const float eps) {
    __shared__ float red[4];
    uint32_t i1, i2, i3;
    row_coords(blockIdx.x, (uint32_t) sx.ne[1], (uint32_t) sx.ne[2], i1, i2, i3);
    const float * px = (const float *) (x + i1 * sx.nb[1] + i2 * sx.nb[2] + i3 * sx.nb[3]);
    float *       py = (float *) (y + i1 * sy.nb[1] + i2 * sy.nb[2] + i3 * sy.nb[3]);
    const uint32_t n = (uint32_t) sx.ne[0];
    float sum = 0.0f;
    for (uint32_t i = threadIdx.x; i < n; i += 256) sum += px[i];
    const float mean = block_reduce<false>(sum, red) / (float) n;
    float sum2 = 0.0f;
    for (uint32_t i = threadIdx.x; i < n; i += 256) { const float v = px[i] - mean; sum2 += v * v; }
    const float scale = 1.0f / sqrtf(block_reduce<false>(sum2, red) / (float) n + eps);
    for (uint32_t i = threadIdx.x; i < n; i += 256) py[i] = (px[i] - mean) * scale;
}

// The same for rows of up to NT*16 floats with 16-byte aligned rows: the row is read once with float4 loads and kept in
// registers, the norm weight is requested before the reduction so its latency hides behind it.  Few rows (token generation)
// run with 1024 threads per row, many rows (prefill) with 256.
// ADD: x = a + b first, also written to `s` — the residual add that feeds the norm (ffn_inp = cur + inpSA; next layer's
// inpL = ffn_out + ffn_inp, src/llama-model.cpp:4280, 4340-4346) in the same pass.
template <bool MUL, bool ADD, int NT>
__global__ void __launch_bounds__(NT)
rms_norm_vec_kernel(const char * __restrict__ x, const char * __restrict__ b, const float * __restrict__ w, char * __restrict__ y,
                    char * __restrict__ s, const Shape sx, const Shape sb, const Shape sy, const Shape ss, const float eps) {
    __shared__ float red[17];
    uint32_t i1, i2, i3;
    row_coords(blockIdx.x, (uint32_t) sx.ne[1], (uint32_t) sx.ne[2], i1, i2, i3);
    const float4 * px = (const float4 *) (x + i1 * sx.nb[1] + i2 * sx.nb[2] + i3 * sx.nb[3]);
    float4 *       py = (float4 *) (y + i1 * sy.nb[1] + i2 * sy.nb[2] + i3 * sy.nb[3]);
    const uint32_t n4 = (uint32_t) sx.ne[0] / 4;
    float4 v[4], wv[4];
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t i = threadIdx.x + j * NT;
        v[j] = i < n4 ? px[i] : make_float4(0, 0, 0, 0);
        if (MUL) wv[j] = i < n4 ? ((const float4 *) w)[i] : make_float4(0, 0, 0, 0);
    }
    if (ADD) {
        const float4 * pb = (const float4 *) (b + i1 * sb.nb[1] + i2 * sb.nb[2] + i3 * sb.nb[3]);
        float4 *       ps = (float4 *) (s + i1 * ss.nb[1] + i2 * ss.nb[2] + i3 * ss.nb[3]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t i = threadIdx.x + j * NT;
            if (i < n4) {
                const float4 t = pb[i];
                v[j].x += t.x; v[j].y += t.y; v[j].z += t.z; v[j].w += t.w;
                ps[i] = v[j];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) sum += v[j].x * v[j].x + v[j].y * v[j].y + v[j].z * v[j].z + v[j].w * v[j].w;
    const float scale = rms_scale_block<NT>(sum, red, (float) sx.ne[0], eps);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t i = threadIdx.x + j * NT;
        if (i < n4) {
            float4 r = make_float4(v[j].x * scale, v[j].y * scale, v[j].z * scale, v[j].w * scale);
            if (MUL) { r.x *= wv[j].x; r.y *= wv[j].y; r.z *= wv[j].z; r.w *= wv[j].w; }
            py[i] = r;
        }
    }
}

// ------------------------------------------------------------------------------------------------ SOFT_MAX
// ggml_compute_forward_soft_max_f32 (ggml-cpu.c:8261-8352): v = x*scale + slope(head)*mask[row % ne01]; softmax over the row.
// One block per row.  Rows up to 8192 values are staged in LDS; longer rows use dst as the staging area.
template <bool MASK_F16>
__global__ void __launch_bounds__(256)
soft_max_kernel(const float * __restrict__ x, const void * __restrict__ mask, float * __restrict__ y, const uint32_t nc, const uint32_t ne01,
                const uint32_t ne02, const float scale, const float max_bias, const float m0, const float m1, const uint32_t n_head_log2) {
    extern __shared__ float stage[];
    __shared__ float red[4];
    const uint32_t row = blockIdx.x;
    const uint32_t h = (row / ne01) % ne02;
    const float slope = max_bias > 0.0f ? (h < n_head_log2 ? powf(m0, (float) (h + 1)) : powf(m1, (float) (2 * (h - n_head_log2) + 1))) : 1.0f;
    const float * px = x + (size_t) row * nc;
    float *       py = y + (size_t) row * nc;
    const bool in_lds = nc <= 8192;
    float * v = in_lds ? stage : py;
    const size_t moff = (size_t) (row % ne01) * nc;
    float mx = -INFINITY;
    for (uint32_t i = threadIdx.x; i < nc; i += 256) {
        float t = px[i] * scale;
        if (mask) t += slope * (MASK_F16 ? __half2float(((const __half *) mask)[moff + i]) : ((const float *) mask)[moff + i]);
        v[i] = t;
        mx = fmaxf(mx, t);
    }
    mx = block_reduce<true>(mx, red);
    float sum = 0.0f;
    for (uint32_t i = threadIdx.x; i < nc; i += 256) {
        const float e = expf(v[i] - mx);          // a fully masked row (-inf everywhere) gives NaN on the CPU too
        v[i] = e;
        sum += e;
    }
    sum = block_reduce<false>(sum, red);
    const float inv = 1.0f / sum;
    for (uint32_t i = threadIdx.x; i < nc; i += 256) py[i] = v[i] * inv;
}

// Rows of up to 1024 values (attention over a prompt of that length): one WAVE per row, the row in registers (float4 per
// lane and trip), reductions on the wave only: no LDS, no barrier, four rows per workgroup.  f32 mask, no ALiBi.
template <int V4>      // float4 per lane: the row has at most 256 * V4 values
__global__ void __launch_bounds__(256)
soft_max_wave_kernel(const float * __restrict__ x, const float * __restrict__ mask, float * __restrict__ y, const uint32_t nc, const uint32_t ne01,
                     const uint32_t rows, const float scale) {
    const uint32_t row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float4 * px = (const float4 *) (x + (size_t) row * nc);
    const float4 * pm = mask ? (const float4 *) (mask + (size_t) (row % ne01) * nc) : nullptr;
    float4 *       py = (float4 *) (y + (size_t) row * nc);
    const uint32_t n4 = nc / 4;
    float4 v[V4];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < V4; ++j) {
        const uint32_t i = lane + 64 * j;
        if (i < n4) {
            float4 t = px[i];
            t.x *= scale; t.y *= scale; t.z *= scale; t.w *= scale;
            if (pm) { const float4 m = pm[i]; t.x += m.x; t.y += m.y; t.z += m.z; t.w += m.w; }
            v[j] = t;
            mx = fmaxf(fmaxf(mx, fmaxf(t.x, t.y)), fmaxf(t.z, t.w));
        } else {
            v[j] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        }
    }
    mx = wave_max(mx);
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < V4; ++j) {
        v[j].x = expf(v[j].x - mx); v[j].y = expf(v[j].y - mx); v[j].z = expf(v[j].z - mx); v[j].w = expf(v[j].w - mx);
        sum += v[j].x + v[j].y + v[j].z + v[j].w;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int j = 0; j < V4; ++j) {
        const uint32_t i = lane + 64 * j;
        if (i < n4) py[i] = make_float4(v[j].x * inv, v[j].y * inv, v[j].z * inv, v[j].w * inv);
    }
}

// ------------------------------------------------------------------------------------------------ ROPE
// ggml_compute_forward_rope_f32 (ggml-cpu.c:8708-8893) with rope_yarn / ggml_rope_cache_init (:8610-8648): modes "normal"
// (pairs (2p, 2p+1)) and NEOX (pairs (p, p + n_dims/2)); channels >= n_dims are copied.  theta is built by the same
// repeated f32 multiply as the CPU cache (theta *= theta_scale), so it carries the same rounding.
struct RopeParams {
    int   n_dims, neox;
    float theta_scale, freq_scale, ext_factor, attn_factor, corr0, corr1;
};
// cos / sin of pair p at position pos (rope_yarn + ggml_rope_cache_init, ggml-cpu.c:8610-8648): theta by the CPU's repeated f32 multiply
__device__ __forceinline__ void rope_cs_ff(const float pos, const uint32_t p, const RopeParams & rp, const float ffp, float & c, float & s);
__device__ __forceinline__ void rope_cs(const float pos, const uint32_t p, const RopeParams & rp, const float * __restrict__ ff, float & c, float & s) {
    rope_cs_ff(pos, p, rp, ff ? ff[p] : 1.0f, c, s);
}
// (ffp: the pair's frequency factor, already loaded)
__device__ __forceinline__ void rope_cs_ff(const float pos, const uint32_t p, const RopeParams & rp, const float ffp, float & c, float & s) {
    float theta = pos;
    if (rp.n_dims <= 128) {
        // the CPU's repeated multiply, lane p keeping step p of it: 64 straight-line steps.  (As `for (k < p)` the trip count
        // differs per lane and every step pays a compare, an exec update and a branch: 1.2 us of the decode attention's prologue)
        float t = pos;
#pragma unroll
        for (uint32_t k = 0; k < 64; ++k) { theta = k == p ? t : theta; t *= rp.theta_scale; }
    } else {
        for (uint32_t k = 0; k < p; ++k) theta *= rp.theta_scale;
    }
    const float theta_extrap = theta / ffp;
    const float theta_interp = rp.freq_scale * theta_extrap;
    float th = theta_interp, mscale = rp.attn_factor;
    if (rp.ext_factor != 0.0f) {
        const float yv = ((float) p - rp.corr0) / fmaxf(0.001f, rp.corr1 - rp.corr0);
        const float ramp_mix = (1.0f - fminf(1.0f, fmaxf(0.0f, yv))) * rp.ext_factor;
        th = theta_interp * (1.0f - ramp_mix) + theta_extrap * ramp_mix;
        mscale *= 1.0f + 0.1f * logf(1.0f / rp.freq_scale);
    }
    // th can be thousands of radians (position times the first frequencies): ocml's sinf / cosf take their large-argument
    // path there.  One reduction in double (exact to ~1e-13 rad) and the fast pair on the remainder give the same values
    // to f32 rounding at a fraction of the instructions.
    // Round 3: the reduction goes to the quadrant, |y| <= pi/4, and the pair comes from the two short polynomials libm itself uses
    // on that interval (cephes sinf / cosf, ~1 ulp): a third of the instructions of ocml's pair, which re-check the argument's range.
    const double td = (double) th;
    const double qd = rint(td * 0.6366197723675814);
    const float y = (float) fma(qd, -1.5707963267948966, td), z = y * y;
    const float ys = y + y * z * (-1.6666654611e-1f + z * (8.3321608736e-3f + z * -1.9515295891e-4f));
    const float yc = 1.0f - 0.5f * z + z * z * (4.166664568298827e-2f + z * (-1.388731625493765e-3f + z * 2.443315711809948e-5f));
    const int q = (int) (long long) qd & 3;
    c = (q == 0 ? yc : q == 1 ? -ys : q == 2 ? -yc : ys) * mscale;
    s = (q == 0 ? ys : q == 1 ? yc : q == 2 ? -ys : -yc) * mscale;
}
// one pair of one row; TD = float or __half (the K-cache store of build_attn is rope(k) -> f16)
template <typename TD>
__device__ __forceinline__ void rope_pair(const char * __restrict__ x, const int32_t * __restrict__ pos, const float * __restrict__ ff,
                                          char * __restrict__ y, const Shape & sx, const Shape & sy, const RopeParams & rp, const uint32_t gid) {
    const uint32_t half = (uint32_t) sx.ne[0] / 2;
    const uint32_t p = gid % half;
    uint32_t i1, i2, i3;
    row_coords(gid / half, (uint32_t) sx.ne[1], (uint32_t) sx.ne[2], i1, i2, i3);
    const float * px = (const float *) (x + i1 * sx.nb[1] + i2 * sx.nb[2] + i3 * sx.nb[3]);
    TD *          py = (TD *) (y + i1 * sy.nb[1] + i2 * sy.nb[2] + i3 * sy.nb[3]);
    const uint32_t i0 = 2 * p;
    if (i0 >= (uint32_t) rp.n_dims) {             // pass-through channels
        py[i0] = (TD) px[i0];
        py[i0 + 1] = (TD) px[i0 + 1];
        return;
    }
    float c, s;
    rope_cs((float) pos[i2], p, rp, ff, c, s);
    const uint32_t ia = rp.neox ? p : i0, ib = rp.neox ? p + rp.n_dims / 2 : i0 + 1;
    const float x0 = px[ia], x1 = px[ib];
    py[ia] = (TD) (x0 * c - x1 * s);
    py[ib] = (TD) (x0 * s + x1 * c);
}
__global__ void __launch_bounds__(256)
rope_kernel(const char * __restrict__ x, const int32_t * __restrict__ pos, const float * __restrict__ ff, char * __restrict__ y,
            const Shape sx, const Shape sy, const RopeParams rp, const uint32_t total_pairs) {
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    if (gid < total_pairs) rope_pair<float>(x, pos, ff, y, sx, sy, rp, gid);
}

// ------------------------------------------------------------------------------------------------ CPY / CONT / DUP
// ggml_compute_forward_dup (ggml-cpu.c): element i of src (in src's index order) goes to element i of dst (in dst's
// index order); shapes may differ, element counts agree.  F32 / F16 either side.
template <typename T> __device__ __forceinline__ float ld_as_f32(const char * p);
template <> __device__ __forceinline__ float ld_as_f32<float>(const char * p) { return *(const float *) p; }
template <> __device__ __forceinline__ float ld_as_f32<__half>(const char * p) { return __half2float(*(const __half *) p); }
template <typename T> __device__ __forceinline__ void st_from_f32(char * p, float v);
template <> __device__ __forceinline__ void st_from_f32<float>(char * p, float v) { *(float *) p = v; }
template <> __device__ __forceinline__ void st_from_f32<__half>(char * p, float v) { *(__half *) p = __float2half(v); }

__device__ __forceinline__ size_t elem_offset(uint32_t i, const Shape & s) {
    const uint32_t i0 = i % (uint32_t) s.ne[0];
    uint32_t t = i / (uint32_t) s.ne[0];
    const uint32_t i1 = t % (uint32_t) s.ne[1];
    t /= (uint32_t) s.ne[1];
    const uint32_t i2 = t % (uint32_t) s.ne[2], i3 = t / (uint32_t) s.ne[2];
    return (size_t) i0 * s.nb[0] + (size_t) i1 * s.nb[1] + (size_t) i2 * s.nb[2] + (size_t) i3 * s.nb[3];
}
template <typename TS, typename TD>
__global__ void __launch_bounds__(256)
cpy_kernel(const char * __restrict__ x, char * __restrict__ y, const Shape sx, const Shape sy, const uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    st_from_f32<TD>(y + elem_offset(i, sy), ld_as_f32<TS>(x + elem_offset(i, sx)));
}
// 2-D transpose through LDS for the case the scalar kernel does worst: src is walked along its dim 1 (stride one element)
// while dst rows are dense — the V-cache store of build_attn (v_cur^T [n_tokens, n_embd] -> rows of the transposed cache).
// src element (i0, i1) at x + i0*sx0 + i1*sx1 with sx1 == sizeof(TS); dst element at y + i0*sizeof(TD) + i1*sy1.
template <typename TS, typename TD>
__global__ void __launch_bounds__(256)
cpy_transpose_kernel(const char * __restrict__ x, char * __restrict__ y, const uint32_t ne0, const uint32_t ne1, const int64_t sx0,
                     const int64_t sy1) {
    __shared__ float tile[32][33];
    const uint32_t b0 = blockIdx.x * 32, b1 = blockIdx.y * 32;
    const uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t i0 = b0 + ty + j * 8, i1 = b1 + tx;                 // consecutive threads walk src's dense direction
        if (i0 < ne0 && i1 < ne1) tile[ty + j * 8][tx] = ld_as_f32<TS>(x + (size_t) i0 * sx0 + (size_t) i1 * sizeof(TS));
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t i0 = b0 + tx, i1 = b1 + ty + j * 8;                 // ... and dst's dense direction
        if (i0 < ne0 && i1 < ne1) st_from_f32<TD>(y + (size_t) i0 * sizeof(TD) + (size_t) i1 * sy1, tile[tx][ty + j * 8]);
    }
}

// What follows the q/k/v projections of a few-token batch in build_attn (src/llama-graph.cpp:1306-1365), as ONE launch:
//   rope(q) -> f32            rope(k) -> the f16 K cache (ggml_rope_ext + ggml_cpy(k_cur, k_cache_view))
//   v       -> the f16 (transposed) V cache (ggml_cpy(v_cur^T, v_cache_view))
// The grid is the concatenation of the three index spaces; k and v parts are optional (nk = 0 / nv = 0).
// the same for ROPE_HC heads of one (pair, token): cos / sin (the double-precision reduction and two polynomials, ~150 instructions) once for
// all of them; round 3: one thread per (pair, head, token) made the 512-token launch ALU-bound at 14.5 us for 22 MB of traffic
constexpr uint32_t ROPE_HC = 8;
template <typename TD>
__device__ __forceinline__ void rope_pair_heads(const char * __restrict__ x, const int32_t * __restrict__ pos, const float * __restrict__ ff,
                                                char * __restrict__ y, const Shape & sx, const Shape & sy, const RopeParams & rp, const uint32_t gid) {
    const uint32_t half = (uint32_t) sx.ne[0] / 2, ne1 = (uint32_t) sx.ne[1], nch = (ne1 + ROPE_HC - 1) / ROPE_HC;
    const uint32_t p = gid % half;
    uint32_t c1, i2, i3;
    row_coords(gid / half, nch, (uint32_t) sx.ne[2], c1, i2, i3);
    const uint32_t i0 = 2 * p, h0 = c1 * ROPE_HC, h1 = h0 + ROPE_HC < ne1 ? h0 + ROPE_HC : ne1;
    const bool pass = i0 >= (uint32_t) rp.n_dims;                // pass-through channels
    float c = 1.0f, s = 0.0f;
    if (!pass) rope_cs((float) pos[i2], p, rp, ff, c, s);
    const uint32_t ia = pass || !rp.neox ? i0 : p, ib = pass || !rp.neox ? i0 + 1 : p + rp.n_dims / 2;
    for (uint32_t i1 = h0; i1 < h1; ++i1) {
        const float * px = (const float *) (x + i1 * sx.nb[1] + i2 * sx.nb[2] + i3 * sx.nb[3]);
        TD *          py = (TD *) (y + i1 * sy.nb[1] + i2 * sy.nb[2] + i3 * sy.nb[3]);
        const float x0 = px[ia], x1 = px[ib];
        if (pass) { py[ia] = (TD) x0; py[ib] = (TD) x1; }
        else      { py[ia] = (TD) (x0 * c - x1 * s); py[ib] = (TD) (x0 * s + x1 * c); }
    }
}
__host__ __device__ inline uint32_t rope_heads_threads(const Shape & sx) {       // threads rope_pair_heads wants for a tensor
    return (uint32_t) (sx.ne[0] / 2 * ((sx.ne[1] + ROPE_HC - 1) / ROPE_HC) * sx.ne[2] * sx.ne[3]);
}

struct RopeStoreArgs {
    const char * q; char * qd; const char * k; char * kd; const char * v; char * vd;
    const int32_t * pos; const float * ff;
    Shape sq, sqd, sk, skd, sv, svd;
    RopeParams rp;
    uint32_t nq, nk, nv;          // threads for q, for k (rope_heads_threads: a thread takes a pair of ROPE_HC heads), elements of v
};
// VT: the v part is a 2-D transpose (v_cur^T, dense along its dim 1, into rows of the transposed cache, dense along dim 0) and
// large enough for 32 x 32 tiles through LDS: blocks behind the rope blocks take one tile each, reading along the source's dense
// direction and writing along the destination's (the element-wise form reads 4-byte values 4 KB apart)
template <bool VT>
__global__ void __launch_bounds__(256)
rope_store_kernel(const RopeStoreArgs g, const uint32_t pair_blocks, const uint32_t tiles0) {
    if (VT && blockIdx.x >= pair_blocks) {
        __shared__ float tile[32][33];
        const uint32_t t = blockIdx.x - pair_blocks, b0 = (t % tiles0) * 32, b1 = (t / tiles0) * 32;
        const uint32_t ne0 = (uint32_t) g.sv.ne[0], ne1 = (uint32_t) g.sv.ne[1];
        const uint32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t i0 = b0 + ty + j * 8, i1 = b1 + tx;
            if (i0 < ne0 && i1 < ne1) tile[ty + j * 8][tx] = *(const float *) (g.v + (size_t) i0 * g.sv.nb[0] + (size_t) i1 * 4);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t i0 = b0 + tx, i1 = b1 + ty + j * 8;
            if (i0 < ne0 && i1 < ne1) *(__half *) (g.vd + (size_t) i0 * 2 + (size_t) i1 * g.svd.nb[1]) = __float2half(tile[tx][ty + j * 8]);
        }
        return;
    }
    uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    if (gid < g.nq) { rope_pair_heads<float>(g.q, g.pos, g.ff, g.qd, g.sq, g.sqd, g.rp, gid); return; }
    gid -= g.nq;
    if (gid < g.nk) { rope_pair_heads<__half>(g.k, g.pos, g.ff, g.kd, g.sk, g.skd, g.rp, gid); return; }
    gid -= g.nk;
    if (!VT && gid < g.nv) st_from_f32<__half>(g.vd + elem_offset(gid, g.svd), ld_as_f32<float>(g.v + elem_offset(gid, g.sv)));
}

// ------------------------------------------------------------------------------------------------ GET_ROWS
// ggml_compute_forward_get_rows (ggml-cpu.c): dst[:, i10, i11, i12] = row src1[i10, i11, i12] of src0[:, :, i11, i12] as f32.
template <typename TS>
__global__ void __launch_bounds__(256)
get_rows_kernel(const char * __restrict__ x, const char * __restrict__ ids, char * __restrict__ y, const Shape sx, const Shape si, const Shape sy) {
    uint32_t i10, i11, i12;
    row_coords(blockIdx.x, (uint32_t) si.ne[0], (uint32_t) si.ne[1], i10, i11, i12);
    const int32_t r = *(const int32_t *) (ids + i10 * si.nb[0] + i11 * si.nb[1] + i12 * si.nb[2]);
    if (r < 0 || r >= sx.ne[1]) return;                      // the CPU asserts; leave the row untouched
    const char * px = x + (size_t) r * sx.nb[1] + i11 * sx.nb[2] + i12 * sx.nb[3];
    float *      py = (float *) (y + i10 * sy.nb[1] + i11 * sy.nb[2] + i12 * sy.nb[3]);
    for (uint32_t i = threadIdx.x; i < (uint32_t) sx.ne[0]; i += 256) py[i] = ld_as_f32<TS>(px + (size_t) i * sizeof(TS));
}
// quantized rows: one unit per thread, the bit-exact unpack of qmm_device.hiph
template <int T>
__global__ void __launch_bounds__(256)
get_rows_q_kernel(const uint8_t * __restrict__ x, const char * __restrict__ ids, char * __restrict__ y, const Shape sx, const Shape si, const Shape sy) {
    uint32_t i10, i11, i12;
    row_coords(blockIdx.x, (uint32_t) si.ne[0], (uint32_t) si.ne[1], i10, i11, i12);
    const int32_t r = *(const int32_t *) (ids + i10 * si.nb[0] + i11 * si.nb[1] + i12 * si.nb[2]);
    if (r < 0 || r >= sx.ne[1]) return;
    const uint8_t * px = x + (size_t) r * sx.nb[1] + i11 * sx.nb[2] + i12 * sx.nb[3];
    float *         py = (float *) (y + i10 * sy.nb[1] + i11 * sy.nb[2] + i12 * sy.nb[3]);
    const int units = (int) (sx.ne[0] / Traits<T>::UNIT_W);
    for (int u = threadIdx.x; u < units; u += 256) {
        Unit<T> un;
        un.load(px, u, (int) sx.ne[0]);
        float out[Traits<T>::UNIT_W];
        un.to_f32(u, out);
#pragma unroll
        for (int rr = 0; rr < Unit<T>::RUNS; ++rr) {
            float * o = py + Unit<T>::k_run(u, rr);
#pragma unroll
            for (int e = 0; e < Unit<T>::RUN_LEN; ++e) o[e] = out[rr * Unit<T>::RUN_LEN + e];
        }
    }
}

// ------------------------------------------------------------------------------------------------ ARGSORT, SUM_ROWS
// The two small ops of the MoE router (build_moe_ffn, src/llama-graph.cpp:842-858: ggml_top_k = argsort + view, ggml_sum_rows
// for the weight normalisation), so that a Mixtral layer stays one split.
// ggml_compute_forward_argsort_f32 (ggml-cpu.c:10199-10236) orders indices by value; here by rank: element i goes to the
// position "number of elements that sort before it" (ties by index), the same permutation whenever the values are distinct.
__global__ void __launch_bounds__(256)
argsort_kernel(const char * __restrict__ x, char * __restrict__ y, const Shape sx, const Shape sy, const int desc) {
    uint32_t i1, i2, i3;
    row_coords(blockIdx.x, (uint32_t) sx.ne[1], (uint32_t) sx.ne[2], i1, i2, i3);
    const float * px = (const float *) (x + i1 * sx.nb[1] + i2 * sx.nb[2] + i3 * sx.nb[3]);
    int32_t *     py = (int32_t *) (y + i1 * sy.nb[1] + i2 * sy.nb[2] + i3 * sy.nb[3]);
    const uint32_t n = (uint32_t) sx.ne[0];
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
        const float v = px[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n; ++j) {
            const float u = px[j];
            rank += (desc ? u > v : u < v) || (u == v && j < i);
        }
        py[rank] = (int32_t) i;
    }
}
// ggml_compute_forward_sum_rows_f32 (ggml-cpu.c): dst[0, i1, i2, i3] = sum over i0; one wave per row
__global__ void __launch_bounds__(256)
sum_rows_kernel(const char * __restrict__ x, char * __restrict__ y, const Shape sx, const Shape sy, const uint32_t rows) {
    const uint32_t row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    uint32_t i1, i2, i3;
    row_coords(row, (uint32_t) sx.ne[1], (uint32_t) sx.ne[2], i1, i2, i3);
    const float * px = (const float *) (x + i1 * sx.nb[1] + i2 * sx.nb[2] + i3 * sx.nb[3]);
    float s = 0.0f;
    for (uint32_t i = lane; i < (uint32_t) sx.ne[0]; i += 64) s += px[i];
    s = wave_sum(s);
    if (lane == 0) *(float *) (y + i1 * sy.nb[1] + i2 * sy.nb[2] + i3 * sy.nb[3]) = s;
}

// ------------------------------------------------------------------------------------------------ MUL_MAT, F16 / F32 src0
// ggml_compute_forward_mul_mat with a non-quantized src0 (ggml-cpu.c:6745-6937): the KQ and KQV products of
// build_attn_mha (src/llama-graph.cpp:1126-1213), whose src0 is a strided view of the F16 KV cache, and small F32 matrices
// (the MoE router).  dst[i13][i12][n][m] = sum_k src0[i13/r3][i12/r2][m][k] * src1[i13][i12][n][k].
//
// F16: as on the CPU (vec_dot_type of F16 is F16: src1 is rounded to f16, products accumulate in f32).  A workgroup of 4 waves
// owns 64 src0 rows x 64 src1 rows; both operands go through LDS as f16 in 32-deep K-steps and each wave drives
// v_mfma_f32_32x32x16_f16 with tokens on the MFMA row index and src0 rows on the lane index, so a lane's 16 results are
// 16 tokens of ONE dst column and every store instruction writes 128 contiguous bytes per half-wave.
struct MmArgs {
    const char * a;  const char * b;  char * d;
    int64_t a_nb1, a_nb2, a_nb3, b_nb1, b_nb2, b_nb3, d_nb1, d_nb2, d_nb3;
    int32_t M, N, K, ne12, r2, r3;
};
constexpr int MM_T = 64, MM_BK = 32, MM_LD = MM_BK + 8;      // LDS row pitch 80 B: 16-byte reads of 32 rows spread over all banks

template <bool VEC>
__global__ void __launch_bounds__(256)
mul_mat_f16_kernel(const MmArgs g) {
    __shared__ __attribute__((aligned(16))) _Float16 sa[MM_T * MM_LD];        // src0 rows (weights side)
    __shared__ __attribute__((aligned(16))) _Float16 sb[MM_T * MM_LD];        // src1 rows (tokens), rounded to f16
    const int i12 = blockIdx.z % g.ne12, i13 = blockIdx.z / g.ne12;
    const char * pa = g.a + (int64_t) (i12 / g.r2) * g.a_nb2 + (int64_t) (i13 / g.r3) * g.a_nb3;
    const char * pb = g.b + (int64_t) i12 * g.b_nb2 + (int64_t) i13 * g.b_nb3;
    char *       pd = g.d + (int64_t) i12 * g.d_nb2 + (int64_t) i13 * g.d_nb3;
    const int m0 = blockIdx.x * MM_T, n0 = blockIdx.y * MM_T;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave & 1) * 32, wn = (wave >> 1) * 32;
    const int lr = tid >> 2, lk = (tid & 3) * 8;                               // staging: row lr, 8 consecutive k from lk
    f32x16v acc = {0};
    for (int k0 = 0; k0 < g.K; k0 += MM_BK) {
        h16x8 va = {0}, vb = {0};
        const int k = k0 + lk;
        if (m0 + lr < g.M) {
            const char * p = pa + (int64_t) (m0 + lr) * g.a_nb1 + (int64_t) k * 2;
            if (VEC && k + 8 <= g.K) {
                va = *(const h16x8 *) p;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (k + e < g.K) va[e] = ((const _Float16 *) p)[e];
            }
        }
        if (n0 + lr < g.N) {
            const float * p = (const float *) (pb + (int64_t) (n0 + lr) * g.b_nb1) + k;
            if (VEC && k + 8 <= g.K) {
                const float4 x = ((const float4 *) p)[0], y = ((const float4 *) p)[1];
                vb = h16x8{ (_Float16) x.x, (_Float16) x.y, (_Float16) x.z, (_Float16) x.w, (_Float16) y.x, (_Float16) y.y, (_Float16) y.z, (_Float16) y.w };
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) if (k + e < g.K) vb[e] = (_Float16) p[e];
            }
        }
        __syncthreads();                                                        // previous K-step's fragment reads are done
        *(h16x8 *) &sa[lr * MM_LD + lk] = va;
        *(h16x8 *) &sb[lr * MM_LD + lk] = vb;
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < MM_BK; kk += 16) {
            const h16x8 fa = *(const h16x8 *) &sb[(wn + (lane & 31)) * MM_LD + kk + (lane >> 5) * 8];     // tokens: MFMA rows
            const h16x8 fb = *(const h16x8 *) &sa[(wm + (lane & 31)) * MM_LD + kk + (lane >> 5) * 8];     // src0 rows: MFMA columns
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0);
        }
    }
    const int m = m0 + wm + (lane & 31);
    if (m < g.M) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = n0 + wn + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
            if (n < g.N) *(float *) (pd + (int64_t) n * g.d_nb1 + (int64_t) m * 4) = acc[r];
        }
    }
}

// The router of a mixture-of-experts block after its logits (build_moe_ffn, src/llama-graph.cpp:818-858), one launch instead of
// five: probs = soft_max(logits); ids = argsort(probs, descending) (top-k = its first n_used entries); weights = probs[ids[:n_used]]
// normalised by their sum (ggml_get_rows, ggml_sum_rows, ggml_div).  One wave per token, lane e holds expert e (n_expert <= 64).
// (the body: one wave, lane e holds the logit of expert e; shared by moe_router_kernel and moe_router_logits_kernel)
__device__ __forceinline__ void router_of_logits(const float x, const int lane, const int n_expert, const int n_used, const int normalise,
                                                 int32_t * __restrict__ ids_row, float * __restrict__ weights_row) {
    const float mx = wave_max(x);
    const float e = lane < n_expert ? expf(x - mx) : 0.0f;
    // soft_max_wave_kernel's arithmetic AND its order of additions: there lane g holds four consecutive values and adds them left to
    // right before the wave reduction; a sum taken in any other order may differ in the last bit, and a last bit of a router weight is
    // enough to flip an int8 rounding one layer further down (round 3: `llama-e2e layers` showed the one-launch router and the five
    // per-node launches 2e-2 of an rms apart on one element of a later layer; the dense models agreed bit for bit)
    const float e0 = __shfl(e, (4 * lane) & 63, 64), e1 = __shfl(e, (4 * lane + 1) & 63, 64), e2 = __shfl(e, (4 * lane + 2) & 63, 64), e3 = __shfl(e, (4 * lane + 3) & 63, 64);
    const float g4 = lane < 16 ? ((e0 + e1) + e2) + e3 : 0.0f;
    const float p = e * (1.0f / wave_sum(g4));                     // exp(x - max) * (1 / sum)
    int rank = 0;
    for (int j = 0; j < n_expert; ++j) {
        const float u = __shfl(p, j, 64);
        rank += u > p || (u == p && j < lane);
    }
    if (lane < n_expert) ids_row[rank] = lane;
    const bool sel = lane < n_expert && rank < n_used;
    // sum_rows_kernel adds the selected weights with the weight of rank r on lane r: the same placement here
    float byrank = 0.0f;
    for (int r = 0; r < n_used; ++r) {
        const int src = __ffsll((long long) __ballot(lane < n_expert && rank == r)) - 1;     // (every rank below n_expert has exactly one owner)
        const float v = __shfl(p, src & 63, 64);
        if (lane == r) byrank = v;
    }
    const float sum = wave_sum(byrank);
    if (sel) weights_row[rank] = normalise ? p / sum : p;
}
__global__ void __launch_bounds__(256)
moe_router_kernel(const char * __restrict__ logits, char * __restrict__ ids, char * __restrict__ weights, const int64_t l_nb1, const int64_t i_nb1,
                  const int64_t w_nb1, const int n_expert, const int n_used, const int n_tokens, const int normalise) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= n_tokens) return;
    const float x = lane < n_expert ? ((const float *) (logits + (int64_t) t * l_nb1))[lane] : -INFINITY;
    router_of_logits(x, lane, n_expert, n_used, normalise, (int32_t *) (ids + (int64_t) t * i_nb1), (float *) (weights + (int64_t) t * w_nb1));
}

// The router WITH its logits for a few tokens (round 3: 32 of Mixtral's ~390 launches per generated token): logits = gate_inp x
// (F32 [K, n_expert], build_moe_ffn's first MUL_MAT) and everything moe_router_kernel does, one workgroup per token.  Eight groups of
// DOT_T = 128 threads take eight experts side by side, each group exactly as mul_mat_dot_block_kernel takes one dst element (the same
// shares, the same wave reduction, the same order of the wave sums), so the logits row it writes is the per-node kernel's bit for
// bit; wave 0 then routes the row from LDS.  (First version: four groups of 256, two rounds for Mixtral's eight experts: the launch
// then took as long as the two it replaced, gpu_ms_per_token 2.94 -> 2.93.)
__device__ __forceinline__ float dot_f32_share(const float * __restrict__ pa, const float * __restrict__ pb, const int K, const int t);
// NORM (round 3): x is the un-normed row; the workgroup forms rms_norm(x) * norm_w itself (rms_norm_vec_kernel<true, false, 1024>'s arithmetic and order of
// additions, so the bits of the separate launch), stores it to `normed` for the expert MUL_MAT_IDs and dots the logits against the copy in LDS:
// build_moe_ffn's ffn_norm launch (4.75 us of a Mixtral layer's ~84) disappears.  K <= 16384, K % 4 == 0, 16-byte aligned rows.
template <bool NORM>
__global__ void __launch_bounds__(1024)
moe_router_logits_kernel(const char * __restrict__ wgt, const char * __restrict__ x, char * __restrict__ logits, char * __restrict__ ids,
                         char * __restrict__ weights, const int64_t a_nb1, const int64_t b_nb1, const int64_t l_nb1, const int64_t i_nb1, const int64_t w_nb1,
                         const int K, const int n_expert, const int n_used, const int normalise,
                         const float * __restrict__ norm_w = nullptr, const float eps = 0.0f, char * __restrict__ normed = nullptr, const int64_t y_nb1 = 0) {
    constexpr int NG = 1024 / DOT_T, WPG = DOT_T / 64;     // groups per workgroup (experts side by side), waves per group
    extern __shared__ __attribute__((aligned(16))) float xs[];
    __shared__ float red[17];
    __shared__ float lg[64];
    const int n = blockIdx.x, tid = threadIdx.x, grp = tid / DOT_T, tg = tid % DOT_T, lane = tid & 63, wg = (tid >> 6) % WPG;
    const float * pb = (const float *) (x + (int64_t) n * b_nb1);
    if (NORM) {
        const float4 * px = (const float4 *) pb;
        float4 *       py = (float4 *) (normed + (int64_t) n * y_nb1);
        const uint32_t n4 = (uint32_t) K / 4;
        float4 v[4], wv[4];
        float sum = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t i = tid + j * 1024;
            v[j]  = i < n4 ? px[i] : make_float4(0, 0, 0, 0);
            wv[j] = i < n4 ? ((const float4 *) norm_w)[i] : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) sum += v[j].x * v[j].x + v[j].y * v[j].y + v[j].z * v[j].z + v[j].w * v[j].w;
        const float scale = rms_scale_block<1024>(sum, red, (float) K, eps);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t i = tid + j * 1024;
            if (i < n4) {
                float4 r = make_float4(v[j].x * scale, v[j].y * scale, v[j].z * scale, v[j].w * scale);
                r.x *= wv[j].x; r.y *= wv[j].y; r.z *= wv[j].z; r.w *= wv[j].w;
                py[i] = r;
                ((float4 *) xs)[i] = r;
            }
        }
        __syncthreads();
        pb = xs;
    }
    for (int e0 = 0; e0 < n_expert; e0 += NG) {              // uniform trip count: every thread meets every barrier
        const int e = e0 + grp, ec = e < n_expert ? e : n_expert - 1;
        float s = dot_f32_share((const float *) (wgt + (int64_t) ec * a_nb1), pb, K, tg);
        s = wave_sum(s);
        __syncthreads();
        if (lane == 0) red[grp * WPG + wg] = s;
        __syncthreads();
        float r = red[grp * WPG];
        for (int i = 1; i < WPG; ++i) r += red[grp * WPG + i];
        if (e < n_expert && tg == 0) { lg[e] = r; ((float *) (logits + (int64_t) n * l_nb1))[e] = r; }
    }
    __syncthreads();
    if (tid < 64)
        router_of_logits(lane < n_expert ? lg[lane] : -INFINITY, lane, n_expert, n_used, normalise, (int32_t *) (ids + (int64_t) n * i_nb1),
                         (float *) (weights + (int64_t) n * w_nb1));
}

// The other end of a mixture-of-experts block (build_moe_ffn, src/llama-graph.cpp:896-911): experts * weights, then the sum over the
// used experts through 2-D views, as one launch: out[n][c] = ((x[n][0][c] w[n][0] + x[n][1][c] w[n][1]) + ...), in the graph's order.
// (four columns of one token: ((x0 w0) + x1 w1) + ..., shared by moe_combine_kernel and moe_combine_add_norm_kernel)
__device__ __forceinline__ float4 moe_combine4(const char * __restrict__ px, const char * __restrict__ pw, const int64_t x_nb1, const int64_t w_nb1, const int U) {
    float4 acc = *(const float4 *) px;
    {
        const float w0 = *(const float *) pw;
        acc.x *= w0; acc.y *= w0; acc.z *= w0; acc.w *= w0;
    }
    for (int u = 1; u < U; ++u) {
        const float4 v = *(const float4 *) (px + (int64_t) u * x_nb1);
        const float wu = *(const float *) (pw + (int64_t) u * w_nb1);
        acc.x += v.x * wu; acc.y += v.y * wu; acc.z += v.z * wu; acc.w += v.w * wu;
    }
    return acc;
}
__global__ void __launch_bounds__(256)
moe_combine_kernel(const char * __restrict__ x, const char * __restrict__ w, char * __restrict__ out, const int64_t x_nb1, const int64_t x_nb2,
                   const int64_t w_nb1, const int64_t w_nb2, const int64_t o_nb1, const int E, const int U) {
    const int n = blockIdx.y;
    const int c = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (c >= E) return;
    *(float4 *) (out + (int64_t) n * o_nb1 + (int64_t) c * 4) = moe_combine4(x + (int64_t) n * x_nb2 + (int64_t) c * 4, w + (int64_t) n * w_nb2, x_nb1, w_nb1, U);
}

// The block's tail with what follows it in the layer (round 3): out = combine(x, w) as above, sum = out + b (the residual: l_out),
// y = rms_norm(sum) * nw (the next attn_norm, or result_norm): moe_combine_kernel's and rms_norm_vec_kernel<true, true, NT>'s arithmetic
// in that kernel's partition and order of additions, so both results are the two launches' bit for bit.
template <int NT>
__global__ void __launch_bounds__(NT)
moe_combine_add_norm_kernel(const char * __restrict__ x, const char * __restrict__ wts, const char * __restrict__ b, const float * __restrict__ nw,
                            char * __restrict__ y, char * __restrict__ s, const int64_t x_nb1, const int64_t x_nb2, const int64_t w_nb1, const int64_t w_nb2,
                            const int64_t b_nb1, const int64_t y_nb1, const int64_t s_nb1, const int E, const int U, const float eps) {
    __shared__ float red[17];
    const int n = blockIdx.x;
    const float4 * pb = (const float4 *) (b + (int64_t) n * b_nb1);
    float4 *       py = (float4 *) (y + (int64_t) n * y_nb1);
    float4 *       ps = (float4 *) (s + (int64_t) n * s_nb1);
    const uint32_t n4 = (uint32_t) E / 4;
    float4 v[4], wv[4];
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t i = threadIdx.x + j * NT;
        v[j] = i < n4 ? moe_combine4(x + (int64_t) n * x_nb2 + (int64_t) i * 16, wts + (int64_t) n * w_nb2, x_nb1, w_nb1, U) : make_float4(0, 0, 0, 0);
        wv[j] = i < n4 ? ((const float4 *) nw)[i] : make_float4(0, 0, 0, 0);
        if (i < n4) {
            const float4 t = pb[i];
            v[j].x += t.x; v[j].y += t.y; v[j].z += t.z; v[j].w += t.w;
        }
    }
    // every input of this token is in registers before the first output is stored: with ONE token (one workgroup: token generation) the
    // two results may therefore lie on ANY of the inputs, which is where ggml-alloc puts them in llama.cpp's graphs (l_out on the
    // dead router weights, the normed row on the residual); with more tokens the caller keeps them clear of what other workgroups read
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t i = threadIdx.x + j * NT;
        if (i < n4) ps[i] = v[j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) sum += v[j].x * v[j].x + v[j].y * v[j].y + v[j].z * v[j].z + v[j].w * v[j].w;
    const float scale = rms_scale_block<NT>(sum, red, (float) E, eps);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t i = threadIdx.x + j * NT;
        if (i < n4) {
            float4 r = make_float4(v[j].x * scale, v[j].y * scale, v[j].z * scale, v[j].w * scale);
            r.x *= wv[j].x; r.y *= wv[j].y; r.z *= wv[j].z; r.w *= wv[j].w;
            py[i] = r;
        }
    }
}

// Few outputs with a long K (the MoE router at batch 1: 8 x 4096): one WORKGROUP per dst element, so K is spread over DOT_T = 128
// threads (256 until round 3: eight experts then did not fit one workgroup of moe_router_logits_kernel side by side) instead of 64 (the wave-per-element kernel walks K = 4096 in 64 dependent trips: 27 us per call, 0.87 ms per Mixtral token).
// (thread t of DOT_T's share of the dot product; shared with moe_router_logits_kernel so that both give the same bits)
__device__ __forceinline__ float dot_f32_share(const float * __restrict__ pa, const float * __restrict__ pb, const int K, const int t) {
    float s = 0.0f;
    if ((((uintptr_t) pa | (uintptr_t) pb) & 15) == 0) {
        const int k4 = K / 4;
        for (int k = t; k < k4; k += DOT_T) {
            const float4 x = ((const float4 *) pa)[k], y = ((const float4 *) pb)[k];
            s += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
        }
        for (int k = k4 * 4 + t; k < K; k += DOT_T) s += pa[k] * pb[k];
    } else {
        for (int k = t; k < K; k += DOT_T) s += pa[k] * pb[k];
    }
    return s;
}
__global__ void __launch_bounds__(DOT_T)
mul_mat_dot_block_kernel(const MmArgs g) {
    __shared__ float red[4];
    const int i12 = blockIdx.z % g.ne12, i13 = blockIdx.z / g.ne12;
    const int m = blockIdx.x % g.M, n = blockIdx.x / g.M;
    const float * pa = (const float *) (g.a + (int64_t) (i12 / g.r2) * g.a_nb2 + (int64_t) (i13 / g.r3) * g.a_nb3 + (int64_t) m * g.a_nb1);
    const float * pb = (const float *) (g.b + (int64_t) i12 * g.b_nb2 + (int64_t) i13 * g.b_nb3 + (int64_t) n * g.b_nb1);
    float s = dot_f32_share(pa, pb, g.K, threadIdx.x);
    s = block_reduce<false>(s, red);
    if (threadIdx.x == 0) *(float *) (g.d + (int64_t) i12 * g.d_nb2 + (int64_t) i13 * g.d_nb3 + (int64_t) n * g.d_nb1 + (int64_t) m * 4) = s;
}

// F32 src0 (small matrices such as the MoE router ffn_gate_inp): one wave per dst element, f32 FMA, lanes stride K.
template <typename TA>
__global__ void __launch_bounds__(256)
mul_mat_dot_kernel(const MmArgs g) {
    const int i12 = blockIdx.z % g.ne12, i13 = blockIdx.z / g.ne12;
    const int64_t e = (int64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= (int64_t) g.M * g.N) return;
    const int m = (int) (e % g.M), n = (int) (e / g.M), lane = threadIdx.x & 63;
    const char *  pa = g.a + (int64_t) (i12 / g.r2) * g.a_nb2 + (int64_t) (i13 / g.r3) * g.a_nb3 + (int64_t) m * g.a_nb1;
    const float * pb = (const float *) (g.b + (int64_t) i12 * g.b_nb2 + (int64_t) i13 * g.b_nb3 + (int64_t) n * g.b_nb1);
    float s = 0.0f;
    for (int k = lane; k < g.K; k += 64) s += ld_as_f32<TA>(pa + (int64_t) k * sizeof(TA)) * pb[k];
    s = wave_sum(s);
    if (lane == 0) *(float *) (g.d + (int64_t) i12 * g.d_nb2 + (int64_t) i13 * g.d_nb3 + (int64_t) n * g.d_nb1 + (int64_t) m * 4) = s;
}

// ------------------------------------------------------------------------------------------------ attention, few tokens
// The chain build_attn_mha emits without flash attention (src/llama-graph.cpp:1166-1203),
//     kq = mul_mat(k, q);  p = soft_max_ext(kq, mask, scale);  kqv = mul_mat(v, p);  cont(permute(kqv, 0, 2, 1, 3))
// for a batch of a few tokens (token generation) as ONE launch instead of four: a workgroup owns one (head, token), keeps the
// n_kv scores in LDS and writes its head's slice of the merged output row.  Arithmetic as the CPU path has it: q and p are
// rounded to f16 (vec_dot_type of an F16 src0 is F16, ggml-cpu.c type_traits_cpu), products accumulate in f32, the softmax
// is the one of soft_max_kernel.  K rows are [D] f16 (dense), V is the transposed cache: row d holds n_kv f16.
struct AttnArgs {
    const char * q; const char * k; const char * v; const char * mask; char * dst;
    int64_t q_nb1, q_nb2, k_nb1, k_nb2, v_nb1, v_nb2, m_nb1, d_nb1;
    int32_t D, Dv, n_kv, H, gqa;
    float   scale;
};
// 16 waves per workgroup and several independent loads in flight per lane: with one (head, token) per workgroup the kernel is
// bound by load latency, not bandwidth (the first version, 4 waves and one row per lane group at a time, took 22 us at
// n_kv = 640; the four separate launches it replaces took 16).
// FRESH: the batch's own K / V rows are not in the cache yet — the launch also does what precedes the attention in build_attn
// (src/llama-graph.cpp:1306-1365): rope(q), rope(k) -> K cache, v -> V cache.  q arrives un-roped; every workgroup ropes the N
// new K rows of its kv head and converts the N new V rows into LDS and takes them from there for cache positions
// j0 .. j0 + N (no workgroup reads those positions from memory, so the one workgroup per kv head that also stores them races
// with nobody).  Normal-mode RoPE over the whole head (n_dims == D).
struct AttnFresh {
    const char * kraw; const char * vraw; char * kd; char * vd; const int32_t * pos; const float * ff;
    int64_t kraw_nbh, kraw_nbn, vraw_nbn, kd_nbh, kd_nbn, vd_nbc;
    RopeParams rp;
    int32_t N, j0;
};
#ifdef ATTN_STAMPS                      // development builds only (profiles/tools/attn_dev.hip): 100 MHz stamps of workgroup phases
__device__ unsigned long long attn_stamps[64][8];
#define ATTN_STAMP(slot) do { if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 64) attn_stamps[blockIdx.x][slot] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ATTN_STAMP(slot) do { } while (0)
#endif
template <int D, bool FRESH>
__global__ void __launch_bounds__(1024)
attn_decode_kernel(const AttnArgs g, const AttnFresh f) {
    extern __shared__ float sc[];                       // n_kv scores, then probabilities [FRESH: + q row f32, new K rows, new V rows f16]
    __shared__ float red[16];
    ATTN_STAMP(0);
    // (all kernel arguments in one round trip: see attn_decode_short_kernel)
    asm volatile("" :: "s"(g.q), "s"(g.k), "s"(g.v), "s"(g.mask), "s"(g.q_nb1), "s"(g.q_nb2), "s"(g.k_nb1), "s"(g.k_nb2), "s"(g.v_nb1), "s"(g.v_nb2), "s"(g.m_nb1),
                 "s"(g.n_kv), "s"(g.gqa), "s"(g.Dv));
    if (FRESH) asm volatile("" :: "s"(f.kraw), "s"(f.vraw), "s"(f.pos), "s"(f.ff), "s"(f.kraw_nbh), "s"(f.kraw_nbn), "s"(f.vraw_nbn), "s"(f.N));
    const int h = blockIdx.x, n = blockIdx.y, hk = h / g.gqa;
    const int tid = threadIdx.x, l8 = tid & 7, grp = tid >> 3;          // 128 groups of 8 lanes: one K row per group
    constexpr int CH = D / 8;                           // halves of a K row per lane
    constexpr int NV = CH / 8;                          // 16-byte loads per lane and row
    float *    qs   = sc + g.n_kv;                      // FRESH only
    _Float16 * knew = reinterpret_cast<_Float16 *>(qs + D);
    _Float16 * vnew = knew + (FRESH ? f.N * D : 0);
    const float * pm = (const float *) (g.mask + (int64_t) n * g.m_nb1);
    const char *  pk = g.k + (int64_t) hk * g.k_nb2 + (int64_t) l8 * CH * 2;
    // (measured and dropped: all of a group's K rows and all of a wave's V rows requested up front, 64 + 64 VGPRs: tg128 403 -> 395)
    // two rows (jt, jt + 128) per trip, their loads issued together and one trip ahead of the arithmetic; the first trip is
    // requested before anything else, so with FRESH the cache rows are already on their way while q / k / v are prepared
    auto load_trip = [&](int jt, h16x8 (&kv)[2][NV], float (&mk)[2]) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int j = jt + r * 128, jc = j < g.n_kv ? j : jt;
            const h16x8 * row = (const h16x8 *) (pk + (int64_t) jc * g.k_nb1);
#pragma unroll
            for (int c = 0; c < NV; ++c) kv[r][c] = row[c];
            mk[r] = pm[jc];
        }
    };
    h16x8 kv[2][NV];
    float mk[2];
    if (grp < g.n_kv) load_trip(grp, kv, mk);
    float qf[CH];
    if (FRESH) {
        const bool writer = h % g.gqa == 0 && n == 0;
        for (int idx = tid; idx < f.N * (D / 2); idx += 1024) {
            const int n2 = idx / (D / 2), p = idx % (D / 2);
            const float * kr = (const float *) (f.kraw + (int64_t) hk * f.kraw_nbh + (int64_t) n2 * f.kraw_nbn);
            float c, s;
            rope_cs((float) f.pos[n2], (uint32_t) p, f.rp, f.ff, c, s);
            const float x0 = kr[2 * p], x1 = kr[2 * p + 1];
            const _Float16 y0 = (_Float16) (x0 * c - x1 * s), y1 = (_Float16) (x0 * s + x1 * c);
            knew[n2 * D + 2 * p] = y0;
            knew[n2 * D + 2 * p + 1] = y1;
            if (writer) {
                _Float16 * kd = reinterpret_cast<_Float16 *>(f.kd + (int64_t) hk * f.kd_nbh + (int64_t) n2 * f.kd_nbn);
                kd[2 * p] = y0;
                kd[2 * p + 1] = y1;
            }
        }
        // the three preparations start on different waves (for one token: K on waves 0, V on waves 8-9, q on wave 4), so their
        // load -> sincos -> store chains overlap instead of queueing on the same threads
        for (int idx = (tid + 512) & 1023; idx < f.N * g.Dv; idx += 1024) {
            const int n2 = idx / g.Dv, d = idx % g.Dv, ch = hk * g.Dv + d;
            const _Float16 v = (_Float16) *(const float *) (f.vraw + (int64_t) ch * 4 + (int64_t) n2 * f.vraw_nbn);
            vnew[n2 * g.Dv + d] = v;
            if (writer) *reinterpret_cast<_Float16 *>(f.vd + (int64_t) n2 * 2 + (int64_t) ch * f.vd_nbc) = v;
        }
        if (const int t = tid - 256; t >= 0 && t < D / 2) {
            const float * pq = (const float *) (g.q + (int64_t) n * g.q_nb1 + (int64_t) h * g.q_nb2);
            float c, s;
            rope_cs((float) f.pos[n], (uint32_t) t, f.rp, f.ff, c, s);
            const float x0 = pq[2 * t], x1 = pq[2 * t + 1];
            qs[2 * t] = x0 * c - x1 * s;
            qs[2 * t + 1] = x0 * s + x1 * c;
        }
        ATTN_STAMP(1);
        __syncthreads();
        ATTN_STAMP(2);
#pragma unroll
        for (int e = 0; e < CH; ++e) qf[e] = (float) (_Float16) qs[l8 * CH + e];
    } else {
        const float * pq = (const float *) (g.q + (int64_t) n * g.q_nb1 + (int64_t) h * g.q_nb2) + l8 * CH;
#pragma unroll
        for (int e = 0; e < CH; ++e) qf[e] = (float) (_Float16) pq[e];
    }
    float mx = -INFINITY;
    for (int jt = grp; jt < g.n_kv; jt += 256) {
        h16x8 kvn[2][NV];
        float mkn[2];
        const bool more = jt + 256 < g.n_kv;
        if (more) load_trip(jt + 256, kvn, mkn);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int j = jt + r * 128;
            if (FRESH && (unsigned) (j - f.j0) < (unsigned) f.N) {
#pragma unroll
                for (int c = 0; c < NV; ++c) kv[r][c] = *(const h16x8 *) &knew[(j - f.j0) * D + l8 * CH + c * 8];
            }
            float s = 0.0f;
#pragma unroll
            for (int c = 0; c < NV; ++c)
#pragma unroll
                for (int e = 0; e < 8; ++e) s += (float) kv[r][c][e] * qf[c * 8 + e];
            s = sum8(s);
            s = s * g.scale + mk[r];
            if (j < g.n_kv) {
                if (l8 == 0) sc[j] = s;
                mx = fmaxf(mx, s);
            }
        }
        if (more) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
#pragma unroll
                for (int c = 0; c < NV; ++c) kv[r][c] = kvn[r][c];
                mk[r] = mkn[r];
            }
        }
    }
    if (g.n_kv <= 512) {
        // Short caches (token generation from an empty context: n_kv = 256): the kernel is a chain of latencies, not of bytes (12.9 us
        // per layer for 64 KB of K and V), so the softmax is done by EVERY wave for itself from the scores in LDS: one barrier instead
        // of seven, no block reductions, and the wave's eight V rows are requested together instead of in two trips.  A lane owns the
        // eight columns it multiplies (j = 8 lane ...).  Same max, same exponentials; the sum runs in another order than below.
        ATTN_STAMP(3);
        __syncthreads();                                // sc[] complete
        ATTN_STAMP(4);
        const int lane = tid & 63, wave = tid >> 6;
        const int jl = lane * 8;
        const bool live = jl < g.n_kv;                  // (n_kv is a multiple of 8: qmm_attn_decode_supported)
        const int jc = live ? jl : 0;
        const char * pv = g.v + (int64_t) hk * g.v_nb2;
        float * out = (float *) (g.dst + (int64_t) n * g.d_nb1) + (int64_t) h * g.Dv;
        h16x8 vv[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {                   // rows wave + 16 r of the first 128; requested before the softmax arithmetic
            const int d = wave + 16 * r < g.Dv ? wave + 16 * r : wave;
            vv[r] = *(const h16x8 *) (pv + (int64_t) d * g.v_nb1 + (int64_t) jc * 2);
        }
        float sv[8], m = -INFINITY;
#pragma unroll
        for (int e = 0; e < 8; ++e) { sv[e] = live ? sc[jl + e] : -INFINITY; m = fmaxf(m, sv[e]); }
        m = wave_max(m);
        float sum = 0.0f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { sv[e] = live ? expf(sv[e] - m) : 0.0f; sum += sv[e]; }
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
        float pr[8], pfresh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) pr[e] = (float) (_Float16) (sv[e] * inv);
        ATTN_STAMP(5);
        if (FRESH) {                                    // the batch's own positions: probability set aside, column weight 0 (see below)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                pfresh[i] = i < f.N ? (float) (_Float16) (expf(sc[f.j0 + i] - m) * inv) : 0.0f;
#pragma unroll
                for (int e = 0; e < 8; ++e) if (i < f.N && jl + e == f.j0 + i) pr[e] = 0.0f;
            }
        }
        for (int d0 = wave; d0 < g.Dv; d0 += 128) {
            if (d0 != wave) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int d = d0 + 16 * r < g.Dv ? d0 + 16 * r : d0;
                    vv[r] = *(const h16x8 *) (pv + (int64_t) d * g.v_nb1 + (int64_t) jc * 2);
                }
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                float acc = 0.0f;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc += (float) vv[r][e] * pr[e];
                float t = wave_sum(acc);
                const int d = d0 + 16 * r;
                if (lane == 0 && d < g.Dv) {
                    if (FRESH) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) if (i < f.N) t += pfresh[i] * (float) vnew[i * g.Dv + d];
                    }
                    out[d] = t;
                }
            }
        }
        ATTN_STAMP(6);
        return;
    }
    mx = block_reduce<true>(mx, red);                   // its barriers also publish sc[]
    float sum = 0.0f;
    for (int j = tid; j < g.n_kv; j += 1024) {
        const float e = expf(sc[j] - mx);
        sc[j] = e;
        sum += e;
    }
    sum = block_reduce<false>(sum, red);
    const float inv = 1.0f / sum;
    for (int j = tid; j < g.n_kv; j += 1024) sc[j] = (float) (_Float16) (sc[j] * inv);
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    float * out = (float *) (g.dst + (int64_t) n * g.d_nb1) + (int64_t) h * g.Dv;
    const char * pv = g.v + (int64_t) hk * g.v_nb2;
    // FRESH: the probabilities of the batch's own positions are set aside and zeroed in sc[], so the main loop runs over the
    // (stale, finite) cache values of those columns with weight 0 and the new V rows come in as a rank-N update at the end
    float pfresh[8];
    if (FRESH) {
#pragma unroll
        for (int i = 0; i < 8; ++i) pfresh[i] = i < f.N ? sc[f.j0 + i] : 0.0f;
        __syncthreads();
        if (tid < f.N) sc[f.j0 + tid] = 0.0f;
        __syncthreads();
    }
    for (int d0 = wave; d0 < g.Dv; d0 += 64) {          // four V rows (d0, +16, +32, +48) per trip
        float acc[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
        for (int j = lane * 8; j < g.n_kv; j += 1024) {   // two column blocks (j, j + 512) per pass: eight loads requested together
            h16x8 vv[2][4];
            const int j2 = j + 512 < g.n_kv ? j + 512 : j;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int d = d0 + 16 * r < g.Dv ? d0 + 16 * r : d0;
                vv[0][r] = *(const h16x8 *) (pv + (int64_t) d * g.v_nb1 + (int64_t) j * 2);
                vv[1][r] = *(const h16x8 *) (pv + (int64_t) d * g.v_nb1 + (int64_t) j2 * 2);
            }
            float p[2][8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { p[0][e] = sc[j + e]; p[1][e] = j + 512 < g.n_kv ? sc[j2 + e] : 0.0f; }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[r] += (float) vv[0][r][e] * p[0][e] + (float) vv[1][r][e] * p[1][e];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float t = wave_sum(acc[r]);
            const int d = d0 + 16 * r;
            if (lane == 0 && d < g.Dv) {
                if (FRESH) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) if (i < f.N) t += pfresh[i] * (float) vnew[i * g.Dv + d];
                }
                out[d] = t;
            }
        }
    }
}

// Short caches (n_kv <= 1024 in one or two halves of 512 columns, D and Dv <= 128: token generation inside llama.cpp's first windows, where the launch is a chain of
// latencies: 64 KB of K and V per head).  Round 3 stamps of the kernel above at n_kv = 256 (profiles/tools/attn_dev.hip): 2.9 us until
// the new rows are roped and in LDS, 0.9 us scores, 0.8 us softmax, 2.6 us for the V product (its loads requested behind the second
// barrier), 7.9 us in the kernel for a 10.5 us slot.  This one
//   * requests EVERYTHING at entry: the thread's share of the new q / k / v values first, then its four K rows, masks and its
//     eight V rows (the V loads were 1.5 us of exposed latency behind the softmax), and only then starts to compute;
//   * spreads the preparation of one token over waves 0-3, one per SIMD (q pairs, K pairs, V values; they were waves 0, 4 and 8:
//     the same SIMD);
//   * takes one exponential per column (thread j, through LDS) instead of eight per lane in every wave;
//   * multiplies f16 x f16 into f32 directly (v_fma_mix: the products of the conversions, exactly as before);
//   * folds the eight V-row sums of a wave into one register with v_permlane32_swap / v_permlane16_swap before the DPP steps
//     (20 instructions instead of 8 wave_sums).
// Same arithmetic as the CPU chain (q, p rounded to f16, f32 accumulation, soft_max_kernel's max / exp / sum); sums in another order.
template <int D, bool FRESH, int WIDTH>
__global__ void __launch_bounds__(1024)
attn_decode_short_kernel(const AttnArgs g, const AttnFresh f) {
    extern __shared__ float sc[];       // n_kv scores | n_kv exponentials [FRESH: | q row f16 (D halves, D floats reserved) | new K rows f16 | new V rows f16]
    ATTN_STAMP(0);
    // every kernel argument the address arithmetic below needs is requested HERE, in one round trip: left to itself hipcc issues the
    // s_loads where the values are first used, three dependent waits (1.2 us from entry to the first global load, stamped)
    asm volatile("" :: "s"(g.q), "s"(g.k), "s"(g.v), "s"(g.mask), "s"(g.q_nb1), "s"(g.q_nb2), "s"(g.k_nb1), "s"(g.k_nb2), "s"(g.v_nb1), "s"(g.v_nb2), "s"(g.m_nb1),
                 "s"(g.n_kv), "s"(g.gqa), "s"(g.Dv));
    if (FRESH) asm volatile("" :: "s"(f.kraw), "s"(f.vraw), "s"(f.pos), "s"(f.ff), "s"(f.kraw_nbh), "s"(f.kraw_nbn), "s"(f.vraw_nbn), "s"(f.N));
    const int h = blockIdx.x, n = blockIdx.y, hk = h / g.gqa;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l8 = tid & 7, grp = tid >> 3;
    constexpr int CH = D / 8, NV = CH / 8, HP = D / 2;
    static_assert(WIDTH == 256 || WIDTH == 512 || WIDTH == 1024, "cache columns this instantiation covers");
    constexpr int NR = WIDTH == 256 ? 2 : 4;            // K rows per 8-lane group and trip (n_kv <= 256: two)
    constexpr int CPL = WIDTH == 256 ? 4 : 8;           // cache columns per lane and trip in the softmax and the V product
    constexpr int TRIPS = WIDTH == 1024 ? 2 : 1;        // n_kv <= 1024: the cache in two halves of 512 columns through the same registers
    typedef _Float16 hcol __attribute__((ext_vector_type(CPL)));
    float *    ex   = sc + g.n_kv;
    _Float16 * qs   = reinterpret_cast<_Float16 *>(ex + g.n_kv);
    _Float16 * knew = qs + 2 * D;
    _Float16 * vnew = knew + (FRESH ? f.N * D : 0);
    // ---- requests.  FRESH: item `it` of the batch's preparation: q pairs [0, HP), K pairs (HP per token), then V values
    const int nk = FRESH ? f.N * HP : 0, items = FRESH ? HP + nk + f.N * g.Dv : 0;
    auto item = [&](int it, const float *& p0, const float *& p1, int & tok) {
        if (it < HP) {
            const float * pq = (const float *) (g.q + (int64_t) n * g.q_nb1 + (int64_t) h * g.q_nb2);
            p0 = pq + 2 * it; p1 = p0 + 1; tok = n;
        } else if (it < HP + nk) {
            const int n2 = (it - HP) / HP, p = (it - HP) % HP;
            const float * kr = (const float *) (f.kraw + (int64_t) hk * f.kraw_nbh + (int64_t) n2 * f.kraw_nbn);
            p0 = kr + 2 * p; p1 = p0 + 1; tok = n2;
        } else {
            const int n2 = (it - HP - nk) / g.Dv, d = (it - HP - nk) % g.Dv;
            p0 = p1 = (const float *) (f.vraw + (int64_t) (hk * g.Dv + d) * 4 + (int64_t) n2 * f.vraw_nbn); tok = n2;
        }
    };
    float x0 = 0.0f, x1 = 0.0f, xff = 1.0f;
    int32_t xpos = 0;
    if (FRESH) {
        const float * p0, * p1; int tok;
        const int it0 = tid < items ? tid : 0;
        item(it0, p0, p1, tok);
        const float * pf = f.ff ? f.ff + (it0 < HP ? it0 : it0 < HP + nk ? (it0 - HP) % HP : 0) : p0;     // the pair's frequency factor
        // asm: as C++ loads hipcc sinks these four into the branches that use them, BEHIND the cache loads below (vmcnt retires in
        // order), and as volatile loads it waits for each.  The wait that releases them counts the cache loads: see below
        const int32_t * pp = f.pos + tok;
        asm volatile("global_load_dword %0, %4, off\n\tglobal_load_dword %1, %5, off\n\tglobal_load_dword %2, %6, off\n\tglobal_load_dword %3, %7, off"
                     : "=&v"(xpos), "=&v"(x0), "=&v"(x1), "=&v"(xff) : "v"(pp), "v"(p0), "v"(p1), "v"(pf) : "memory");
    }
    h16x8 kv[NR][NV];
    float mk[NR];
    {
        const float * pm = (const float *) (g.mask + (int64_t) n * g.m_nb1);
        const char *  pk = g.k + (int64_t) hk * g.k_nb2 + (int64_t) l8 * CH * 2;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int j = grp + 128 * r, jc = j < g.n_kv ? j : g.n_kv - 1;
            const h16x8 * row = (const h16x8 *) (pk + (int64_t) jc * g.k_nb1);
#pragma unroll
            for (int c = 0; c < NV; ++c) kv[r][c] = row[c];
            mk[r] = pm[jc];
        }
    }
    const int jl = lane * CPL;
    const bool live = jl < g.n_kv;                      // (n_kv is a multiple of 8: qmm_attn_decode_supported)
    hcol vv[8];
    {
        const char * pv = g.v + (int64_t) hk * g.v_nb2 + (int64_t) (live ? jl : 0) * 2;
#pragma unroll
        for (int r = 0; r < 8; ++r) {                   // rows wave + 16 r
            const int d = wave + 16 * r < g.Dv ? wave + 16 * r : 0;
            vv[r] = *(const hcol *) (pv + (int64_t) d * g.v_nb1);
        }
    }
    _Float16 qh[CH];
    if (!FRESH) {
        const float * pq = (const float *) (g.q + (int64_t) n * g.q_nb1 + (int64_t) h * g.q_nb2) + l8 * CH;
#pragma unroll
        for (int e = 0; e < CH; ++e) qh[e] = (_Float16) pq[e];
    }
    ATTN_STAMP(7);
    // ---- the batch's own rows: rope(q) -> LDS, rope(k) -> LDS and K cache, v -> LDS and V cache
    if (FRESH) {
        constexpr int BEHIND = NR * NV + NR + 8;        // K rows, masks and V rows requested behind the four asm loads
        static_assert(BEHIND == 20 || BEHIND == 16 || BEHIND == 14 || BEHIND == 12, "vmcnt immediates below");
        if (BEHIND == 20)      asm volatile("s_waitcnt vmcnt(20)" : "+v"(xpos), "+v"(x0), "+v"(x1), "+v"(xff) :: "memory");
        else if (BEHIND == 16) asm volatile("s_waitcnt vmcnt(16)" : "+v"(xpos), "+v"(x0), "+v"(x1), "+v"(xff) :: "memory");
        else if (BEHIND == 14) asm volatile("s_waitcnt vmcnt(14)" : "+v"(xpos), "+v"(x0), "+v"(x1), "+v"(xff) :: "memory");
        else                   asm volatile("s_waitcnt vmcnt(12)" : "+v"(xpos), "+v"(x0), "+v"(x1), "+v"(xff) :: "memory");
        if (!f.ff) xff = 1.0f;
        const bool writer = h % g.gqa == 0 && n == 0;
        auto prepare = [&](int it, float a, float b, float posf, float ffp) {
            if (it < HP + nk) {
                const int p = it < HP ? it : (it - HP) % HP;
                float c, s;
                rope_cs_ff(posf, (uint32_t) p, f.rp, ffp, c, s);
                const _Float16 h0 = (_Float16) (a * c - b * s), h1 = (_Float16) (a * s + b * c);
                if (it < HP) { qs[2 * p] = h0; qs[2 * p + 1] = h1; }        // (q is rounded to f16 by the product with K anyway)
                else {
                    const int n2 = (it - HP) / HP;
                    knew[n2 * D + 2 * p] = h0; knew[n2 * D + 2 * p + 1] = h1;
                    if (writer) {
                        _Float16 * kd = reinterpret_cast<_Float16 *>(f.kd + (int64_t) hk * f.kd_nbh + (int64_t) n2 * f.kd_nbn);
                        kd[2 * p] = h0; kd[2 * p + 1] = h1;
                    }
                }
            } else {
                const int n2 = (it - HP - nk) / g.Dv, d = (it - HP - nk) % g.Dv;
                const _Float16 v = (_Float16) a;
                vnew[n2 * g.Dv + d] = v;
                if (writer) *reinterpret_cast<_Float16 *>(f.vd + (int64_t) n2 * 2 + (int64_t) (hk * g.Dv + d) * f.vd_nbc) = v;
            }
        };
        if (wave * 64 < items) {                        // wave-uniform: the other waves go straight to the barrier
            if (tid < items) prepare(tid, x0, x1, (float) xpos, xff);
            for (int it = tid + 1024; it < items; it += 1024) {     // batches of several tokens
                const float * p0, * p1; int tok;
                item(it, p0, p1, tok);
                prepare(it, *p0, *p1, (float) f.pos[tok], f.ff && it < HP + nk ? f.ff[it < HP ? it : (it - HP) % HP] : 1.0f);
            }
        }
        ATTN_STAMP(1);
        __syncthreads();
        ATTN_STAMP(2);
#pragma unroll
        for (int c = 0; c < NV; ++c) {
            const h16x8 t = *(const h16x8 *) &qs[l8 * CH + c * 8];
#pragma unroll
            for (int e = 0; e < 8; ++e) qh[c * 8 + e] = t[e];
        }
    }
    // ---- scores
#pragma unroll
    for (int trip = 0; trip < TRIPS; ++trip) {
        if (trip > 0) {                                 // the second half's rows into the registers the first half has released
            const float * pm = (const float *) (g.mask + (int64_t) n * g.m_nb1);
            const char *  pk = g.k + (int64_t) hk * g.k_nb2 + (int64_t) l8 * CH * 2;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int j = 512 * trip + grp + 128 * r, jc = j < g.n_kv ? j : g.n_kv - 1;
                const h16x8 * row = (const h16x8 *) (pk + (int64_t) jc * g.k_nb1);
#pragma unroll
                for (int c = 0; c < NV; ++c) kv[r][c] = row[c];
                mk[r] = pm[jc];
            }
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int j = 512 * trip + grp + 128 * r;
            if (FRESH && (unsigned) (j - f.j0) < (unsigned) f.N) {
#pragma unroll
                for (int c = 0; c < NV; ++c) kv[r][c] = *(const h16x8 *) &knew[(j - f.j0) * D + l8 * CH + c * 8];
            }
            float s = 0.0f;
#pragma unroll
            for (int c = 0; c < NV; ++c)
#pragma unroll
                for (int e = 0; e < 8; ++e) s += (float) kv[r][c][e] * (float) qh[c * 8 + e];
            s = sum8(s);
            s = s * g.scale + mk[r];
            if (j < g.n_kv && l8 == 0) sc[j] = s;
        }
    }
    // the second half's V rows are requested here (the K rows' registers are free), in front of the softmax arithmetic
    hcol vv2[8];
    const int jl2 = 512 + jl;
    const bool live2 = TRIPS > 1 && jl2 < g.n_kv;
    if (TRIPS > 1) {
        const char * pv = g.v + (int64_t) hk * g.v_nb2 + (int64_t) (live2 ? jl2 : 0) * 2;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int d = wave + 16 * r < g.Dv ? wave + 16 * r : 0;
            vv2[r] = *(const hcol *) (pv + (int64_t) d * g.v_nb1);
        }
    }
    ATTN_STAMP(3);
    __syncthreads();
    ATTN_STAMP(4);
    // ---- softmax: every wave takes the maximum for itself, thread j the exponential of column j
    float m = -INFINITY;
#pragma unroll
    for (int trip = 0; trip < TRIPS; ++trip) {
        const int jt = 512 * trip + jl;
#pragma unroll
        for (int e = 0; e < CPL; e += 4) {
            const float4 a = jt < g.n_kv ? *(const float4 *) &sc[jt + e] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
            m = fmaxf(m, fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)));
        }
    }
    m = wave_max(m);
    if (tid < g.n_kv) ex[tid] = expf(sc[tid] - m);
    __syncthreads();
    float e8[TRIPS][CPL];
    float sum = 0.0f;
#pragma unroll
    for (int trip = 0; trip < TRIPS; ++trip) {
        const int jt = 512 * trip + jl;
#pragma unroll
        for (int e = 0; e < CPL; e += 4) {
            const float4 a = jt < g.n_kv ? *(const float4 *) &ex[jt + e] : make_float4(0.f, 0.f, 0.f, 0.f);
            e8[trip][e] = a.x; e8[trip][e + 1] = a.y; e8[trip][e + 2] = a.z; e8[trip][e + 3] = a.w;
        }
#pragma unroll
        for (int e = 0; e < CPL; ++e) sum += e8[trip][e];
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    _Float16 pr[TRIPS][CPL];
#pragma unroll
    for (int trip = 0; trip < TRIPS; ++trip)
#pragma unroll
        for (int e = 0; e < CPL; ++e) {                 // the batch's own positions: column weight 0, their rows come from LDS below
            const bool fresh = FRESH && (unsigned) (512 * trip + jl + e - f.j0) < (unsigned) f.N;
            pr[trip][e] = fresh ? (_Float16) 0.0f : (_Float16) (e8[trip][e] * inv);
        }
    ATTN_STAMP(5);
    // ---- V product: eight rows per wave, their 64 partial sums each folded pairwise into one register
    float acc[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        acc[r] = 0.0f;
#pragma unroll
        for (int e = 0; e < CPL; ++e) acc[r] += (float) vv[r][e] * (float) pr[0][e];
        if (TRIPS > 1) {
#pragma unroll
            for (int e = 0; e < CPL; ++e) acc[r] += (float) vv2[r][e] * (float) pr[TRIPS - 1][e];
        }
    }
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    float t4[4], t2[2];
#pragma unroll
    for (int r = 0; r < 4; ++r) {                       // lanes 0-31: row r, lanes 32-63: row r + 4
        const u32x2 w = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[r]), __float_as_uint(acc[r + 4]), false, false);
        t4[r] = __uint_as_float(w.x) + __uint_as_float(w.y);
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {                       // 16-lane rows: r, r + 2, r + 4, r + 6
        const u32x2 w = __builtin_amdgcn_permlane16_swap(__float_as_uint(t4[r]), __float_as_uint(t4[r + 2]), false, false);
        t2[r] = __uint_as_float(w.x) + __uint_as_float(w.y);
    }
    t2[0] += dpp_mov<DPP_ROW_MIRROR>(t2[0]);
    t2[1] += dpp_mov<DPP_ROW_MIRROR>(t2[1]);
    float t = sum8(lane & 8 ? t2[1] : t2[0]);           // lanes 16 R + 8 b ...: row 2 R + b
    const int d = wave + 16 * (2 * (lane >> 4) + ((lane >> 3) & 1));
    if (l8 == 0 && d < g.Dv) {
        if (FRESH)
            for (int i = 0; i < f.N; ++i) t += (float) (_Float16) (ex[f.j0 + i] * inv) * (float) vnew[i * g.Dv + d];
        ((float *) (g.dst + (int64_t) n * g.d_nb1) + (int64_t) h * g.Dv)[d] = t;
    }
    ATTN_STAMP(6);
}

// Long caches at batch <= 8: one workgroup per (head, token) walks the whole cache (45 us per layer at n_kv = 4160).  From
// n_kv = 1024 the kv range is cut into S pieces, one workgroup each (grid z), which leave (max, sum, unnormalised output row) in
// the workspace; attn_combine_kernel merges them with the usual rescaling.  p stays f32 here (the single-workgroup kernel and
// the CPU round the normalised p to f16 before the product with V): results agree to ~2^-11 relative.
template <int D>
__global__ void __launch_bounds__(1024)
attn_decode_split_kernel(const AttnArgs g, float * __restrict__ part, const int S, const int chunk) {
    extern __shared__ float sc[];                       // this piece's scores, then exp(score - max)
    __shared__ float red[16];
    const int h = blockIdx.x, n = blockIdx.y, sp = blockIdx.z, hk = h / g.gqa;
    const int j_lo = sp * chunk, j_hi = min(g.n_kv, j_lo + chunk), len = max(j_hi - j_lo, 0);
    const int tid = threadIdx.x, l8 = tid & 7, grp = tid >> 3;
    constexpr int CH = D / 8, NV = CH / 8;
    float qf[CH];
    {
        const float * pq = (const float *) (g.q + (int64_t) n * g.q_nb1 + (int64_t) h * g.q_nb2) + l8 * CH;
#pragma unroll
        for (int e = 0; e < CH; ++e) qf[e] = (float) (_Float16) pq[e];
    }
    const float * pm = (const float *) (g.mask + (int64_t) n * g.m_nb1);
    const char *  pk = g.k + (int64_t) hk * g.k_nb2 + (int64_t) l8 * CH * 2;
    float mx = -INFINITY;
    for (int jt = j_lo + grp; jt < j_hi; jt += 256) {
        h16x8 kv[2][NV];
        float mk[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int j = jt + r * 128, jc = j < j_hi ? j : jt;
            const h16x8 * row = (const h16x8 *) (pk + (int64_t) jc * g.k_nb1);
#pragma unroll
            for (int c = 0; c < NV; ++c) kv[r][c] = row[c];
            mk[r] = pm[jc];
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int j = jt + r * 128;
            float s = 0.0f;
#pragma unroll
            for (int c = 0; c < NV; ++c)
#pragma unroll
                for (int e = 0; e < 8; ++e) s += (float) kv[r][c][e] * qf[c * 8 + e];
            s = sum8(s);
            s = s * g.scale + mk[r];
            if (j < j_hi) {
                if (l8 == 0) sc[j - j_lo] = s;
                mx = fmaxf(mx, s);
            }
        }
    }
    mx = block_reduce<true>(mx, red);
    float sum = 0.0f;
    for (int j = tid; j < len; j += 1024) {
        const float e = mx == -INFINITY ? 0.0f : expf(sc[j] - mx);
        sc[j] = e;
        sum += e;
    }
    sum = block_reduce<false>(sum, red);                // its barriers publish sc[]
    const int lane = tid & 63, wave = tid >> 6;
    float * po = part + ((int64_t) (h * gridDim.y + n) * S + sp) * (g.Dv + 2);
    if (tid == 0) { po[0] = mx; po[1] = sum; }
    const char * pv = g.v + (int64_t) hk * g.v_nb2;
    for (int d0 = wave; d0 < g.Dv; d0 += 64) {
        float acc[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
        for (int j = lane * 8; j < len; j += 512) {
            h16x8 vv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int d = d0 + 16 * r < g.Dv ? d0 + 16 * r : d0;
                vv[r] = *(const h16x8 *) (pv + (int64_t) d * g.v_nb1 + (int64_t) (j_lo + j) * 2);
            }
            float p[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) p[e] = j + e < len ? sc[j + e] : 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[r] += (float) vv[r][e] * p[e];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float t = wave_sum(acc[r]);
            if (lane == 0 && d0 + 16 * r < g.Dv) po[2 + d0 + 16 * r] = t;
        }
    }
}
__global__ void __launch_bounds__(256)
attn_combine_kernel(const float * __restrict__ part, char * __restrict__ dst, const int64_t d_nb1, const int Dv, const int S) {
    const int h = blockIdx.x, n = blockIdx.y;
    const float * pp = part + (int64_t) (h * gridDim.y + n) * S * (Dv + 2);
    float M = -INFINITY;
    for (int s = 0; s < S; ++s) M = fmaxf(M, pp[(int64_t) s * (Dv + 2)]);
    float L = 0.0f;
    for (int s = 0; s < S; ++s) L += pp[(int64_t) s * (Dv + 2) + 1] * expf(pp[(int64_t) s * (Dv + 2)] - M);
    float * out = (float *) (dst + (int64_t) n * d_nb1) + (int64_t) h * Dv;
    for (int d = threadIdx.x; d < Dv; d += 256) {
        float o = 0.0f;
        for (int s = 0; s < S; ++s) o += pp[(int64_t) s * (Dv + 2) + 2 + d] * expf(pp[(int64_t) s * (Dv + 2)] - M);
        out[d] = o / L;
    }
}

// ------------------------------------------------------------------------------------------------ attention, prompt batches
// The same chain for a prompt batch whose scores fit LDS (n_kv <= 512: llama-bench's pp512), one workgroup of 4 waves per
// (64 tokens, head).  The three-launch form is bound by the f32 score tensor (33 MB per layer at 512 x 512 x 32: written by KQ,
// read and written by soft_max, read by KQV); here the 64 x n_kv scores live in LDS from the first MFMA to the last:
//   1. S = scale * Q K^T + mask      K tiles of 64 rows through LDS, Q fragments in registers, v_mfma_f32_32x32x16_f16
//   2. soft_max per row in LDS       a wave per row, the row in registers; p is stored back as f16 in the row's own bytes
//   3. O = P V                       V^T tiles of 32 columns through LDS, P fragments read from the score rows
// q and p are rounded to f16 as on the CPU path (F16 vec_dot); the result goes out in the merged-heads layout.
constexpr int AP_TN = 64, AP_KT = 64, AP_CH = 512;                        // tokens per workgroup, K rows / V columns per tile, kv columns per chunk
// Caches longer than one chunk are walked chunk by chunk with the usual running (max, sum) per token row: the output accumulators
// are rescaled by exp(max_old - max_new) before a chunk's P V is added and divided by the sum at the end.  With several chunks
// p is stored unnormalised (exp(s - max), f16): against the CPU, which rounds the normalised p, results agree to f16 rounding
// (~1e-3 of an output at worst); a cache of one chunk keeps the CPU's order of operations.
template <int D>
__global__ void __launch_bounds__(256)
attn_prefill_kernel(const AttnArgs g, const int N) {
    extern __shared__ __attribute__((aligned(16))) uint8_t ap_smem[];
    __shared__ int tile_dead[AP_CH / AP_KT];                               // kv tile fully masked for all 64 tokens (the causal upper triangle)
    __shared__ float row_max[AP_TN], row_sum[AP_TN], row_alpha[AP_TN];
    const int CW = g.n_kv < AP_CH ? g.n_kv : AP_CH;                        // chunk width
    const bool single = g.n_kv <= AP_CH;
    const int SP = CW + 4;                                                 // score row pitch in floats: 16-byte reads of 32 rows hit distinct banks
    float *    S  = reinterpret_cast<float *>(ap_smem);
    _Float16 * tl = reinterpret_cast<_Float16 *>(ap_smem + (size_t) AP_TN * SP * 4);      // K tile [64][D + 8], later V tile [D][64 + 8]
    constexpr int KP = D + 8, VP = AP_KT + 8;
    constexpr int NCH = AP_KT * D / 8 / 256;                               // 16-byte chunks of a tile per thread (both tiles hold 64 * D halves)
    const int h = blockIdx.y, hk = h / g.gqa, n0 = blockIdx.x * AP_TN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, hh = lane >> 5;
    const int tw = wave & 1, kw = wave >> 1;

    // Q fragments of this wave's 32 tokens (MFMA rows), f32 -> f16
    h16x8 qf[D / 16];
    {
        const int n = n0 + 32 * tw + l32;
        const float * pq = (const float *) (g.q + (int64_t) (n < N ? n : N - 1) * g.q_nb1 + (int64_t) h * g.q_nb2);
#pragma unroll
        for (int kk = 0; kk < D / 16; ++kk) {
            const float4 x = *(const float4 *) (pq + 16 * kk + 8 * hh), y = *(const float4 *) (pq + 16 * kk + 8 * hh + 4);
            qf[kk] = h16x8{ (_Float16) x.x, (_Float16) x.y, (_Float16) x.z, (_Float16) x.w, (_Float16) y.x, (_Float16) y.y, (_Float16) y.z, (_Float16) y.w };
        }
    }
    if (tid < AP_TN) { row_max[tid] = -INFINITY; row_sum[tid] = 0.0f; }
    constexpr int DT = D / 64;                                             // 32-wide d tiles per wave
    constexpr int CPR = D / 8;                                             // 16-byte chunks per K row
    constexpr int VPR = AP_KT / 8;                                         // 16-byte chunks per V^T row of the tile
    f32x16v oacc[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) oacc[t] = f32x16v{0};
    const char * pk = g.k + (int64_t) hk * g.k_nb2;
    const char * pv = g.v + (int64_t) hk * g.v_nb2;
    h16x8 pre[NCH];

    for (int c0 = 0; c0 < g.n_kv; c0 += AP_CH) {
        const int c1 = c0 + AP_CH < g.n_kv ? c0 + AP_CH : g.n_kv;          // this chunk: kv columns c0 .. c1
        // 1. scores.  A tile's K rows are requested one tile ahead, so the loads fly under the MFMAs of the current one.
        auto fetch_k = [&](int j0) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int c = tid + 256 * i, row = c / CPR, col = c % CPR;
                pre[i] = h16x8{0};
                if (j0 + row < c1) pre[i] = *(const h16x8 *) (pk + (int64_t) (j0 + row) * g.k_nb1 + col * 16);
            }
        };
        fetch_k(c0);
        int live_tiles = 0;
        for (int j0 = c0, jt = 0; j0 < c1; j0 += AP_KT, ++jt) {
            const int j = j0 + 32 * kw + l32;
            float m[16];
            int dead = 1;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + 32 * tw + 8 * (r >> 2) + 4 * hh + (r & 3);
                m[r] = j < c1 ? *(const float *) (g.mask + (int64_t) (n < N ? n : N - 1) * g.m_nb1 + (int64_t) j * 4) : -INFINITY;
                dead &= m[r] == -INFINITY;
            }
            dead = __syncthreads_and(dead);                                // also: the previous tile's fragment reads are done
            if (tid == 0) tile_dead[jt] = dead;
            live_tiles += !dead;
            if (!dead) {
#pragma unroll
                for (int i = 0; i < NCH; ++i) {
                    const int c = tid + 256 * i;
                    *(h16x8 *) &tl[(c / CPR) * KP + (c % CPR) * 8] = pre[i];
                }
            }
            __syncthreads();
            if (j0 + AP_KT < c1) fetch_k(j0 + AP_KT);
            f32x16v acc = {0};
            if (!dead) {
#pragma unroll
                for (int kk = 0; kk < D / 16; ++kk) {
                    const h16x8 kf = *(const h16x8 *) &tl[(32 * kw + l32) * KP + 16 * kk + 8 * hh];
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(qf[kk], kf, acc, 0, 0, 0);
                }
            }
            if (j < c1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) S[(32 * tw + 8 * (r >> 2) + 4 * hh + (r & 3)) * SP + j - c0] = dead ? -INFINITY : acc[r] * g.scale + m[r];
            }
        }
        __syncthreads();
        if (live_tiles == 0) continue;                                     // block-uniform: the whole chunk is masked for these tokens
        // 2. soft_max step: 16 rows per wave, <= 512 values = two float4 per lane; running max / sum per row
        const int cw = c1 - c0;
        typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
        for (int n4 = wave * 16; n4 < wave * 16 + 16; n4 += 4) {            // four rows at a time: their reads, reductions and exps interleave
            float4 v[4][2];
            float mx[4], m_old[4], m_new[4], sum[4], alpha[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float * row = S + (n4 + q) * SP;
                mx[q] = -INFINITY;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int i = 4 * (lane + 64 * t);
                    v[q][t] = i < cw ? *(const float4 *) (row + i) : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
                    mx[q] = fmaxf(fmaxf(mx[q], fmaxf(v[q][t].x, v[q][t].y)), fmaxf(v[q][t].z, v[q][t].w));
                }
                m_old[q] = row_max[n4 + q];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) m_new[q] = fmaxf(m_old[q], wave_max(mx[q]));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                sum[q] = 0.0f;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    float4 & x = v[q][t];
                    if (m_new[q] == -INFINITY) x = make_float4(0.f, 0.f, 0.f, 0.f);
                    else { x.x = __expf(x.x - m_new[q]); x.y = __expf(x.y - m_new[q]); x.z = __expf(x.z - m_new[q]); x.w = __expf(x.w - m_new[q]); }
                    sum[q] += x.x + x.y + x.z + x.w;
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                sum[q] = wave_sum(sum[q]);
                alpha[q] = m_old[q] == -INFINITY ? 0.0f : __expf(m_old[q] - m_new[q]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (single) {                                              // one chunk: normalise before the f16 rounding, exactly as the CPU path does
                    const float inv = 1.0f / sum[q];
#pragma unroll
                    for (int t = 0; t < 2; ++t) { v[q][t].x *= inv; v[q][t].y *= inv; v[q][t].z *= inv; v[q][t].w *= inv; }
                    sum[q] = 1.0f;
                }
                const int nl = n4 + q;
                if (lane == 0) { row_max[nl] = m_new[q]; row_sum[nl] = row_sum[nl] * alpha[q] + sum[q]; row_alpha[nl] = alpha[q]; }
                _Float16 * prow = reinterpret_cast<_Float16 *>(S + nl * SP);    // p as f16 over the first half of the row's own bytes
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int i = 4 * (lane + 64 * t);
                    if (i < cw) *(h16x4 *) (prow + i) = h16x4{ (_Float16) v[q][t].x, (_Float16) v[q][t].y, (_Float16) v[q][t].z, (_Float16) v[q][t].w };
                }
            }
        }
        __syncthreads();                                                   // tile_dead, row_alpha and the p rows are visible to everyone
        // 3. O = alpha * O + P V: wave (tw, kw) owns tokens 32 tw .. +32 and the d tiles kw * DT .. + DT; dead kv tiles are not loaded
#pragma unroll
        for (int t = 0; t < DT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[t][r] *= row_alpha[32 * tw + 8 * (r >> 2) + 4 * hh + (r & 3)];
        auto fetch_v = [&](int j0) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int c = tid + 256 * i, row = c / VPR, col = c % VPR;
                pre[i] = h16x8{0};
                if (j0 + col * 8 < c1) pre[i] = *(const h16x8 *) (pv + (int64_t) row * g.v_nb1 + (int64_t) (j0 + col * 8) * 2);
            }
        };
        int jn = c0, jtn = 0;                                              // next live tile
        while (jn < c1 && tile_dead[jtn]) { jn += AP_KT; ++jtn; }
        if (jn < c1) fetch_v(jn);
        while (jn < c1) {
            const int j0 = jn;
            jn += AP_KT; ++jtn;
            while (jn < c1 && tile_dead[jtn]) { jn += AP_KT; ++jtn; }
            __syncthreads();                                               // the previous tile's fragment reads are done
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int c = tid + 256 * i;
                *(h16x8 *) &tl[(c / VPR) * VP + (c % VPR) * 8] = pre[i];
            }
            __syncthreads();
            if (jn < c1) fetch_v(jn);
#pragma unroll
            for (int ks = 0; ks < AP_KT / 16; ++ks) {
                if (j0 + 16 * ks >= c1) break;
                const h16x8 pf = *(const h16x8 *) (reinterpret_cast<const _Float16 *>(S + (32 * tw + l32) * SP) + (j0 - c0) + 16 * ks + 8 * hh);
#pragma unroll
                for (int t = 0; t < DT; ++t) {
                    const h16x8 vf = *(const h16x8 *) &tl[(32 * (kw * DT + t) + l32) * VP + 16 * ks + 8 * hh];
                    oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pf, vf, oacc[t], 0, 0, 0);
                }
            }
        }
        __syncthreads();                                                   // the score tile and tile_dead are free for the next chunk
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < DT; ++t) {
        const int d = 32 * (kw * DT + t) + l32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int nl = 32 * tw + 8 * (r >> 2) + 4 * hh + (r & 3), n = n0 + nl;
            if (n < N) *(float *) (g.dst + (int64_t) n * g.d_nb1 + ((int64_t) h * g.Dv + d) * 4) = oacc[t][r] / row_sum[nl];
        }
    }
}

// ------------------------------------------------------------------------------------------------ support predicates

bool is_binary(int op) { return op >= QMM_OP_ADD && op <= QMM_OP_DIV; }
bool is_unary(int op) { return op >= QMM_OP_SCALE && op <= QMM_OP_EXP; }
bool quant_type(int t) { return type_known(t); }
int  quant_blck(int t) { return type_blck(t); }

bool sup_binary(const qmm_tensor * a, const qmm_tensor * b, const qmm_tensor * d) {
    if (!a || !b || !d || a->type != G_F32 || b->type != G_F32 || d->type != G_F32) return false;
    if (!same_shape(a, d) || !dense_rows(a) || !dense_rows(d) || !fits_u32(d) || nelements(d) == 0) return false;
    for (int i = 0; i < 4; ++i) if (b->ne[i] <= 0 || d->ne[i] % b->ne[i]) return false;       // ggml_can_repeat(b, a)
    return a->nb[1] % 4 == 0 && a->nb[2] % 4 == 0 && a->nb[3] % 4 == 0 && b->nb[0] % 4 == 0 && b->nb[1] % 4 == 0 && b->nb[2] % 4 == 0 &&
           b->nb[3] % 4 == 0;
}
bool sup_unary(const qmm_tensor * a, const qmm_tensor * d) {
    return a && d && a->type == G_F32 && d->type == G_F32 && same_shape(a, d) && contiguous(a) && contiguous(d) && fits_u32(d);
}
bool sup_rms_norm(const qmm_tensor * a, const qmm_tensor * d) {
    return a && d && a->type == G_F32 && d->type == G_F32 && same_shape(a, d) && dense_rows(a) && dense_rows(d) && fits_u32(d) &&
           nelements(d) > 0;
}
bool sup_rope(const qmm_tensor * a, const qmm_tensor * pos, const qmm_tensor * ff, const qmm_tensor * d) {
    if (!a || !pos || !d || a->type != G_F32 || d->type != G_F32 || pos->type != G_I32 || !same_shape(a, d)) return false;
    if (!dense_rows(a) || !dense_rows(d) || !contiguous(pos) || !fits_u32(d) || nelements(d) == 0 || a->ne[0] % 2) return false;
    const int n_dims = d->op_params[1], mode = d->op_params[2];
    if (mode != 0 && mode != 2) return false;                                     // normal or NEOX; no M-RoPE / vision
    if (n_dims <= 0 || n_dims % 2 || n_dims > a->ne[0] || pos->ne[0] != a->ne[2]) return false;
    if (ff && (ff->type != G_F32 || ff->ne[0] < n_dims / 2 || !contiguous(ff))) return false;
    return true;
}
bool sup_soft_max(const qmm_tensor * a, const qmm_tensor * mask, const qmm_tensor * d) {
    if (!a || !d || a->type != G_F32 || d->type != G_F32 || !same_shape(a, d) || !contiguous(a) || !contiguous(d)) return false;
    if (nelements(d) == 0 || nrows(d) >= ((int64_t) 1 << 31)) return false;
    if (mask) {
        if (mask->type != G_F32 && mask->type != G_F16) return false;
        if (!contiguous(mask) || mask->ne[0] != a->ne[0] || mask->ne[1] < a->ne[1] || mask->ne[2] != 1 || mask->ne[3] != 1) return false;
    }
    return true;
}
bool sup_cpy(const qmm_tensor * a, const qmm_tensor * d) {
    if (!a || !d || !esize(a->type) || !esize(d->type) || a->type == G_I32 || d->type == G_I32) return false;
    return nelements(a) == nelements(d) && fits_u32(a) && a->nb[0] % esize(a->type) == 0 && d->nb[0] % esize(d->type) == 0;
}
bool sup_get_rows(const qmm_tensor * a, const qmm_tensor * ids, const qmm_tensor * d) {
    if (!a || !ids || !d || ids->type != G_I32 || d->type != G_F32 || !dense_rows(d) || nelements(d) == 0) return false;
    if (d->ne[0] != a->ne[0] || d->ne[1] != ids->ne[0] || d->ne[2] != ids->ne[1] || d->ne[3] != ids->ne[2] || ids->ne[3] != 1) return false;
    if (a->ne[2] != ids->ne[1] || a->ne[3] != ids->ne[2] || nrows(d) >= ((int64_t) 1 << 31)) return false;
    if (quant_type(a->type)) return a->ne[0] % quant_blck(a->type) == 0 && a->nb[1] % 2 == 0;
    return (a->type == G_F32 || a->type == G_F16) && dense_rows(a);
}
bool sup_mul_mat_f(const qmm_tensor * a, const qmm_tensor * b, const qmm_tensor * d) {
    if (!a || !b || !d || (a->type != G_F16 && a->type != G_F32) || b->type != G_F32 || d->type != G_F32) return false;
    if (a->ne[0] != b->ne[0] || d->ne[0] != a->ne[1] || d->ne[1] != b->ne[1] || d->ne[2] != b->ne[2] || d->ne[3] != b->ne[3]) return false;
    if (a->ne[2] <= 0 || a->ne[3] <= 0 || b->ne[2] % a->ne[2] || b->ne[3] % a->ne[3] || nelements(d) == 0 || a->ne[0] == 0) return false;
    if (a->nb[0] != esize(a->type) || b->nb[0] != 4 || d->nb[0] != 4) return false;         // K dense in both operands
    if (b->ne[2] * b->ne[3] > 65535 || a->ne[1] >= ((int64_t) 1 << 30) || b->ne[1] >= ((int64_t) 1 << 22)) return false;
    return a->nb[1] % esize(a->type) == 0 && a->nb[2] % esize(a->type) == 0 && a->nb[3] % esize(a->type) == 0 && b->nb[1] % 4 == 0 &&
           b->nb[2] % 4 == 0 && b->nb[3] % 4 == 0 && d->nb[1] % 4 == 0 && d->nb[2] % 4 == 0 && d->nb[3] % 4 == 0;
}

// ------------------------------------------------------------------------------------------------ launchers

template <int OP>
int launch_binary(hipStream_t st, const qmm_tensor * a, const qmm_tensor * b, const qmm_tensor * d) {
    const uint32_t rows = (uint32_t) nrows(d);
    const bool vec = d->ne[0] % 4 == 0 && b->ne[0] == d->ne[0] && b->nb[0] == 4 && aligned_to(a, 16) && aligned_to(b, 16) && aligned_to(d, 16);
    const uint32_t per_row = (uint32_t) (vec ? d->ne[0] / 4 : d->ne[0]);
    const uint32_t rpb = per_row >= 256 ? 1 : 256 / per_row;
    const dim3 grid((rows + rpb - 1) / rpb);
    if (vec) hipLaunchKernelGGL((binary_kernel<OP, true>), grid, dim3(256), 0, st, (const char *) a->data, (const char *) b->data, (char *) d->data,
                                shape_of(a), shape_of(b), shape_of(d), rows);
    else     hipLaunchKernelGGL((binary_kernel<OP, false>), grid, dim3(256), 0, st, (const char *) a->data, (const char *) b->data, (char *) d->data,
                                shape_of(a), shape_of(b), shape_of(d), rows);
    HIP_TRY(hipGetLastError());
    return QMM_OK;
}

template <int OP, bool MUL2>
int launch_unary(hipStream_t st, const qmm_tensor * a, const qmm_tensor * b, const qmm_tensor * d, float p) {
    const uint32_t n = (uint32_t) nelements(d);
    if (n == 0) return QMM_OK;
    hipLaunchKernelGGL((unary_kernel<OP, MUL2>), dim3((n + 1023) / 1024), dim3(256), 0, st, (const float *) a->data,
                       b ? (const float *) b->data : nullptr, (float *) d->data, n, p);
    HIP_TRY(hipGetLastError());
    return QMM_OK;
}

template <typename TS, typename TD>
int launch_cpy_t(hipStream_t st, const qmm_tensor * a, const qmm_tensor * d) {
    const uint32_t n = (uint32_t) nelements(a);
    // transposed source into dense rows (the V-cache store): 2-D, same extents, src dense along dim 1, dst dense along dim 0
    if (a->ne[2] == 1 && a->ne[3] == 1 && d->ne[2] == 1 && d->ne[3] == 1 && a->ne[0] == d->ne[0] && a->ne[1] == d->ne[1] &&
        a->nb[1] == (int64_t) sizeof(TS) && d->nb[0] == (int64_t) sizeof(TD) && a->ne[0] >= 32 && a->ne[1] >= 8) {
        hipLaunchKernelGGL((cpy_transpose_kernel<TS, TD>), dim3((unsigned) ((a->ne[0] + 31) / 32), (unsigned) ((a->ne[1] + 31) / 32)), dim3(256), 0, st,
                           (const char *) a->data, (char *) d->data, (uint32_t) a->ne[0], (uint32_t) a->ne[1], a->nb[0], d->nb[1]);
    } else {
        hipLaunchKernelGGL((cpy_kernel<TS, TD>), dim3((n + 255) / 256), dim3(256), 0, st, (const char *) a->data, (char *) d->data, shape_of(a),
                           shape_of(d), n);
    }
    HIP_TRY(hipGetLastError());
    return QMM_OK;
}
int launch_cpy(hipStream_t st, const qmm_tensor * a, const qmm_tensor * d) {
    if (nelements(a) == 0) return QMM_OK;
    if (a->type == d->type && contiguous(a) && contiguous(d)) {
        if (a->data != d->data) HIP_TRY(hipMemcpyAsync(d->data, a->data, (size_t) nelements(a) * esize(a->type), hipMemcpyDeviceToDevice, st));
        return QMM_OK;
    }
    if (a->type == G_F32 && d->type == G_F32) return launch_cpy_t<float, float>(st, a, d);
    if (a->type == G_F32 && d->type == G_F16) return launch_cpy_t<float, __half>(st, a, d);
    if (a->type == G_F16 && d->type == G_F16) return launch_cpy_t<__half, __half>(st, a, d);
    return launch_cpy_t<__half, float>(st, a, d);
}

template <int T>
int launch_get_rows_q(hipStream_t st, const qmm_tensor * a, const qmm_tensor * ids, const qmm_tensor * d) {
    hipLaunchKernelGGL((get_rows_q_kernel<T>), dim3((unsigned) nrows(d)), dim3(256), 0, st, (const uint8_t *) a->data, (const char *) ids->data,
                       (char *) d->data, shape_of(a), shape_of(ids), shape_of(d));
    HIP_TRY(hipGetLastError());
    return QMM_OK;
}

int launch_mul_mat_f(hipStream_t st, const qmm_tensor * a, const qmm_tensor * b, const qmm_tensor * d) {
    MmArgs g;
    g.a = (const char *) a->data;  g.b = (const char *) b->data;  g.d = (char *) d->data;
    g.a_nb1 = a->nb[1]; g.a_nb2 = a->nb[2]; g.a_nb3 = a->nb[3];
    g.b_nb1 = b->nb[1]; g.b_nb2 = b->nb[2]; g.b_nb3 = b->nb[3];
    g.d_nb1 = d->nb[1]; g.d_nb2 = d->nb[2]; g.d_nb3 = d->nb[3];
    g.M = (int32_t) a->ne[1]; g.N = (int32_t) b->ne[1]; g.K = (int32_t) a->ne[0];
    g.ne12 = (int32_t) b->ne[2]; g.r2 = (int32_t) (b->ne[2] / a->ne[2]); g.r3 = (int32_t) (b->ne[3] / a->ne[3]);
    const unsigned batch = (unsigned) (b->ne[2] * b->ne[3]);
    if (a->type == G_F16) {
        const dim3 grid((g.M + MM_T - 1) / MM_T, (g.N + MM_T - 1) / MM_T, batch);
        const bool vec = aligned_to(a, 16) && aligned_to(b, 16);
        if (vec) hipLaunchKernelGGL((mul_mat_f16_kernel<true>), grid, dim3(256), 0, st, g);
        else     hipLaunchKernelGGL((mul_mat_f16_kernel<false>), grid, dim3(256), 0, st, g);
    } else {
        const int64_t e = (int64_t) g.M * g.N;
        if (e <= 2048 && g.K >= 1024) hipLaunchKernelGGL(mul_mat_dot_block_kernel, dim3((unsigned) e, 1, batch), dim3(DOT_T), 0, st, g);
        else                          hipLaunchKernelGGL((mul_mat_dot_kernel<float>), dim3((unsigned) ((e + 3) / 4), 1, batch), dim3(256), 0, st, g);
    }
    HIP_TRY(hipGetLastError());
    return QMM_OK;
}

template <bool MUL, bool ADD>
int launch_rms_norm_vec(hipStream_t st, const qmm_tensor * a, const qmm_tensor * b, const qmm_tensor * w, const qmm_tensor * d, const qmm_tensor * sum,
                        float eps) {
    const unsigned rows = (unsigned) nrows(d);
    const bool wide = rows < 256 || a->ne[0] > 256 * 16;
    const Shape sb = b ? shape_of(b) : shape_of(a), ss = sum ? shape_of(sum) : shape_of(d);
    if (wide)
        hipLaunchKernelGGL((rms_norm_vec_kernel<MUL, ADD, 1024>), dim3(rows), dim3(1024), 0, st, (const char *) a->data, b ? (const char *) b->data : nullptr,
                           w ? (const float *) w->data : nullptr, (char *) d->data, sum ? (char *) sum->data : nullptr, shape_of(a), sb, shape_of(d), ss, eps);
    else
        hipLaunchKernelGGL((rms_norm_vec_kernel<MUL, ADD, 256>), dim3(rows), dim3(256), 0, st, (const char *) a->data, b ? (const char *) b->data : nullptr,
                           w ? (const float *) w->data : nullptr, (char *) d->data, sum ? (char *) sum->data : nullptr, shape_of(a), sb, shape_of(d), ss, eps);
    HIP_TRY(hipGetLastError());
    return QMM_OK;
}

// y = rms_norm(a [+ b]) [* w]; `sum` receives a + b when b is given
int launch_rms_norm(hipStream_t st, const qmm_tensor * a, const qmm_tensor * b, const qmm_tensor * w, const qmm_tensor * d, const qmm_tensor * sum, float eps) {
    if (eps < 0.0f) return fail(QMM_EINVAL, "RMS_NORM: eps < 0");
    const bool vec = a->ne[0] % 4 == 0 && a->ne[0] <= 1024 * 16 && aligned_to(a, 16) && aligned_to(d, 16) && (!w || (uintptr_t) w->data % 16 == 0) &&
                     (!b || (aligned_to(b, 16) && aligned_to(sum, 16)));
    if (vec) {
        if (b) return w ? launch_rms_norm_vec<true, true>(st, a, b, w, d, sum, eps) : launch_rms_norm_vec<false, true>(st, a, b, w, d, sum, eps);
        return w ? launch_rms_norm_vec<true, false>(st, a, b, w, d, sum, eps) : launch_rms_norm_vec<false, false>(st, a, b, w, d, sum, eps);
    }
    if (b) return fail(QMM_EUNSUPPORTED, "ADD + RMS_NORM: rows must be 16-byte aligned, ne0 %% 4 == 0 and ne0 <= 16384");
    const dim3 grid((unsigned) nrows(d));
    if (w) hipLaunchKernelGGL((rms_norm_kernel<true>), grid, dim3(256), 0, st, (const char *) a->data, (const float *) w->data, (char *) d->data,
                              shape_of(a), shape_of(d), eps);
    else   hipLaunchKernelGGL((rms_norm_kernel<false>), grid, dim3(256), 0, st, (const char *) a->data, (const float *) nullptr, (char *) d->data,
                              shape_of(a), shape_of(d), eps);
    HIP_TRY(hipGetLastError());
    return QMM_OK;
}

float f32_param(const qmm_tensor * d, int i) {
    float v;
    memcpy(&v, &d->op_params[i], sizeof(float));
    return v;
}

// ggml_rope_yarn_corr_dims (ggml.c:3738-3754)
void rope_corr_dims(int n_dims, int n_ctx_orig, float freq_base, float beta_fast, float beta_slow, float dims[2]) {
    auto corr_dim = [&](float n_rot) { return n_dims * logf(n_ctx_orig / (n_rot * 2 * (float) M_PI)) / (2 * logf(freq_base)); };
    const float start = floorf(corr_dim(beta_fast)), end = ceilf(corr_dim(beta_slow));
    dims[0] = start > 0 ? start : 0;
    dims[1] = end < n_dims - 1 ? end : n_dims - 1;
}

// op_params of ggml_rope_ext (ggml.c ggml_rope_impl) -> kernel parameters
RopeParams rope_params(const qmm_tensor * d) {
    RopeParams rp;
    rp.n_dims = d->op_params[1];
    rp.neox = (d->op_params[2] & 2) != 0;
    const int n_ctx_orig = d->op_params[4];
    const float freq_base = f32_param(d, 5), beta_fast = f32_param(d, 9), beta_slow = f32_param(d, 10);
    rp.freq_scale = f32_param(d, 6);
    rp.ext_factor = f32_param(d, 7);
    rp.attn_factor = f32_param(d, 8);
    rp.theta_scale = powf(freq_base, -2.0f / rp.n_dims);
    float cd[2];
    rope_corr_dims(rp.n_dims, n_ctx_orig, freq_base, beta_fast, beta_slow, cd);
    rp.corr0 = cd[0];
    rp.corr1 = cd[1];
    return rp;
}

} // namespace

extern "C" {

int qmm_op_supported(int op, const qmm_tensor * a, const qmm_tensor * b, const qmm_tensor * c, const qmm_tensor * d) {
    if (is_binary(op)) return sup_binary(a, b, d);
    if (is_unary(op)) return sup_unary(a, d);
    switch (op) {
        case QMM_OP_RMS_NORM:     return sup_rms_norm(a, d);
        case QMM_OP_NORM:         return sup_rms_norm(a, d);
        case QMM_OP_RMS_NORM_MUL: return sup_rms_norm(a, d) && b && b->type == G_F32 && contiguous(b) && b->ne[0] == a->ne[0] && nelements(b) == b->ne[0];
        case QMM_OP_SILU_MUL:     return sup_unary(a, d) && b && sup_unary(b, d);
        case QMM_OP_ROPE:         return sup_rope(a, b, c, d);
        case QMM_OP_SOFT_MAX:     return sup_soft_max(a, b, d);
        case QMM_OP_CPY:          return sup_cpy(a, d);
        case QMM_OP_GET_ROWS:     return sup_get_rows(a, b, d);
        case QMM_OP_MUL_MAT_F:    return sup_mul_mat_f(a, b, d);
        case QMM_OP_ARGSORT:      return a && d && a->type == G_F32 && d->type == G_I32 && same_shape(a, d) && dense_rows(a) && dense_rows(d) &&
                                         nelements(a) > 0 && a->ne[0] <= 4096 && nrows(a) < ((int64_t) 1 << 31) && (d->op_params[0] == 0 || d->op_params[0] == 1);
        case QMM_OP_SUM_ROWS:     return a && d && a->type == G_F32 && d->type == G_F32 && dense_rows(a) && d->ne[0] == 1 && d->ne[1] == a->ne[1] &&
                                         d->ne[2] == a->ne[2] && d->ne[3] == a->ne[3] && nelements(a) > 0 && nrows(a) < ((int64_t) 1 << 31);
        default:                  return 0;
    }
}

int qmm_op_add_rms_norm_supported(const qmm_tensor * a, const qmm_tensor * b, const qmm_tensor * w, const qmm_tensor * sum, const qmm_tensor * dst) {
    if (!a || !b || !sum || !dst || !sup_rms_norm(sum, dst) || !sup_binary(a, b, sum) || !same_shape(a, b) || !dense_rows(b)) return 0;
    if (a->ne[0] % 4 || a->ne[0] > 1024 * 16 || a->nb[1] % 16 || a->nb[2] % 16 || a->nb[3] % 16 || b->nb[1] % 16 || b->nb[2] % 16 || b->nb[3] % 16 ||
        sum->nb[1] % 16 || sum->nb[2] % 16 || sum->nb[3] % 16 || dst->nb[1] % 16 || dst->nb[2] % 16 || dst->nb[3] % 16) return 0;
    if (w && !(w->type == G_F32 && contiguous(w) && w->ne[0] == a->ne[0] && nelements(w) == w->ne[0])) return 0;
    return 1;
}

int qmm_op_compute(qmm_ctx * ctx, int op, const qmm_tensor * a, const qmm_tensor * b, const qmm_tensor * c, const qmm_tensor * d, void * stream) {
    if (!ctx || !d) return fail(QMM_EINVAL, "qmm_op_compute: NULL context or dst");
    if (!qmm_op_supported(op, a, b, c, d)) return fail(QMM_EUNSUPPORTED, "qmm_op_compute: op %d with these types / shapes / strides is not implemented", op);
    HIP_TRY(hipSetDevice(ctx->device));
    QMM_CHAIN_FLUSH(ctx);
    hipStream_t st = ctx->s(stream);
    switch (op) {
        case QMM_OP_ADD: return launch_binary<QMM_OP_ADD>(st, a, b, d);
        case QMM_OP_SUB: return launch_binary<QMM_OP_SUB>(st, a, b, d);
        case QMM_OP_MUL: return launch_binary<QMM_OP_MUL>(st, a, b, d);
        case QMM_OP_DIV: return launch_binary<QMM_OP_DIV>(st, a, b, d);
        case QMM_OP_SCALE:      return launch_unary<QMM_OP_SCALE, false>(st, a, nullptr, d, f32_param(d, 0));
        case QMM_OP_SILU:       return launch_unary<QMM_OP_SILU, false>(st, a, nullptr, d, 0);
        case QMM_OP_GELU:       return launch_unary<QMM_OP_GELU, false>(st, a, nullptr, d, 0);
        case QMM_OP_GELU_QUICK: return launch_unary<QMM_OP_GELU_QUICK, false>(st, a, nullptr, d, 0);
        case QMM_OP_RELU:       return launch_unary<QMM_OP_RELU, false>(st, a, nullptr, d, 0);
        case QMM_OP_TANH:       return launch_unary<QMM_OP_TANH, false>(st, a, nullptr, d, 0);
        case QMM_OP_SIGMOID:    return launch_unary<QMM_OP_SIGMOID, false>(st, a, nullptr, d, 0);
        case QMM_OP_NEG:        return launch_unary<QMM_OP_NEG, false>(st, a, nullptr, d, 0);
        case QMM_OP_EXP:        return launch_unary<QMM_OP_EXP, false>(st, a, nullptr, d, 0);
        case QMM_OP_SILU_MUL:   return launch_unary<QMM_OP_SILU, true>(st, a, b, d, 0);
        case QMM_OP_NORM: {
            const float eps = f32_param(d, 0);
            if (eps < 0.0f) return fail(QMM_EINVAL, "NORM: eps < 0");
            hipLaunchKernelGGL(norm_kernel, dim3((unsigned) nrows(d)), dim3(256), 0, st, (const char *) a->data, (char *) d->data, shape_of(a), shape_of(d), eps);
            HIP_TRY(hipGetLastError());
            return QMM_OK;
        }
        case QMM_OP_RMS_NORM:
        case QMM_OP_RMS_NORM_MUL:
            return launch_rms_norm(st, a, nullptr, op == QMM_OP_RMS_NORM_MUL ? b : nullptr, d, nullptr, f32_param(d, 0));
        case QMM_OP_ROPE: {
            const RopeParams rp = rope_params(d);
            const uint32_t pairs = (uint32_t) (nelements(d) / 2);
            hipLaunchKernelGGL(rope_kernel, dim3((pairs + 255) / 256), dim3(256), 0, st, (const char *) a->data, (const int32_t *) b->data,
                               c ? (const float *) c->data : nullptr, (char *) d->data, shape_of(a), shape_of(d), rp, pairs);
            HIP_TRY(hipGetLastError());
            return QMM_OK;
        }
        case QMM_OP_SOFT_MAX: {
            const float scale = f32_param(d, 0), max_bias = f32_param(d, 1);
            const uint32_t nc = (uint32_t) a->ne[0], ne01 = (uint32_t) a->ne[1], ne02 = (uint32_t) a->ne[2];
            const uint32_t n_head_log2 = 1u << (uint32_t) floor(log2((double) ne02));
            const float m0 = powf(2.0f, -(max_bias) / n_head_log2), m1 = powf(2.0f, -(max_bias / 2.0f) / n_head_log2);
            const size_t lds = nc <= 8192 ? (size_t) nc * 4 : 0;
            const dim3 grid((unsigned) nrows(d));
            const uint32_t rows = (uint32_t) nrows(d);
            if (max_bias == 0.0f && (!b || b->type == G_F32) && nc % 4 == 0 && nc <= 1024 && rows >= 1024 && (uintptr_t) a->data % 16 == 0 &&
                (uintptr_t) d->data % 16 == 0 && (!b || (uintptr_t) b->data % 16 == 0)) {
                const float * m = b ? (const float *) b->data : nullptr;
                if (nc <= 256)      hipLaunchKernelGGL((soft_max_wave_kernel<1>), dim3((rows + 3) / 4), dim3(256), 0, st, (const float *) a->data, m, (float *) d->data, nc, ne01, rows, scale);
                else if (nc <= 512) hipLaunchKernelGGL((soft_max_wave_kernel<2>), dim3((rows + 3) / 4), dim3(256), 0, st, (const float *) a->data, m, (float *) d->data, nc, ne01, rows, scale);
                else                hipLaunchKernelGGL((soft_max_wave_kernel<4>), dim3((rows + 3) / 4), dim3(256), 0, st, (const float *) a->data, m, (float *) d->data, nc, ne01, rows, scale);
                HIP_TRY(hipGetLastError());
                return QMM_OK;
            }
            if (b && b->type == G_F16)
                hipLaunchKernelGGL((soft_max_kernel<true>), grid, dim3(256), lds, st, (const float *) a->data, (const void *) b->data, (float *) d->data,
                                   nc, ne01, ne02, scale, max_bias, m0, m1, n_head_log2);
            else
                hipLaunchKernelGGL((soft_max_kernel<false>), grid, dim3(256), lds, st, (const float *) a->data, b ? (const void *) b->data : nullptr,
                                   (float *) d->data, nc, ne01, ne02, scale, max_bias, m0, m1, n_head_log2);
            HIP_TRY(hipGetLastError());
            return QMM_OK;
        }
        case QMM_OP_CPY: return launch_cpy(st, a, d);
        case QMM_OP_GET_ROWS: {
            const dim3 grid((unsigned) nrows(d));
            switch (a->type) {
                case G_F32: hipLaunchKernelGGL((get_rows_kernel<float>), grid, dim3(256), 0, st, (const char *) a->data, (const char *) b->data,
                                               (char *) d->data, shape_of(a), shape_of(b), shape_of(d)); break;
                case G_F16: hipLaunchKernelGGL((get_rows_kernel<__half>), grid, dim3(256), 0, st, (const char *) a->data, (const char *) b->data,
                                               (char *) d->data, shape_of(a), shape_of(b), shape_of(d)); break;
                default: {
#define QMM_X(TT) return launch_get_rows_q<TT>(st, a, b, d)
                    QMM_FOR_TYPE(a->type, QMM_X)
#undef QMM_X
                }
            }
            HIP_TRY(hipGetLastError());
            return QMM_OK;
        }
        case QMM_OP_MUL_MAT_F: return launch_mul_mat_f(st, a, b, d);
        case QMM_OP_ARGSORT:
            hipLaunchKernelGGL(argsort_kernel, dim3((unsigned) nrows(a)), dim3(256), 0, st, (const char *) a->data, (char *) d->data, shape_of(a), shape_of(d),
                               d->op_params[0]);
            HIP_TRY(hipGetLastError());
            return QMM_OK;
        case QMM_OP_SUM_ROWS: {
            const uint32_t rows = (uint32_t) nrows(a);
            hipLaunchKernelGGL(sum_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, (const char *) a->data, (char *) d->data, shape_of(a), shape_of(d), rows);
            HIP_TRY(hipGetLastError());
            return QMM_OK;
        }
        default: return fail(QMM_EUNSUPPORTED, "qmm_op_compute: unknown op %d", op);
    }
}

int qmm_op_add_rms_norm(qmm_ctx * ctx, const qmm_tensor * a, const qmm_tensor * b, const qmm_tensor * w, const qmm_tensor * sum,
                        const qmm_tensor * dst, float eps, void * stream) {
    if (!ctx || !qmm_op_add_rms_norm_supported(a, b, w, sum, dst)) return fail(QMM_EUNSUPPORTED, "qmm_op_add_rms_norm: operands not supported");
    HIP_TRY(hipSetDevice(ctx->device));
    QMM_CHAIN_FLUSH(ctx);
    return launch_rms_norm(ctx->s(stream), a, b, w, dst, sum, eps);
}

static bool attn_short_on() {                            // GGML_MI355X_ATTN_SHORT=0: the general kernel at every n_kv (A/B runs)
    static const bool on = [] { const char * e = getenv("GGML_MI355X_ATTN_SHORT"); return !(e && atoi(e) == 0); }();
    return on;
}

int qmm_attn_decode_supported(const qmm_tensor * q, const qmm_tensor * k, const qmm_tensor * v, const qmm_tensor * mask, const qmm_tensor * dst) {
    if (!q || !k || !v || !mask || !dst) return 0;
    if (q->type != G_F32 || k->type != G_F16 || v->type != G_F16 || mask->type != G_F32 || dst->type != G_F32) return 0;
    const int64_t D = k->ne[0], n_kv = k->ne[1], Hk = k->ne[2], N = q->ne[1], H = q->ne[2], Dv = v->ne[1];
    if (D != 64 && D != 128 && D != 256) return 0;
    if (q->ne[0] != D || v->ne[0] != n_kv || v->ne[2] != Hk || Hk <= 0 || H % Hk || q->ne[3] != 1 || k->ne[3] != 1 || v->ne[3] != 1) return 0;
    if (N < 1 || N > 8 || n_kv < 8 || n_kv % 8 || n_kv > 16384 || Dv < 1 || Dv > 1024 || H > 65535) return 0;
    if (q->nb[0] != 4 || k->nb[0] != 2 || v->nb[0] != 2 || mask->nb[0] != 4 || dst->nb[0] != 4) return 0;
    if (mask->ne[0] != n_kv || mask->ne[1] < N || dst->ne[0] != Dv * H || dst->ne[1] != N || dst->ne[2] != 1 || dst->ne[3] != 1) return 0;
    if (q->nb[1] % 4 || q->nb[2] % 4 || k->nb[1] % 16 || k->nb[2] % 16 || v->nb[1] % 16 || v->nb[2] % 16 || mask->nb[1] % 4 || dst->nb[1] % 4) return 0;
    return 1;
}

int qmm_attn_decode(qmm_ctx * ctx, const qmm_tensor * q, const qmm_tensor * k, const qmm_tensor * v, const qmm_tensor * mask, const qmm_tensor * dst,
                    float scale, void * stream) {
    if (!ctx || !qmm_attn_decode_supported(q, k, v, mask, dst)) return fail(QMM_EUNSUPPORTED, "qmm_attn_decode: operands not supported");
    if ((uintptr_t) k->data % 16 || (uintptr_t) v->data % 16) return fail(QMM_EINVAL, "qmm_attn_decode: K / V must be 16-byte aligned");
    HIP_TRY(hipSetDevice(ctx->device));
    QMM_CHAIN_FLUSH(ctx);
    AttnArgs g;
    g.q = (const char *) q->data; g.k = (const char *) k->data; g.v = (const char *) v->data; g.mask = (const char *) mask->data; g.dst = (char *) dst->data;
    g.q_nb1 = q->nb[1]; g.q_nb2 = q->nb[2]; g.k_nb1 = k->nb[1]; g.k_nb2 = k->nb[2]; g.v_nb1 = v->nb[1]; g.v_nb2 = v->nb[2];
    g.m_nb1 = mask->nb[1]; g.d_nb1 = dst->nb[1];
    g.D = (int32_t) k->ne[0]; g.Dv = (int32_t) v->ne[1]; g.n_kv = (int32_t) k->ne[1]; g.H = (int32_t) q->ne[2]; g.gqa = (int32_t) (q->ne[2] / k->ne[2]);
    g.scale = scale;
    const dim3 grid((unsigned) g.H, (unsigned) q->ne[1]);
    const size_t lds = (size_t) g.n_kv * 4;
    hipStream_t st = ctx->s(stream);
    static const bool split_on = [] { const char * e = getenv("GGML_MI355X_ATTN_SPLIT"); return !(e && atoi(e) == 0); }();
    if (split_on && g.n_kv >= 1024 && g.D <= 128) {
        // kv range over S workgroups per (head, token), then the merge (long caches: one workgroup per head is latency-bound)
        const int S = g.n_kv / 256 < 16 ? g.n_kv / 256 : 16;
        const int chunk = ((g.n_kv + S - 1) / S + 7) / 8 * 8;
        const int N = (int) q->ne[1];
        const size_t bytes = (size_t) g.H * N * S * (g.Dv + 2) * sizeof(float);
        int rc = ensure_ws(ctx, bytes);
        if (rc) return rc;
        float * part = (float *) ctx->ws;
        const dim3 sgrid((unsigned) g.H, (unsigned) N, (unsigned) S);
        if (g.D == 64) hipLaunchKernelGGL((attn_decode_split_kernel<64>), sgrid, dim3(1024), (size_t) chunk * 4, st, g, part, S, chunk);
        else           hipLaunchKernelGGL((attn_decode_split_kernel<128>), sgrid, dim3(1024), (size_t) chunk * 4, st, g, part, S, chunk);
        hipLaunchKernelGGL(attn_combine_kernel, dim3((unsigned) g.H, (unsigned) N), dim3(256), 0, st, (const float *) part, g.dst, g.d_nb1, g.Dv, S);
        HIP_TRY(hipGetLastError());
        return QMM_OK;
    }
    const AttnFresh none{};
    if (attn_short_on() && g.n_kv <= 1024 && g.D <= 128 && g.Dv <= 128) {
        const size_t lds2 = (size_t) g.n_kv * 8;
        const int w = g.n_kv <= 256 ? 0 : g.n_kv <= 512 ? 1 : 2;
        auto k64  = w == 0 ? attn_decode_short_kernel<64, false, 256> : w == 1 ? attn_decode_short_kernel<64, false, 512> : attn_decode_short_kernel<64, false, 1024>;
        auto k128 = w == 0 ? attn_decode_short_kernel<128, false, 256> : w == 1 ? attn_decode_short_kernel<128, false, 512> : attn_decode_short_kernel<128, false, 1024>;
        if (g.D == 64) hipLaunchKernelGGL(k64, grid, dim3(1024), lds2, st, g, none);
        else           hipLaunchKernelGGL(k128, grid, dim3(1024), lds2, st, g, none);
        HIP_TRY(hipGetLastError());
        return QMM_OK;
    }
    if (g.D == 64) {
        hipLaunchKernelGGL((attn_decode_kernel<64, false>), grid, dim3(1024), lds, st, g, none);
    } else if (g.D == 128) {
        hipLaunchKernelGGL((attn_decode_kernel<128, false>), grid, dim3(1024), lds, st, g, none);
    } else {
        hipLaunchKernelGGL((attn_decode_kernel<256, false>), grid, dim3(1024), lds, st, g, none);
    }
    HIP_TRY(hipGetLastError());
    return QMM_OK;
}

int qmm_rope_kv_store_supported(const qmm_tensor * q, const qmm_tensor * pos, const qmm_tensor * ff, const qmm_tensor * q_dst,
                                const qmm_tensor * k, const qmm_tensor * k_dst, const qmm_tensor * v, const qmm_tensor * v_dst) {
    if (!sup_rope(q, pos, ff, q_dst)) return 0;
    if (k) {
        if (!k_dst || k->type != G_F32 || k_dst->type != G_F16 || !same_shape(k, k_dst) || !dense_rows(k) || !dense_rows(k_dst) || !fits_u32(k)) return 0;
        if (k->ne[0] != q->ne[0] || k->ne[2] != q->ne[2] || k->ne[3] != q->ne[3] || nelements(k) == 0) return 0;
        if (k->nb[1] % 4 || k->nb[2] % 4 || k->nb[3] % 4 || k_dst->nb[1] % 2 || k_dst->nb[2] % 2 || k_dst->nb[3] % 2) return 0;
    }
    if (v) {
        if (!v_dst || v->type != G_F32 || v_dst->type != G_F16 || nelements(v) != nelements(v_dst) || !fits_u32(v) || nelements(v) == 0) return 0;
        if (v->nb[0] % 4 || v->nb[1] % 4 || v->nb[2] % 4 || v->nb[3] % 4 || v_dst->nb[0] % 2 || v_dst->nb[1] % 2 || v_dst->nb[2] % 2 || v_dst->nb[3] % 2) return 0;
    }
    return nelements(q) / 2 + (k ? nelements(k) / 2 : 0) + (v ? nelements(v) : 0) < ((int64_t) 1 << 31);
}

int qmm_rope_kv_store(qmm_ctx * ctx, const qmm_tensor * q, const qmm_tensor * pos, const qmm_tensor * ff, const qmm_tensor * q_dst,
                      const qmm_tensor * k, const qmm_tensor * k_dst, const qmm_tensor * v, const qmm_tensor * v_dst, void * stream) {
    if (!ctx || !qmm_rope_kv_store_supported(q, pos, ff, q_dst, k, k_dst, v, v_dst)) return fail(QMM_EUNSUPPORTED, "qmm_rope_kv_store: operands not supported");
    HIP_TRY(hipSetDevice(ctx->device));
    QMM_CHAIN_FLUSH(ctx);
    RopeStoreArgs g;
    g.q = (const char *) q->data; g.qd = (char *) q_dst->data; g.sq = shape_of(q); g.sqd = shape_of(q_dst);
    g.k = k ? (const char *) k->data : nullptr; g.kd = k ? (char *) k_dst->data : nullptr; g.sk = shape_of(k ? k : q); g.skd = shape_of(k ? k_dst : q_dst);
    g.v = v ? (const char *) v->data : nullptr; g.vd = v ? (char *) v_dst->data : nullptr; g.sv = shape_of(v ? v : q); g.svd = shape_of(v ? v_dst : q_dst);
    g.pos = (const int32_t *) pos->data; g.ff = ff ? (const float *) ff->data : nullptr;
    g.rp = rope_params(q_dst);
    g.nq = rope_heads_threads(g.sq); g.nk = k ? rope_heads_threads(g.sk) : 0; g.nv = v ? (uint32_t) nelements(v) : 0;
    const bool vt = v && v->ne[2] == 1 && v->ne[3] == 1 && v_dst->ne[2] == 1 && v_dst->ne[3] == 1 && v->ne[0] == v_dst->ne[0] && v->ne[1] == v_dst->ne[1] &&
                    v->nb[1] == 4 && v_dst->nb[0] == 2 && v->ne[0] >= 32 && v->ne[1] >= 32;
    if (vt) {
        const uint32_t pair_blocks = (g.nq + g.nk + 255) / 256, tiles0 = (uint32_t) ((v->ne[0] + 31) / 32), tiles1 = (uint32_t) ((v->ne[1] + 31) / 32);
        hipLaunchKernelGGL((rope_store_kernel<true>), dim3(pair_blocks + tiles0 * tiles1), dim3(256), 0, ctx->s(stream), g, pair_blocks, tiles0);
    } else {
        const uint32_t total = g.nq + g.nk + g.nv;
        hipLaunchKernelGGL((rope_store_kernel<false>), dim3((total + 255) / 256), dim3(256), 0, ctx->s(stream), g, 0u, 0u);
    }
    HIP_TRY(hipGetLastError());
    return QMM_OK;
}

int qmm_attn_prefill_supported(const qmm_tensor * q, const qmm_tensor * k, const qmm_tensor * v, const qmm_tensor * mask, const qmm_tensor * dst) {
    if (!q || !k || !v || !mask || !dst) return 0;
    if (q->type != G_F32 || k->type != G_F16 || v->type != G_F16 || mask->type != G_F32 || dst->type != G_F32) return 0;
    const int64_t D = k->ne[0], n_kv = k->ne[1], Hk = k->ne[2], N = q->ne[1], H = q->ne[2], Dv = v->ne[1];
    if ((D != 64 && D != 128) || Dv != D) return 0;
    if (q->ne[0] != D || v->ne[0] != n_kv || v->ne[2] != Hk || Hk <= 0 || H % Hk || q->ne[3] != 1 || k->ne[3] != 1 || v->ne[3] != 1) return 0;
    if (N < 1 || N > (1 << 20) || n_kv < 32 || n_kv % 32 || n_kv > (1 << 20) || H > 65535) return 0;
    if (q->nb[0] != 4 || k->nb[0] != 2 || v->nb[0] != 2 || mask->nb[0] != 4 || dst->nb[0] != 4) return 0;
    if (mask->ne[0] != n_kv || mask->ne[1] < N || dst->ne[0] != Dv * H || dst->ne[1] != N || dst->ne[2] != 1 || dst->ne[3] != 1) return 0;
    if (q->nb[1] % 16 || q->nb[2] % 16 || k->nb[1] % 16 || k->nb[2] % 16 || v->nb[1] % 16 || v->nb[2] % 16 || mask->nb[1] % 4 || dst->nb[1] % 4) return 0;
    return 1;
}

int qmm_attn_prefill(qmm_ctx * ctx, const qmm_tensor * q, const qmm_tensor * k, const qmm_tensor * v, const qmm_tensor * mask, const qmm_tensor * dst,
                     float scale, void * stream) {
    if (!ctx || !qmm_attn_prefill_supported(q, k, v, mask, dst)) return fail(QMM_EUNSUPPORTED, "qmm_attn_prefill: operands not supported");
    if ((uintptr_t) k->data % 16 || (uintptr_t) v->data % 16 || (uintptr_t) q->data % 16) return fail(QMM_EINVAL, "qmm_attn_prefill: q / K / V must be 16-byte aligned");
    HIP_TRY(hipSetDevice(ctx->device));
    QMM_CHAIN_FLUSH(ctx);
    AttnArgs g;
    g.q = (const char *) q->data; g.k = (const char *) k->data; g.v = (const char *) v->data; g.mask = (const char *) mask->data; g.dst = (char *) dst->data;
    g.q_nb1 = q->nb[1]; g.q_nb2 = q->nb[2]; g.k_nb1 = k->nb[1]; g.k_nb2 = k->nb[2]; g.v_nb1 = v->nb[1]; g.v_nb2 = v->nb[2];
    g.m_nb1 = mask->nb[1]; g.d_nb1 = dst->nb[1];
    g.D = (int32_t) k->ne[0]; g.Dv = (int32_t) v->ne[1]; g.n_kv = (int32_t) k->ne[1]; g.H = (int32_t) q->ne[2]; g.gqa = (int32_t) (q->ne[2] / k->ne[2]);
    g.scale = scale;
    const int N = (int) q->ne[1];
    const dim3 grid((unsigned) ((N + AP_TN - 1) / AP_TN), (unsigned) g.H);
    const size_t tile = (size_t) 128 * (AP_KT + 8) * 2;                      // >= K tile 64 x (D + 8) and V tile D x (64 + 8) halves, D <= 128
    const size_t lds = (size_t) AP_TN * ((g.n_kv < AP_CH ? g.n_kv : AP_CH) + 4) * 4 + tile;
    hipStream_t st = ctx->s(stream);
    if (g.D == 128) {
        auto kern = attn_prefill_kernel<128>;
        if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, g, N);
    } else {
        auto kern = attn_prefill_kernel<64>;
        if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, g, N);
    }
    HIP_TRY(hipGetLastError());
    return QMM_OK;
}

int qmm_attn_decode_rope_supported(const qmm_tensor * q, const qmm_tensor * pos, const qmm_tensor * ff, const qmm_tensor * q_rope,
                                   const qmm_tensor * k_new, const qmm_tensor * k_store, const qmm_tensor * v_new, const qmm_tensor * v_store,
                                   const qmm_tensor * k, const qmm_tensor * v, const qmm_tensor * mask, const qmm_tensor * dst, int64_t j0) {
    if (!q || !pos || !q_rope || !k_new || !k_store || !v_new || !v_store || !k || !v || !mask || !dst) return 0;
    // q [D, H, N] un-roped (q_rope: the ROPE node, for its op_params); the permuted view of it is what qmm_attn_decode takes
    qmm_tensor qp = *q;
    qp.ne[1] = q->ne[2]; qp.ne[2] = q->ne[1]; qp.nb[1] = q->nb[2]; qp.nb[2] = q->nb[1];
    if (!qmm_attn_decode_supported(&qp, k, v, mask, dst)) return 0;
    if (!qmm_rope_kv_store_supported(q, pos, ff, q_rope, k_new, k_store, v_new, v_store)) return 0;
    const int64_t D = k->ne[0], N = q->ne[2], Hk = k->ne[2], Dv = v->ne[1], n_kv = k->ne[1];
    if (q_rope->op_params[2] != 0 || q_rope->op_params[1] != D || D > 128) return 0;                 // normal mode over the whole head
    if (k_new->ne[0] != D || k_new->ne[1] != Hk || k_new->ne[2] != N || k_new->ne[3] != 1 || k_new->nb[0] != 4) return 0;
    if (k_store->nb[0] != 2 || k_store->nb[1] % 2 || k_store->nb[2] % 2) return 0;
    // v_new is v_cur^T [N, Dv * Hk] (element (n, c) at n * nb0 + c * nb1), v_store the transposed cache view [N, Dv * Hk]
    if (v_new->ne[0] != N || v_new->ne[1] != Dv * Hk || v_new->ne[2] != 1 || v_new->ne[3] != 1 || v_new->nb[1] != 4) return 0;
    if (v_store->ne[0] != N || v_store->ne[1] != Dv * Hk || v_store->nb[0] != 2 || v_store->nb[1] % 2) return 0;
    if (j0 < 0 || j0 + N > n_kv) return 0;
    return 1;
}

int qmm_attn_decode_rope(qmm_ctx * ctx, const qmm_tensor * q, const qmm_tensor * pos, const qmm_tensor * ff, const qmm_tensor * q_rope,
                         const qmm_tensor * k_new, const qmm_tensor * k_store, const qmm_tensor * v_new, const qmm_tensor * v_store,
                         const qmm_tensor * k, const qmm_tensor * v, const qmm_tensor * mask, const qmm_tensor * dst, float scale, int64_t j0,
                         void * stream) {
    if (!ctx || !qmm_attn_decode_rope_supported(q, pos, ff, q_rope, k_new, k_store, v_new, v_store, k, v, mask, dst, j0))
        return fail(QMM_EUNSUPPORTED, "qmm_attn_decode_rope: operands not supported");
    if ((uintptr_t) k->data % 16 || (uintptr_t) v->data % 16) return fail(QMM_EINVAL, "qmm_attn_decode_rope: K / V must be 16-byte aligned");
    HIP_TRY(hipSetDevice(ctx->device));
    QMM_CHAIN_FLUSH(ctx);
    AttnArgs g;
    g.q = (const char *) q->data; g.k = (const char *) k->data; g.v = (const char *) v->data; g.mask = (const char *) mask->data; g.dst = (char *) dst->data;
    g.q_nb1 = q->nb[2]; g.q_nb2 = q->nb[1];                                  // token stride, head stride of the un-permuted q
    g.k_nb1 = k->nb[1]; g.k_nb2 = k->nb[2]; g.v_nb1 = v->nb[1]; g.v_nb2 = v->nb[2];
    g.m_nb1 = mask->nb[1]; g.d_nb1 = dst->nb[1];
    g.D = (int32_t) k->ne[0]; g.Dv = (int32_t) v->ne[1]; g.n_kv = (int32_t) k->ne[1]; g.H = (int32_t) q->ne[1]; g.gqa = (int32_t) (q->ne[1] / k->ne[2]);
    g.scale = scale;
    AttnFresh f;
    f.kraw = (const char *) k_new->data; f.vraw = (const char *) v_new->data; f.kd = (char *) k_store->data; f.vd = (char *) v_store->data;
    f.pos = (const int32_t *) pos->data; f.ff = ff ? (const float *) ff->data : nullptr;
    f.kraw_nbh = k_new->nb[1]; f.kraw_nbn = k_new->nb[2]; f.vraw_nbn = v_new->nb[0];
    f.kd_nbh = k_store->nb[1]; f.kd_nbn = k_store->nb[2]; f.vd_nbc = v_store->nb[1];
    f.rp = rope_params(q_rope);
    f.N = (int32_t) q->ne[2]; f.j0 = (int32_t) j0;
    const dim3 grid((unsigned) g.H, (unsigned) f.N);
    const size_t lds = (size_t) g.n_kv * 4 + (size_t) g.D * 4 + (size_t) f.N * (g.D + g.Dv) * 2;
    hipStream_t st = ctx->s(stream);
    if (attn_short_on() && g.n_kv <= 1024 && g.Dv <= 128) {            // (D <= 128: qmm_attn_decode_rope_supported)
        const size_t lds2 = lds + (size_t) g.n_kv * 4;
        const int w = g.n_kv <= 256 ? 0 : g.n_kv <= 512 ? 1 : 2;
        auto k64  = w == 0 ? attn_decode_short_kernel<64, true, 256> : w == 1 ? attn_decode_short_kernel<64, true, 512> : attn_decode_short_kernel<64, true, 1024>;
        auto k128 = w == 0 ? attn_decode_short_kernel<128, true, 256> : w == 1 ? attn_decode_short_kernel<128, true, 512> : attn_decode_short_kernel<128, true, 1024>;
        if (g.D == 64) hipLaunchKernelGGL(k64, grid, dim3(1024), lds2, st, g, f);
        else           hipLaunchKernelGGL(k128, grid, dim3(1024), lds2, st, g, f);
        HIP_TRY(hipGetLastError());
        return QMM_OK;
    }
    if (g.D == 64) {
        auto kern = attn_decode_kernel<64, true>;
        if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
        hipLaunchKernelGGL(kern, grid, dim3(1024), lds, st, g, f);
    } else {
        auto kern = attn_decode_kernel<128, true>;
        if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
        hipLaunchKernelGGL(kern, grid, dim3(1024), lds, st, g, f);
    }
    HIP_TRY(hipGetLastError());
    return QMM_OK;
}

int qmm_moe_router_supported(const qmm_tensor * logits, const qmm_tensor * ids, const qmm_tensor * weights, int64_t n_used) {
    if (!logits || !ids || !weights || logits->type != G_F32 || ids->type != G_I32 || weights->type != G_F32) return 0;
    const int64_t E = logits->ne[0], N = logits->ne[1];
    if (E < 1 || E > 64 || n_used < 1 || n_used > E || N < 1 || N >= ((int64_t) 1 << 30)) return 0;
    if (logits->ne[2] != 1 || logits->ne[3] != 1 || ids->ne[0] != E || ids->ne[1] != N || ids->ne[2] != 1 || ids->ne[3] != 1) return 0;
    if (weights->ne[0] * weights->ne[1] * weights->ne[2] * weights->ne[3] != n_used * N) return 0;
    return logits->nb[0] == 4 && ids->nb[0] == 4 && weights->nb[0] == 4 && logits->nb[1] % 4 == 0 && ids->nb[1] % 4 == 0;
}

int qmm_moe_router(qmm_ctx * ctx, const qmm_tensor * logits, const qmm_tensor * ids, const qmm_tensor * weights, int64_t n_used, int normalise,
                   void * stream) {
    if (!ctx || !qmm_moe_router_supported(logits, ids, weights, n_used)) return fail(QMM_EUNSUPPORTED, "qmm_moe_router: operands not supported");
    HIP_TRY(hipSetDevice(ctx->device));
    QMM_CHAIN_FLUSH(ctx);
    const int N = (int) logits->ne[1];
    hipLaunchKernelGGL(moe_router_kernel, dim3((N + 3) / 4), dim3(256), 0, ctx->s(stream), (const char *) logits->data, (char *) ids->data,
                       (char *) weights->data, logits->nb[1], ids->nb[1], (int64_t) n_used * 4, (int) logits->ne[0], (int) n_used, N, normalise);
    HIP_TRY(hipGetLastError());
    return QMM_OK;
}

int qmm_moe_router_logits_supported(const qmm_tensor * gate_inp, const qmm_tensor * x, const qmm_tensor * logits, const qmm_tensor * ids,
                                    const qmm_tensor * weights, int64_t n_used) {
    if (!gate_inp || !x || !qmm_moe_router_supported(logits, ids, weights, n_used)) return 0;
    if (gate_inp->type != G_F32 || x->type != G_F32) return 0;
    const int64_t K = gate_inp->ne[0], E = gate_inp->ne[1], N = x->ne[1];
    // (more tokens: the tiled MUL_MAT + moe_router_kernel; K < 1024: qmm_op(MUL_MAT) takes a wave per element there, another order of additions)
    if (K < 1024 || K >= ((int64_t) 1 << 30) || E != logits->ne[0] || N != logits->ne[1] || N > 8 || x->ne[0] != K) return 0;
    if (gate_inp->ne[2] != 1 || gate_inp->ne[3] != 1 || x->ne[2] != 1 || x->ne[3] != 1) return 0;
    return gate_inp->nb[0] == 4 && x->nb[0] == 4 && gate_inp->nb[1] % 4 == 0 && x->nb[1] % 4 == 0;
}

int qmm_moe_router_logits(qmm_ctx * ctx, const qmm_tensor * gate_inp, const qmm_tensor * x, const qmm_tensor * logits, const qmm_tensor * ids,
                          const qmm_tensor * weights, int64_t n_used, int normalise, void * stream) {
    if (!ctx || !qmm_moe_router_logits_supported(gate_inp, x, logits, ids, weights, n_used))
        return fail(QMM_EUNSUPPORTED, "qmm_moe_router_logits: operands not supported");
    HIP_TRY(hipSetDevice(ctx->device));
    QMM_CHAIN_FLUSH(ctx);
    hipLaunchKernelGGL(moe_router_logits_kernel<false>, dim3((unsigned) x->ne[1]), dim3(1024), 0, ctx->s(stream), (const char *) gate_inp->data, (const char *) x->data,
                       (char *) logits->data, (char *) ids->data, (char *) weights->data, gate_inp->nb[1], x->nb[1], logits->nb[1], ids->nb[1], (int64_t) n_used * 4,
                       (int) gate_inp->ne[0], (int) logits->ne[0], (int) n_used, normalise, nullptr, 0.0f, nullptr, 0);
    HIP_TRY(hipGetLastError());
    return QMM_OK;
}

int qmm_moe_router_logits_norm_supported(const qmm_tensor * gate_inp, const qmm_tensor * x, const qmm_tensor * norm_w, const qmm_tensor * normed,
                                         const qmm_tensor * logits, const qmm_tensor * ids, const qmm_tensor * weights, int64_t n_used) {
    if (!norm_w || !normed || !qmm_moe_router_logits_supported(gate_inp, x, logits, ids, weights, n_used)) return 0;
    const int64_t K = x->ne[0];
    if (norm_w->type != G_F32 || normed->type != G_F32 || K % 4 || K > 16384) return 0;
    if (norm_w->ne[0] != K || norm_w->ne[1] * norm_w->ne[2] * norm_w->ne[3] != 1 || norm_w->nb[0] != 4) return 0;
    for (int i = 0; i < 4; ++i) if (normed->ne[i] != x->ne[i]) return 0;
    if (normed->nb[0] != 4 || x->nb[1] % 16 || normed->nb[1] % 16 || gate_inp->nb[1] % 16) return 0;
    return (uintptr_t) x->data % 16 == 0 && (uintptr_t) normed->data % 16 == 0 && (uintptr_t) norm_w->data % 16 == 0 && (uintptr_t) gate_inp->data % 16 == 0;
}

int qmm_moe_router_logits_norm(qmm_ctx * ctx, const qmm_tensor * gate_inp, const qmm_tensor * x, const qmm_tensor * norm_w, float eps, const qmm_tensor * normed,
                               const qmm_tensor * logits, const qmm_tensor * ids, const qmm_tensor * weights, int64_t n_used, int normalise, void * stream) {
    if (!ctx || eps < 0.0f || !qmm_moe_router_logits_norm_supported(gate_inp, x, norm_w, normed, logits, ids, weights, n_used))
        return fail(QMM_EUNSUPPORTED, "qmm_moe_router_logits_norm: operands not supported");
    HIP_TRY(hipSetDevice(ctx->device));
    QMM_CHAIN_FLUSH(ctx);
    hipLaunchKernelGGL(moe_router_logits_kernel<true>, dim3((unsigned) x->ne[1]), dim3(1024), (size_t) x->ne[0] * 4, ctx->s(stream), (const char *) gate_inp->data,
                       (const char *) x->data, (char *) logits->data, (char *) ids->data, (char *) weights->data, gate_inp->nb[1], x->nb[1], logits->nb[1], ids->nb[1],
                       (int64_t) n_used * 4, (int) gate_inp->ne[0], (int) logits->ne[0], (int) n_used, normalise, (const float *) norm_w->data, eps,
                       (char *) normed->data, normed->nb[1]);
    HIP_TRY(hipGetLastError());
    return QMM_OK;
}

int qmm_moe_combine_supported(const qmm_tensor * x, const qmm_tensor * w, const qmm_tensor * out) {
    if (!x || !w || !out || x->type != G_F32 || w->type != G_F32 || out->type != G_F32) return 0;
    const int64_t E = x->ne[0], U = x->ne[1], N = x->ne[2];
    if (E < 4 || E % 4 || U < 1 || U > 64 || N < 1 || N > 65535 || x->ne[3] != 1) return 0;
    if (w->ne[0] != 1 || w->ne[1] != U || w->ne[2] != N || w->ne[3] != 1 || out->ne[0] != E || out->ne[1] != N || out->ne[2] != 1 || out->ne[3] != 1) return 0;
    if (x->nb[0] != 4 || out->nb[0] != 4 || x->nb[1] % 16 || x->nb[2] % 16 || out->nb[1] % 16 || w->nb[1] % 4 || w->nb[2] % 4) return 0;
    return 1;
}

int qmm_moe_combine(qmm_ctx * ctx, const qmm_tensor * x, const qmm_tensor * w, const qmm_tensor * out, void * stream) {
    if (!ctx || !qmm_moe_combine_supported(x, w, out)) return fail(QMM_EUNSUPPORTED, "qmm_moe_combine: operands not supported");
    if ((uintptr_t) x->data % 16 || (uintptr_t) out->data % 16) return fail(QMM_EINVAL, "qmm_moe_combine: x / out must be 16-byte aligned");
    HIP_TRY(hipSetDevice(ctx->device));
    QMM_CHAIN_FLUSH(ctx);
    const int E = (int) x->ne[0];
    hipLaunchKernelGGL(moe_combine_kernel, dim3((unsigned) ((E / 4 + 255) / 256), (unsigned) x->ne[2]), dim3(256), 0, ctx->s(stream), (const char *) x->data,
                       (const char *) w->data, (char *) out->data, x->nb[1], x->nb[2], w->nb[1], w->nb[2], out->nb[1], E, (int) x->ne[1]);
    HIP_TRY(hipGetLastError());
    return QMM_OK;
}

int qmm_moe_combine_add_rms_norm_supported(const qmm_tensor * x, const qmm_tensor * w, const qmm_tensor * b, const qmm_tensor * nw, const qmm_tensor * sum,
                                           const qmm_tensor * dst) {
    if (!b || !nw || !sum || !dst || !qmm_moe_combine_supported(x, w, sum)) return 0;
    const int64_t E = x->ne[0], N = x->ne[2];
    if (E > 1024 * 16 || b->type != G_F32 || nw->type != G_F32 || dst->type != G_F32) return 0;
    for (const qmm_tensor * t : { b, sum, dst })
        if (t->ne[0] != E || t->ne[1] != N || t->ne[2] != 1 || t->ne[3] != 1 || t->nb[0] != 4 || t->nb[1] % 16 || (uintptr_t) t->data % 16) return 0;
    return nw->ne[0] == E && nw->ne[1] * nw->ne[2] * nw->ne[3] == 1 && nw->nb[0] == 4 && (uintptr_t) nw->data % 16 == 0 && (uintptr_t) x->data % 16 == 0;
}

int qmm_moe_combine_add_rms_norm(qmm_ctx * ctx, const qmm_tensor * x, const qmm_tensor * w, const qmm_tensor * b, const qmm_tensor * nw, const qmm_tensor * sum,
                                 const qmm_tensor * dst, float eps, void * stream) {
    if (!ctx || !qmm_moe_combine_add_rms_norm_supported(x, w, b, nw, sum, dst)) return fail(QMM_EUNSUPPORTED, "qmm_moe_combine_add_rms_norm: operands not supported");
    if (eps < 0.0f) return fail(QMM_EINVAL, "qmm_moe_combine_add_rms_norm: eps < 0");
    HIP_TRY(hipSetDevice(ctx->device));
    QMM_CHAIN_FLUSH(ctx);
    const int E = (int) x->ne[0], N = (int) x->ne[2];
    const bool wide = N < 256 || E > 256 * 16;              // launch_rms_norm_vec's rule: the same partition as the stand-alone kernel
    hipStream_t st = ctx->s(stream);
    if (wide) hipLaunchKernelGGL((moe_combine_add_norm_kernel<1024>), dim3((unsigned) N), dim3(1024), 0, st, (const char *) x->data, (const char *) w->data, (const char *) b->data,
                                 (const float *) nw->data, (char *) dst->data, (char *) sum->data, x->nb[1], x->nb[2], w->nb[1], w->nb[2], b->nb[1], dst->nb[1], sum->nb[1], E,
                                 (int) x->ne[1], eps);
    else      hipLaunchKernelGGL((moe_combine_add_norm_kernel<256>), dim3((unsigned) N), dim3(256), 0, st, (const char *) x->data, (const char *) w->data, (const char *) b->data,
                                 (const float *) nw->data, (char *) dst->data, (char *) sum->data, x->nb[1], x->nb[2], w->nb[1], w->nb[2], b->nb[1], dst->nb[1], sum->nb[1], E,
                                 (int) x->ne[1], eps);
    HIP_TRY(hipGetLastError());
    return QMM_OK;
}

} // extern "C"
