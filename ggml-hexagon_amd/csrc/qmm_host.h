// qmm_host.h — host-side context and error plumbing shared by the launchers.
#pragma once

#include "../../include/ggml_mi355x_qmm.h"

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace qmm { struct ChainStep; struct ChainSync; }

struct qmm_ctx {
    int         device = 0;
    hipStream_t stream = nullptr;
    int         cus = 0;
    int         act_mode = QMM_ACT_REF;
    int         prec = QMM_PREC_F16_Q8;
    int         mv_bpc = 1;          // mat-vec blocks per CU (tuning knob, GGML_MI355X_MV_BPC)
    int         mv_kmix = 2;         // one mixed-type mat-vec launch for K-quant groups of different types (GGML_MI355X_MV_KMIX=0: off; 1: not for groups that also hold Q8_0 matrices)
    int         mm_group = 1;        // prefill: one tiled launch for same-type matrices that share src1 (GGML_MI355X_MM_GROUP=0: off)
    int         skinny = 1;          // few-token split-K MFMA kernel (GGML_MI355X_SKINNY=0 turns it off)
    int         skinny_max_n = 64;   // ... used for 8 < N <= this, and up to skinny_max_n_few when the matrix has no more
    int         skinny_max_n_few = 128;  //     32-row groups than the chip has CUs (GGML_MI355X_SKINNY_MAXN sets both)
    // workspace of the batched path (16-bit activations, row scales, MoE lists); grown on demand
    void *      ws = nullptr;
    size_t      ws_bytes = 0;
    size_t      ws_base = 0, ws_used = 0;   // prefill runs that are in flight together use disjoint slices: a run works at ws + ws_base and reports what it took
    // Runs of a prefill group that need their own activation prep (another weight format: attn_v in Q6_K beside attn_q / attn_k in
    // Q4_K) go out on side streams between a fork and a join event: their prep / MFMA / reduce launches, each too small to fill the
    // chip, overlap the first run's instead of queueing behind them (GGML_MI355X_SIDE=0: one stream)
    int         side_on = 0;             // (measured on llama3-8b q/k/v with attn_v in Q6_K: 74 us one stream, 78.5 us with the Q6_K run on a side stream: the two events cost more than the overlap wins)
    hipStream_t side[3] = { nullptr, nullptr, nullptr };
    hipEvent_t  ev_fork = nullptr, ev_join[3] = { nullptr, nullptr, nullptr };
    int         splitk_combine = 0;  // GGML_MI355X_SPLITK_COMBINE=1: split-K ranges combined inside the launch (splitk_finish_wave) instead of by splitk_reduce_kernel.
                                     // Bit-identical, measured a wash on llama3-8b at 512 tokens (wo 46.1 -> 43.9 us, q/k/v 72.4 -> 73.0, ffn_down 105.5 -> 107.1): the
                                     // slabs' traffic stays, and the last arriver's four dependent read trips cost what the reduce launch cost.  Opt-in.
    int *       kcnt = nullptr;      // its arrival counters (zero between launches), one per (row tile, token tile, wave)
    int64_t     kcnt_n = 0;
    int *       flag = nullptr;      // device word set by kernels that meet an expert id out of range
    const float * prep_x2 = nullptr; // transient: second operand of a SwiGLU input while qmm_mul_mat_swiglu_in runs (prefill prep)
    int64_t     prep_ldx2 = 0;
    // transient: while qmm_mul_mat_group_ex runs a prefill group with extra->norm_w, the activation prep forms rms_norm(x [+ add]) * w itself
    // (prep_act_q8k_kernel<..., NORM>); behind the first prep of the call x is the stored sum, read as it is
    struct prep_norm_t { const float * w = nullptr; float eps = 0.0f; const float * add = nullptr; int64_t ld_add = 0; float * sum = nullptr; int64_t ld_sum = 0;
                         const float * x_over = nullptr; int64_t ld_over = 0; } prep_norm;
    int64_t     id_calls = 0, id_checked = 0;       // MUL_MAT_ID launches issued / covered by the last look at `flag` (qmm_synchronize reads the word only behind such a launch: a blocking 4-byte copy per synchronize cost llama.cpp's token loop ~70 us per token)
    int64_t     mfma_calls = 0, mfma_checked = 0;   // prefill calls issued / covered by the last non-finite check (qmm_synchronize)
    bool        wide_attr_set = false;
    int         wide = 1;            // 256-token tiles for large Q4_K prefill launches (GGML_MI355X_WIDE=0: off)
    int         r64 = 2;             // Q4_K prefill, 64 rows per wave + LDS DMA (mfma_r64_q4k_kernel): bit 0 = in place of the 256 x 128 kernel, bit 1 = in place of the 256 x 256 one (GGML_MI355X_R64)
    int         r64s = 1;            // ... with the K-step's instruction stream placed by hand (mfma_r64s_q4k_kernel, qmm_mfma_r64s.hiph) where the 256 x 256 r64 kernel would run (GGML_MI355X_R64S=0: the compiler-scheduled kernel)
    int         mv_onepass = 1;      // GGML_MI355X_MV_ONEPASS=0: fused-norm mat-vecs stage through stage_rms_norm + quantize_rows again (A/B runs)
    int         prep_reg = 1;        // GGML_MI355X_PREP_REG=0: prep_act_kernel (LDS staging) for Q8_K rows too (A/B runs, parity tests)
    int         regb_q23 = 1;        // GGML_MI355X_REGB_Q23=0: Q2_K / Q3_K prefill on the LDS-tile kernel again (A/B runs)
    int         splitk = 1;          // split K over workgroups when a MUL_MAT has too few tiles (GGML_MI355X_SPLITK=0: off)
    // chains (qmm_chain.hiph): while recording, one-token MUL_MAT groups are collected instead of launched
    int         chain_enabled = 1;   // GGML_MI355X_CHAIN=0: qmm_chain_begin records nothing, every group is its own launch
    bool        chain_on = false;
    std::vector<qmm::ChainStep> * chain = nullptr;
    qmm::ChainSync * chain_sync = nullptr;      // device: arrival counters, generation, error word
    int         chain_launches = 0;  // persistent launches issued so far (tests / bench read it through qmm_chain_stats)
    int         chain_steps = 0;
    uint64_t *  chain_dbg = nullptr;  // qmm_chain_debug: device buffer the next persistent launches stamp their phases into
    bool        chain_attr_set = false;      // the kernel's dynamic-LDS limit has been raised on this device
    int         chain_checked = 0;   // chain_launches when the error word was last read
    hipStream_t chain_stream = nullptr;         // stream of the recorded steps
    char        name[128] = {0};
    // qmm_trace_begin / qmm_trace_end: while set, every kernel launch of the MUL_MAT path appends its label here (host side only;
    // bench.py buckets its roofline by the kernels a call actually issued, not by a replica of the dispatch rules)
    std::string * trace = nullptr;

    // `st` is used verbatim: NULL is HIP's default stream (what torch's default stream is), not ours
    hipStream_t s(void * st) const { return (hipStream_t) st; }
};

// launches what a recording context has collected (qmm_chain_begin); every entry point that queues other work calls it first, so
// stream order is the order of the calls whether or not a chain is being recorded.  Defined in qmm_api.hip.
int qmm_internal_chain_flush(qmm_ctx * c);
#define QMM_CHAIN_FLUSH(c)                                                      \
    do {                                                                        \
        if ((c)->chain_on) { int rc_ = qmm_internal_chain_flush(c); if (rc_) return rc_; } \
    } while (0)

#define QMM_TRACE(c, ...)                                                                   \
    do {                                                                                    \
        if ((c)->trace) { char b_[96]; snprintf(b_, sizeof(b_), __VA_ARGS__); (c)->trace->append(b_).push_back(';'); } \
    } while (0)

namespace qmm {

inline std::string & last_error() {
    static thread_local std::string e;
    return e;
}

inline int fail(int code, const char * fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error() = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                       \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) return ::qmm::fail(QMM_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// grows the context workspace; synchronizes the device when it has to reallocate
inline int ensure_ws(qmm_ctx * c, size_t bytes) {
    if (bytes <= c->ws_bytes) return QMM_OK;
    HIP_TRY(hipDeviceSynchronize());
    if (c->ws) HIP_TRY(hipFree(c->ws));
    c->ws = nullptr;
    c->ws_bytes = 0;
    const size_t gran = (size_t) 32 << 20;
    const size_t want = (bytes + gran - 1) / gran * gran;
    HIP_TRY(hipMalloc(&c->ws, want));
    c->ws_bytes = want;
    return QMM_OK;
}

} // namespace qmm
