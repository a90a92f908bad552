// qmm_api.hip — host side of the C-ABI declared in include/ggml_mi355x_qmm.h.
// gfx950 only; there is no CPU path: every failure is reported, nothing is silently emulated.

#include "qmm_host.h"

#include <algorithm>

#include "qmm_matvec.hiph"
#include "qmm_mfma.hiph"
#include "qmm_mfma_regb.hiph"
#include "qmm_mfma_r64s.hiph"
#include "qmm_moe.hiph"
#include "qmm_chain.hiph"

using namespace qmm;

namespace {

bool type_ok(int t) { return type_known(t); }
int  blck(int t) { return type_blck(t); }
int  tsize(int t) { return type_tsize(t); }

} // namespace

template <int T>
static int launch_dequant(hipStream_t st, const void * w, int64_t rb, int64_t rows, int64_t K, float * dst) {
    const int64_t n = rows * (K / Traits<T>::UNIT_W);
    if (n == 0) return QMM_OK;
    hipLaunchKernelGGL((dequant_kernel<T>), dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, st,
                       (const uint8_t *) w, rb, rows, (int) K, dst);
    HIP_TRY(hipGetLastError());
    return QMM_OK;
}

static bool group_has_extras(const MatvecGroup & g) {
    bool ex = g.norm_w != nullptr || g.swiglu != 0;
    for (int i = 0; i < g.n; ++i) ex = ex || g.res[i] != nullptr;
    return ex;
}

template <int T, int NTOK>
static int launch_matvec_n(qmm_ctx * c, hipStream_t st, const MatvecGroup & g, const float * x, int64_t ldx, int K) {
    const size_t lds = matvec_lds_bytes<T, NTOK>(K) + (g.norm_w ? (size_t) NTOK * K * 4 : 0);
    if (lds > 160 * 1024) return fail(QMM_EUNSUPPORTED, "matvec: %d tokens x K=%d needs %zu B of LDS", NTOK, K, lds);
    auto kern = group_has_extras(g) ? matvec_kernel<T, NTOK, true> : matvec_kernel<T, NTOK, false>;
    if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    const int total = g.swiglu ? g.row_end[0] : g.row_end[g.n - 1];
    // one block per CU (the activation vector is quantized once per CU); 16 waves per block unless there are
    // fewer rows than that per CU.  Every wave gets a contiguous chunk of rows (+-1 row balance).
    int nw = (total + c->cus - 1) / c->cus;
    nw = nw >= 16 ? 16 : nw > 8 ? 16 : nw > 4 ? 8 : 4;
    int blocks = (total + nw - 1) / nw;
    if (blocks > c->cus * c->mv_bpc) blocks = c->cus * c->mv_bpc;
    QMM_TRACE(c, "matvec_kernel<%d,%d,%s>", T, NTOK, group_has_extras(g) ? "true" : "false");
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(nw * WAVE), lds, st, g, x, ldx, K, c->act_mode | (c->mv_onepass ? 0 : 256), g.row_end[g.n - 1]);
    HIP_TRY(hipGetLastError());
    return QMM_OK;
}

// (q80: the group holds Q8_0 matrices too; no fused norm, N <= 4: mul_mat_group_impl)
template <int NTOK>
static int launch_kmix_n(qmm_ctx * c, hipStream_t st, const MatvecGroup & g, const float * x, int64_t ldx, int K, bool q80 = false) {
    const size_t lds = kmix_lds_bytes(NTOK, K) + (g.norm_w ? (size_t) NTOK * K * 4 : 0) + (q80 ? matvec_lds_bytes<T_Q8_0P, NTOK>(K) : 0);
    if (lds > 160 * 1024) return fail(QMM_EUNSUPPORTED, "mixed-type matvec: %d tokens x K=%d needs %zu B of LDS", NTOK, K, lds);
    auto kern = group_has_extras(g) ? matvec_kmix_kernel<NTOK, true> : matvec_kmix_kernel<NTOK, false>;
    if constexpr (NTOK <= 4) {
        if (q80) kern = group_has_extras(g) ? matvec_kmix_kernel<NTOK, true, true> : matvec_kmix_kernel<NTOK, false, true>;
    } else if (q80) {
        return fail(QMM_EUNSUPPORTED, "mixed-format matvec: up to 4 tokens");
    }
    if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
    const int total = g.row_end[g.n - 1];
    int nw = (total + c->cus - 1) / c->cus;
    nw = nw > 8 ? 16 : nw > 4 ? 8 : 4;
    int blocks = (total + nw - 1) / nw;
    if (blocks > c->cus * c->mv_bpc) blocks = c->cus * c->mv_bpc;
    QMM_TRACE(c, q80 ? "matvec_kmix_kernel<%d,%s,q8_0>" : "matvec_kmix_kernel<%d,%s>", NTOK, group_has_extras(g) ? "true" : "false");
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(nw * WAVE), lds, st, g, x, ldx, K, c->act_mode | (c->mv_onepass ? 0 : 256), total);
    HIP_TRY(hipGetLastError());
    return QMM_OK;
}
static int launch_kmix(qmm_ctx * c, hipStream_t st, const MatvecGroup & g, const float * x, int64_t ldx, int K, int N, bool q80 = false) {
    switch (N) {
        case 1: return launch_kmix_n<1>(c, st, g, x, ldx, K, q80);
        case 2: return launch_kmix_n<2>(c, st, g, x, ldx, K, q80);
        case 3: return launch_kmix_n<3>(c, st, g, x, ldx, K, q80);
        case 4: return launch_kmix_n<4>(c, st, g, x, ldx, K, q80);
        case 5: return launch_kmix_n<5>(c, st, g, x, ldx, K, q80);
        case 6: return launch_kmix_n<6>(c, st, g, x, ldx, K, q80);
        case 7: return launch_kmix_n<7>(c, st, g, x, ldx, K, q80);
        case 8: return launch_kmix_n<8>(c, st, g, x, ldx, K, q80);
        default: return fail(QMM_EINVAL, "kmix matvec: N=%d", N);
    }
}

template <int T>
static int launch_matvec(qmm_ctx * c, hipStream_t st, const MatvecGroup & g, const float * x, int64_t ldx, int K, int N) {
    switch (N) {
        case 1: return launch_matvec_n<T, 1>(c, st, g, x, ldx, K);
        case 2: return launch_matvec_n<T, 2>(c, st, g, x, ldx, K);
        case 3: return launch_matvec_n<T, 3>(c, st, g, x, ldx, K);
        case 4: return launch_matvec_n<T, 4>(c, st, g, x, ldx, K);
        case 5: return launch_matvec_n<T, 5>(c, st, g, x, ldx, K);
        case 6: return launch_matvec_n<T, 6>(c, st, g, x, ldx, K);
        case 7: return launch_matvec_n<T, 7>(c, st, g, x, ldx, K);
        case 8: return launch_matvec_n<T, 8>(c, st, g, x, ldx, K);
        default: return fail(QMM_EINVAL, "matvec: N=%d", N);
    }
}

static int matvec_any(qmm_ctx * c, hipStream_t st, int type, const MatvecGroup & g, const float * x, int64_t ldx, int K, int N) {
#define QMM_X(TT) return launch_matvec<TT>(c, st, g, x, ldx, K, N)
    QMM_FOR_TYPE(type, QMM_X)
#undef QMM_X
}

static int check_mm(int type, const void * w, int64_t rb, int64_t K, const float * x, int64_t ldx, const char * who) {
    if (!type_ok(type)) return fail(QMM_EUNSUPPORTED, "%s: type %d not supported", who, type);
    if (K <= 0 || K % blck(type)) return fail(QMM_EUNSUPPORTED, "%s: K=%lld must be a multiple of %d", who, (long long) K, blck(type));
    if (rb < (int64_t) qmm_row_size(type, K)) return fail(QMM_EINVAL, "%s: weight row stride %lld < row size", who, (long long) rb);
    if ((uintptr_t) x % 16 || ldx % 4 || ldx < K) return fail(QMM_EINVAL, "%s: src1 must be 16-byte aligned with ldx %% 4 == 0", who);
    if (type_planar(type) && (!qmm_planar_type(type_base(type), K, rb) || (uintptr_t) w % 16))
        return fail(QMM_EINVAL, "%s: planar type %d wants 16-byte aligned rows and K a multiple of %d", who, type, type_base(type) == T_Q6_K ? 2048 : 256);
    return QMM_OK;
}


// ------------------------------------------------------------------------------------------- chains (qmm_chain.hiph)

static size_t chain_act_bytes(int fam, int ntok, int K) {
    return fam == CHAIN_FAM_Q8_K ? kmix_lds_bytes(ntok, K) : (((size_t) ntok * K + (size_t) ntok * (K / 32) * 4 + 15) & ~(size_t) 15);
}
static size_t chain_step_lds(const ChainStep & st, int ntok) {
    return chain_act_bytes(st.fam, ntok, st.K) + (st.g.norm_w ? (size_t) ntok * st.K * 4 : 0);
}
constexpr size_t CHAIN_SLAB = (size_t) CHAIN_RPW * CHAIN_NW * 4;          // one token

// the step as a launch of its own (a chain of one, or chains switched off): the kernels of round 1
static int chain_step_plain(qmm_ctx * c, hipStream_t st, const ChainStep & s) {
    bool uniform = true;
    for (int i = 1; i < s.g.n; ++i) uniform = uniform && s.g.type[i] == s.g.type[0];
    if (uniform) return matvec_any(c, st, s.g.type[0], s.g, s.x, s.ldx, s.K, 1);
    return launch_kmix(c, st, s.g, s.x, s.ldx, s.K, 1);
}

static int chain_launch(qmm_ctx * c) {
    if (!c->chain || c->chain->empty()) return QMM_OK;
    std::vector<ChainStep> & v = *c->chain;
    hipStream_t st = c->chain_stream;
    int rc = QMM_OK;
    for (size_t i = 0; i < v.size() && rc == QMM_OK;) {
        size_t n = v.size() - i;
        if (n > (size_t) CHAIN_MAX_STEPS) n = CHAIN_MAX_STEPS;
        if (n == 1) {
            rc = chain_step_plain(c, st, v[i]);
        } else {
            ChainArgs a;
            memset(&a, 0, sizeof(a));
            size_t lds = 0;
            for (size_t k = 0; k < n; ++k) {
                a.s[k] = v[i + k];
                const size_t b = chain_step_lds(a.s[k], 1);
                if (b > lds) lds = b;
            }
            a.sync = c->chain_sync;
            a.n = (int) n;
            a.act_mode = c->act_mode;
            a.timeout = 2000000u;                                           // 20 ms of the 100 MHz clock
            a.res_off = (uint32_t) ((lds + 15) & ~(size_t) 15);
            if (c->chain_dbg) {                                             // diagnostic stamps, consecutive launches behind each other
                a.dbg = c->chain_dbg;
                c->chain_dbg += n * (size_t) c->cus * 8;
            }
            const size_t total = a.res_off + CHAIN_SLAB;
            auto kern = matvec_chain_kernel<1>;
            if (!c->chain_attr_set) {
                hipError_t e = hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
                if (e != hipSuccess) { rc = fail(QMM_EHIP, "chain: hipFuncSetAttribute: %s", hipGetErrorString(e)); break; }
                c->chain_attr_set = true;
            }
            QMM_TRACE(c, "matvec_chain_kernel<1>");
            hipLaunchKernelGGL(kern, dim3(c->cus), dim3(CHAIN_NW * WAVE), total, st, a);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) { rc = fail(QMM_EHIP, "chain launch: %s", hipGetErrorString(e)); break; }
            c->chain_launches++;
            c->chain_steps += (int) n;
        }
        i += n;
    }
    v.clear();
    return rc;
}

// Record one group call (one token) as one step per activation format.  Returns 1 when recorded, 0 when the call is not eligible
// (the caller flushes and launches it the ordinary way), < 0 on error.
static int chain_record(qmm_ctx * c, hipStream_t st, const qmm_weight * ws, int nw, int64_t K, const float * x, int64_t ldx, const qmm_mv_extra * ex) {
    if (nw > MV_MAX_GROUP) return 0;
    const int W = c->cus * CHAIN_NW;
    int fams[2] = { 0, 0 };
    int64_t rows = 0;
    for (int i = 0; i < nw; ++i) {
        if (ws[i].M <= 0) return 0;
        if (ws[i].type != T_Q4_0 && ws[i].type != T_Q8_0 && ws[i].type != T_Q4_K && ws[i].type != T_Q5_K && ws[i].type != T_Q6_K) return 0;
        const bool kq = ws[i].type == T_Q4_K || ws[i].type == T_Q5_K || ws[i].type == T_Q6_K;
        fams[kq ? 1 : 0]++;
        rows += ws[i].M;
        // a result on top of the activations is a race between workgroups in any launch form: leave it to the caller's order
        const char * d0 = (const char *) ws[i].dst, * x0 = (const char *) x;
        if (d0 < x0 + K * 4 && x0 < d0 + ws[i].M * 4) return 0;
    }
    if (fams[1] && K % 256) return 0;
    if ((rows + W - 1) / W > CHAIN_RPW) return 0;
    if (ex && ex->swiglu && fams[0] && fams[1]) return 0;
    if (chain_act_bytes(fams[1] ? CHAIN_FAM_Q8_K : CHAIN_FAM_Q8_0, 1, (int) K) + (ex && ex->norm_w ? (size_t) K * 4 : 0) + CHAIN_SLAB + 1024 > 150 * 1024) return 0;
    if (c->chain->empty()) c->chain_stream = st;
    else if (c->chain_stream != st) { int rc = chain_launch(c); if (rc) return rc; c->chain_stream = st; }
    bool first = true;
    for (int fam = 1; fam >= 0; --fam) {
        if (!fams[fam]) continue;
        ChainStep s;
        memset(&s, 0, sizeof(s));
        int r = 0;
        for (int i = 0; i < nw; ++i) {
            const bool kq = ws[i].type == T_Q4_K || ws[i].type == T_Q5_K || ws[i].type == T_Q6_K;
            if ((kq ? 1 : 0) != fam) continue;
            const int k = s.g.n++;
            s.g.w[k] = (const uint8_t *) ws[i].w;  s.g.dst[k] = ws[i].dst;  s.g.row_bytes[k] = ws[i].w_row_bytes;  s.g.ldd[k] = ws[i].ldd;
            s.g.type[k] = ws[i].type;
            r += (int) ws[i].M;
            s.g.row_end[k] = r;
            s.g.res[k] = ex ? ex->residual[i] : nullptr;
        }
        if (ex) { s.g.norm_w = ex->norm_w; s.g.norm_eps = ex->norm_eps; s.g.swiglu = ex->swiglu; }
        s.x = x;  s.ldx = ldx;  s.K = (int) K;  s.fam = fam;
        s.dep = first ? 1 : 0;                          // the second format of one call reads the same x and writes other rows
        s.restage = 1;
        first = false;
        c->chain->push_back(s);
    }
    return 1;
}

int qmm_internal_chain_flush(qmm_ctx * c) { return c->chain_on ? chain_launch(c) : QMM_OK; }

extern "C" {

int qmm_abi_version(void) { return 2; }

const char * qmm_last_error(void) { return last_error().c_str(); }

int qmm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

qmm_ctx * qmm_create(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) {
        fail(QMM_ENODEV, "qmm_create: no HIP device %d (count %d)", device, n);
        return nullptr;
    }
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) != hipSuccess) {
        fail(QMM_ENODEV, "qmm_create: hipGetDeviceProperties failed");
        return nullptr;
    }
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
        fail(QMM_ENODEV, "qmm_create: device %d is %s; this library is built for gfx950 only", device, p.gcnArchName);
        return nullptr;
    }
    qmm_ctx * c = new qmm_ctx;
    c->device = device;
    c->cus = p.multiProcessorCount;
    snprintf(c->name, sizeof(c->name), "%s", p.name);
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void **) &c->flag, 16) != hipSuccess || hipMemset(c->flag, 0, 16) != hipSuccess) {
        fail(QMM_EHIP, "qmm_create: stream/flag setup failed");
        delete c;
        return nullptr;
    }
    c->kcnt_n = 1 << 16;
    if (hipMalloc((void **) &c->kcnt, (size_t) c->kcnt_n * sizeof(int)) != hipSuccess || hipMemset(c->kcnt, 0, (size_t) c->kcnt_n * sizeof(int)) != hipSuccess) {
        fail(QMM_EHIP, "qmm_create: split-K counters");
        delete c;
        return nullptr;
    }
    c->chain = new std::vector<ChainStep>();
    if (hipMalloc((void **) &c->chain_sync, sizeof(ChainSync)) != hipSuccess || hipMemset(c->chain_sync, 0, sizeof(ChainSync)) != hipSuccess) {
        fail(QMM_EHIP, "qmm_create: chain state setup failed");
        delete c->chain;
        delete c;
        return nullptr;
    }
    for (int j = 0; j < n; ++j) {                                   // row split copies between devices: peer access, best effort
        int can = 0;
        if (j != device && hipDeviceCanAccessPeer(&can, device, j) == hipSuccess && can) {
            const hipError_t pe = hipDeviceEnablePeerAccess(j, 0);
            if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) (void) hipGetLastError();
        }
    }
    (void) hipGetLastError();
    const char * e = getenv("GGML_MI355X_SPLITK");
    if (e) c->splitk = atoi(e);
    e = getenv("GGML_MI355X_ACT_MODE");
    if (e) c->act_mode = atoi(e) ? QMM_ACT_X86 : QMM_ACT_REF;
    e = getenv("GGML_MI355X_MV_KMIX");      // 0: off, 1: K-quant groups only, 2 (default): also groups with Q8_0 matrices
    if (e) c->mv_kmix = atoi(e);
    e = getenv("GGML_MI355X_WIDE");
    if (e) c->wide = atoi(e);
    e = getenv("GGML_MI355X_R64");
    if (e) c->r64 = atoi(e);
    e = getenv("GGML_MI355X_R64S");
    if (e) c->r64s = atoi(e);
    e = getenv("GGML_MI355X_PREP_REG");
    if (e) c->prep_reg = atoi(e);
    e = getenv("GGML_MI355X_REGB_Q23");
    if (e) c->regb_q23 = atoi(e) != 0;
    e = getenv("GGML_MI355X_MV_ONEPASS");
    if (e) c->mv_onepass = atoi(e);
    e = getenv("GGML_MI355X_SIDE");
    if (e) c->side_on = atoi(e);
    e = getenv("GGML_MI355X_SPLITK_COMBINE");
    if (e) c->splitk_combine = atoi(e);
    e = getenv("GGML_MI355X_CHAIN");
    if (e) c->chain_enabled = atoi(e);
    e = getenv("GGML_MI355X_MV_BPC");
    if (e && atoi(e) >= 1 && atoi(e) <= 8) c->mv_bpc = atoi(e);
    e = getenv("GGML_MI355X_MM_GROUP");
    if (e) c->mm_group = atoi(e);
    e = getenv("GGML_MI355X_SKINNY");
    if (e) c->skinny = atoi(e);
    e = getenv("GGML_MI355X_SKINNY_MAXN");
    if (e && atoi(e) >= 9) c->skinny_max_n = c->skinny_max_n_few = atoi(e);
    e = getenv("GGML_MI355X_ABLATE");
    if (e) { int v = atoi(e); (void) hipMemcpyToSymbol(HIP_SYMBOL(g_mfma_dbg), &v, sizeof(int)); }
    e = getenv("GGML_MI355X_PREC");
    if (e) c->prec = (!strcmp(e, "bf16") || !strcmp(e, "0")) ? QMM_PREC_BF16 : QMM_PREC_F16_Q8;
    return c;
}

void qmm_destroy(qmm_ctx * c) {
    if (!c) return;
    (void) hipSetDevice(c->device);
    (void) hipDeviceSynchronize();
    if (c->ws) (void) hipFree(c->ws);
    if (c->flag) (void) hipFree(c->flag);
    if (c->kcnt) (void) hipFree(c->kcnt);
    if (c->chain_sync) (void) hipFree(c->chain_sync);
    delete c->chain;
    delete c->trace;
    for (int l = 0; l < 3; ++l) {
        if (c->side[l]) (void) hipStreamDestroy(c->side[l]);
        if (c->ev_join[l]) (void) hipEventDestroy(c->ev_join[l]);
    }
    if (c->ev_fork) (void) hipEventDestroy(c->ev_fork);
    if (c->stream) (void) hipStreamDestroy(c->stream);
    delete c;
}

int qmm_device(const qmm_ctx * c) { return c ? c->device : -1; }

void * qmm_stream(const qmm_ctx * c) { return c ? (void *) c->stream : nullptr; }

int qmm_device_info(const qmm_ctx * c, char * name, size_t name_len, size_t * mem_free, size_t * mem_total, int * cus) {
    if (!c) return fail(QMM_EINVAL, "null ctx");
    HIP_TRY(hipSetDevice(c->device));
    if (name && name_len) snprintf(name, name_len, "%s", c->name);
    size_t f = 0, t = 0;
    HIP_TRY(hipMemGetInfo(&f, &t));
    if (mem_free) *mem_free = f;
    if (mem_total) *mem_total = t;
    if (cus) *cus = c->cus;
    return QMM_OK;
}

int qmm_set_act_mode(qmm_ctx * c, int m) {
    if (!c || (m != QMM_ACT_REF && m != QMM_ACT_X86)) return fail(QMM_EINVAL, "bad act mode");
    c->act_mode = m;
    return QMM_OK;
}
int qmm_set_precision(qmm_ctx * c, int p) {
    if (!c || (p != QMM_PREC_BF16 && p != QMM_PREC_F16_Q8)) return fail(QMM_EINVAL, "bad precision");
    c->prec = p;
    return QMM_OK;
}

void * qmm_malloc(qmm_ctx * c, size_t bytes) {
    if (!c) return nullptr;
    void * p = nullptr;
    if (hipSetDevice(c->device) != hipSuccess || hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) {
        fail(QMM_ENOMEM, "hipMalloc(%zu) failed", bytes);
        return nullptr;
    }
    return p;
}
void qmm_free(qmm_ctx * c, void * p) {
    if (!c || !p) return;
    (void) hipSetDevice(c->device);
    (void) hipFree(p);
}
// page-locked host memory: transfers from / to it are real DMA and the *_async copies do not stage (the reference's
// counterpart is the rpcmem / ION pool shared with the cDSP, ggml-hexagon.cpp:4698-4747)
void * qmm_host_malloc(qmm_ctx * c, size_t bytes) {
    if (!c) return nullptr;
    void * p = nullptr;
    if (hipSetDevice(c->device) != hipSuccess || hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        (void) hipGetLastError();
        fail(QMM_ENOMEM, "hipHostMalloc(%zu) failed", bytes);
        return nullptr;
    }
    return p;
}
void qmm_host_free(qmm_ctx * c, void * p) {
    if (!c || !p) return;
    (void) hipSetDevice(c->device);
    (void) hipHostFree(p);
}
int qmm_memcpy_h2d(qmm_ctx * c, void * dst, const void * src, size_t n, void * st) {
    HIP_TRY(hipSetDevice(c->device));
    QMM_CHAIN_FLUSH(c);
    HIP_TRY(hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, c->s(st)));
    HIP_TRY(hipStreamSynchronize(c->s(st)));
    return QMM_OK;
}
int qmm_memcpy_d2h(qmm_ctx * c, void * dst, const void * src, size_t n, void * st) {
    HIP_TRY(hipSetDevice(c->device));
    QMM_CHAIN_FLUSH(c);
    HIP_TRY(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, c->s(st)));
    HIP_TRY(hipStreamSynchronize(c->s(st)));
    return QMM_OK;
}
int qmm_memcpy_d2d(qmm_ctx * c, void * dst, const void * src, size_t n, void * st) {
    HIP_TRY(hipSetDevice(c->device));
    QMM_CHAIN_FLUSH(c);
    HIP_TRY(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, c->s(st)));
    return QMM_OK;
}
struct qmm_event { hipEvent_t ev; };
int qmm_memcpy_h2d_async(qmm_ctx * c, void * dst, const void * src, size_t n, void * st) {
    if (!c) return fail(QMM_EINVAL, "null ctx");
    HIP_TRY(hipSetDevice(c->device));
    QMM_CHAIN_FLUSH(c);
    HIP_TRY(hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, c->s(st)));
    return QMM_OK;
}
int qmm_memcpy_d2h_async(qmm_ctx * c, void * dst, const void * src, size_t n, void * st) {
    if (!c) return fail(QMM_EINVAL, "null ctx");
    HIP_TRY(hipSetDevice(c->device));
    QMM_CHAIN_FLUSH(c);
    HIP_TRY(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, c->s(st)));
    return QMM_OK;
}
int qmm_event_synchronize(qmm_ctx * c, qmm_event * e) {
    if (!c || !e) return fail(QMM_EINVAL, "qmm_event_synchronize: null argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventSynchronize(e->ev));
    return QMM_OK;
}
int qmm_memcpy2d_d2d(qmm_ctx * c, void * dst, size_t dpitch, const void * src, size_t spitch, size_t width, size_t height, void * st) {
    if (!c) return fail(QMM_EINVAL, "null ctx");
    if (width == 0 || height == 0) return QMM_OK;
    HIP_TRY(hipSetDevice(c->device));
    QMM_CHAIN_FLUSH(c);
    if (dpitch == width && spitch == width) HIP_TRY(hipMemcpyAsync(dst, src, width * height, hipMemcpyDeviceToDevice, c->s(st)));
    else HIP_TRY(hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyDeviceToDevice, c->s(st)));
    return QMM_OK;
}
qmm_event * qmm_event_create(qmm_ctx * c) {
    if (!c || hipSetDevice(c->device) != hipSuccess) { fail(QMM_EINVAL, "qmm_event_create: bad ctx"); return nullptr; }
    qmm_event * e = new qmm_event;
    if (hipEventCreateWithFlags(&e->ev, hipEventDisableTiming) != hipSuccess) {
        fail(QMM_EHIP, "qmm_event_create: hipEventCreate failed");
        delete e;
        return nullptr;
    }
    return e;
}
qmm_event * qmm_event_create_timing(qmm_ctx * c) {
    if (!c || hipSetDevice(c->device) != hipSuccess) { fail(QMM_EINVAL, "qmm_event_create_timing: bad ctx"); return nullptr; }
    qmm_event * e = new qmm_event;
    if (hipEventCreate(&e->ev) != hipSuccess) {
        fail(QMM_EHIP, "qmm_event_create_timing: hipEventCreate failed");
        delete e;
        return nullptr;
    }
    return e;
}
int qmm_event_elapsed_ms(qmm_ctx * c, qmm_event * e0, qmm_event * e1, float * ms) {
    if (!c || !e0 || !e1 || !ms) return fail(QMM_EINVAL, "qmm_event_elapsed_ms: null argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventElapsedTime(ms, e0->ev, e1->ev));
    return QMM_OK;
}
void qmm_event_destroy(qmm_ctx * c, qmm_event * e) {
    if (!e) return;
    if (c) (void) hipSetDevice(c->device);
    (void) hipEventDestroy(e->ev);
    delete e;
}
int qmm_event_record(qmm_ctx * c, qmm_event * e, void * st) {
    if (!c || !e) return fail(QMM_EINVAL, "qmm_event_record: null argument");
    HIP_TRY(hipSetDevice(c->device));
    QMM_CHAIN_FLUSH(c);
    HIP_TRY(hipEventRecord(e->ev, c->s(st)));
    return QMM_OK;
}
int qmm_stream_wait_event(qmm_ctx * c, void * st, qmm_event * e) {
    if (!c || !e) return fail(QMM_EINVAL, "qmm_stream_wait_event: null argument");
    HIP_TRY(hipSetDevice(c->device));
    QMM_CHAIN_FLUSH(c);
    HIP_TRY(hipStreamWaitEvent(c->s(st), e->ev, 0));
    return QMM_OK;
}
int qmm_memset(qmm_ctx * c, void * dst, int v, size_t n, void * st) {
    HIP_TRY(hipSetDevice(c->device));
    QMM_CHAIN_FLUSH(c);
    HIP_TRY(hipMemsetAsync(dst, v, n, c->s(st)));
    return QMM_OK;
}
int qmm_synchronize(qmm_ctx * c, void * st) {
    HIP_TRY(hipSetDevice(c->device));
    if (c->chain_on && !c->chain->empty()) { int rc = chain_launch(c); if (rc) return rc; }       // nothing recorded stays behind a wait
    HIP_TRY(hipStreamSynchronize(c->s(st)));
    if (c->chain_launches != c->chain_checked) {
        uint32_t err = 0;
        HIP_TRY(hipMemcpy(&err, &c->chain_sync->err[0], sizeof(err), hipMemcpyDeviceToHost));
        c->chain_checked = c->chain_launches;
        if (err) {
            HIP_TRY(hipDeviceSynchronize());
            HIP_TRY(hipMemset(c->chain_sync, 0, sizeof(ChainSync)));
            return fail(QMM_EHIP, "chain: a grid-wide wait timed out (workgroups not co-resident?); results of that launch are undefined");
        }
    }
    if (c->mfma_calls != c->mfma_checked) {                  // prefill launches since the last look: did the f16 mode overflow?
        c->mfma_checked = c->mfma_calls;
        int nf = 0;
        HIP_TRY(hipMemcpyFromSymbol(&nf, HIP_SYMBOL(g_mfma_nonfinite), sizeof(int)));
        if (nf) {
            const int zero = 0;
            HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_mfma_nonfinite), &zero, sizeof(int)));
            return fail(QMM_EUNSUPPORTED, "prefill (f16 on Q8 activations) produced non-finite values: a weight block exceeds the f16 range (|w| >= 65504) "
                                          "or the activations are not finite; use QMM_PREC_BF16 (GGML_MI355X_PREC=bf16) for this model");
        }
    }
    if (c->id_calls != c->id_checked) {                      // MUL_MAT_ID launches since the last look: did a kernel meet an expert id out of range?
        c->id_checked = c->id_calls;
        int flag = 0;
        HIP_TRY(hipMemcpy(&flag, c->flag, sizeof(int), hipMemcpyDeviceToHost));
        if (flag) {
            HIP_TRY(hipMemset(c->flag, 0, sizeof(int)));
            return fail(QMM_EINVAL, "MUL_MAT_ID: expert id out of range seen by the kernel");
        }
    }
    return QMM_OK;
}

size_t qmm_row_size(int type, int64_t k) {
    if (!type_ok(type) || k % blck(type)) return 0;
    return (size_t) (k / blck(type)) * tsize(type);
}

// ------------------------------------------------------------------------------------------- dequantize

int qmm_dequantize(qmm_ctx * c, int type, const void * w, int64_t rb, int64_t rows, int64_t K, float * dst, void * st) {
    if (!c || !type_ok(type)) return fail(QMM_EINVAL, "qmm_dequantize: bad ctx/type %d", type);
    if (K <= 0 || K % blck(type)) return fail(QMM_EINVAL, "qmm_dequantize: K=%lld", (long long) K);
    if (rb < (int64_t) qmm_row_size(type, K)) return fail(QMM_EINVAL, "qmm_dequantize: row stride too small");
    if ((uintptr_t) dst % 16) return fail(QMM_EINVAL, "qmm_dequantize: dst must be 16-byte aligned");
    HIP_TRY(hipSetDevice(c->device));
    QMM_CHAIN_FLUSH(c);
    hipStream_t s = c->s(st);
#define QMM_X(TT) return launch_dequant<TT>(s, w, rb, rows, K, dst)
    QMM_FOR_TYPE(type, QMM_X)
#undef QMM_X
}

// ------------------------------------------------------------------------------------------- planar rows

int qmm_planar_type(int type, int64_t K, int64_t w_row_bytes) {
    if (type != T_Q4_0 && type != T_Q8_0 && type != T_Q6_K) return 0;
    if (K <= 0 || K % (type == T_Q6_K ? 2048 : 256) || w_row_bytes % 16) return 0;     // planes and rows start on 16-byte boundaries
    return type + 100;
}

int qmm_repack_rows(qmm_ctx * c, int type, void * w, int64_t w_row_bytes, int64_t rows, int64_t K, int to_planar, void * st) {
    if (!c) return fail(QMM_EINVAL, "qmm_repack_rows: NULL context");
    if (!qmm_planar_type(type, K, w_row_bytes) || (uintptr_t) w % 16 || w_row_bytes < (int64_t) qmm_row_size(type, K))
        return fail(QMM_EUNSUPPORTED, "qmm_repack_rows: type %d, K=%lld, row stride %lld has no planar form", type, (long long) K, (long long) w_row_bytes);
    if (rows <= 0) return QMM_OK;
    HIP_TRY(hipSetDevice(c->device));
    QMM_CHAIN_FLUSH(c);
    const size_t lds = qmm_row_size(type, K);
    if (lds > 150 * 1024) return fail(QMM_EUNSUPPORTED, "qmm_repack_rows: a row of %zu bytes does not fit LDS", lds);
#define QMM_RP(TT, DIR)                                                                                                             \
    do {                                                                                                                            \
        auto kern = repack_rows_kernel<TT, DIR>;                                                                                    \
        if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); \
        hipLaunchKernelGGL(kern, dim3((unsigned) rows), dim3(256), lds, c->s(st), (uint8_t *) w, w_row_bytes, (int) K);              \
    } while (0)
    if (type == T_Q4_0)      { if (to_planar) QMM_RP(T_Q4_0, true); else QMM_RP(T_Q4_0, false); }
    else if (type == T_Q8_0) { if (to_planar) QMM_RP(T_Q8_0, true); else QMM_RP(T_Q8_0, false); }
    else                     { if (to_planar) QMM_RP(T_Q6_K, true); else QMM_RP(T_Q6_K, false); }
#undef QMM_RP
    HIP_TRY(hipGetLastError());
    return QMM_OK;
}

// ------------------------------------------------------------------------------------------- quantize_act

int qmm_quantize_act(qmm_ctx * c, int vt, const float * x, int64_t rows, int64_t K, int64_t ldx,
                     int8_t * q, float * d, int16_t * bs, void * st) {
    if (!c || (vt != T_Q8_0 && vt != T_Q8_1 && vt != T_Q8_K)) return fail(QMM_EINVAL, "qmm_quantize_act: vec_dot_type %d", vt);
    if (vt == T_Q8_1 && (!bs || (uintptr_t) bs % 4)) return fail(QMM_EINVAL, "qmm_quantize_act: Q8_1 wants the s array (f32 [rows, K/32]) in `bs`");
    if (K <= 0 || K % (vt == T_Q8_K ? 256 : 32) || ldx % 4 || (uintptr_t) x % 16 || (uintptr_t) q % 4)
        return fail(QMM_EINVAL, "qmm_quantize_act: K/ldx/alignment");
    if (rows == 0) return QMM_OK;
    HIP_TRY(hipSetDevice(c->device));
    QMM_CHAIN_FLUSH(c);
    const int rpb = 4;
    dim3 grid((unsigned) ((rows + rpb - 1) / rpb));
    if (vt == T_Q8_0)
        hipLaunchKernelGGL((quantize_act_kernel<T_Q8_0>), grid, dim3(256), 0, c->s(st), x, ldx, (int) rows, (int) K, c->act_mode, q, d, bs, rpb);
    else if (vt == T_Q8_1)
        hipLaunchKernelGGL((quantize_act_kernel<T_Q8_1>), grid, dim3(256), 0, c->s(st), x, ldx, (int) rows, (int) K, c->act_mode, q, d, bs, rpb);
    else
        hipLaunchKernelGGL((quantize_act_kernel<T_Q8_K>), grid, dim3(256), 0, c->s(st), x, ldx, (int) rows, (int) K, c->act_mode, q, d, bs, rpb);
    HIP_TRY(hipGetLastError());
    return QMM_OK;
}

// ------------------------------------------------------------------------------------------- mat-vec

static int mul_mat_group_impl(qmm_ctx * c, const qmm_weight * ws, int nw, int64_t K, const float * x, int64_t N, int64_t ldx, void * stream,
                              const qmm_mv_extra * ex) {
    if (!c || !ws || nw <= 0) return fail(QMM_EINVAL, "qmm_mul_mat_group: bad arguments");
    if (N <= 0) return QMM_OK;
    const float * norm_w = ex ? ex->norm_w : nullptr;
    if (ex) {
        if (N > QMM_MATVEC_MAX_N || nw > MV_MAX_GROUP) return fail(QMM_EUNSUPPORTED, "qmm_mul_mat_group_ex: batches of <= %d tokens, <= %d matrices", QMM_MATVEC_MAX_N, MV_MAX_GROUP);
        if (norm_w && ((uintptr_t) norm_w % 16 || ex->norm_eps < 0.0f)) return fail(QMM_EINVAL, "qmm_mul_mat_group_ex: norm weight must be 16-byte aligned, eps >= 0");
        if (norm_w && (size_t) N * K * 4 + (size_t) N * K * 11 / 8 + 4096 > 150 * 1024) return fail(QMM_EUNSUPPORTED, "qmm_mul_mat_group_ex: %lld rows of %lld do not fit LDS with the norm", (long long) N, (long long) K);
        if (ex->swiglu) {
            if ((ex->swiglu != 1 && ex->swiglu != 2) || nw != 2 || ws[0].type != ws[1].type || ws[0].M != ws[1].M || ws[0].M <= 0 || ex->residual[0] || ex->residual[1] ||
                (size_t) N * K * 5 / 4 + (norm_w ? (size_t) N * K * 4 : 0) + 4096 > 150 * 1024)
                return fail(QMM_EUNSUPPORTED, "qmm_mul_mat_group_ex: swiglu wants two matrices of one type and shape, no residuals, all tokens in one launch");
        }
    }
    auto extras = [&](MatvecGroup & g, const int * src_index, int64_t n0) {      // residual pointers of the matrices in g, norm parameters
        if (!ex) return;
        for (int k = 0; k < g.n; ++k) g.res[k] = ex->residual[src_index[k]] ? ex->residual[src_index[k]] + n0 * g.ldd[k] : nullptr;
        g.norm_w = norm_w;
        g.norm_eps = ex->norm_eps;
        g.swiglu = ex->swiglu;
    };
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = c->s(stream);
    for (int i = 0; i < nw; ++i) {
        int rc = check_mm(ws[i].type, ws[i].w, ws[i].w_row_bytes, K, x, ldx, "qmm_mul_mat");
        if (rc) return rc;
        if (ws[i].M < 0 || ws[i].ldd < ws[i].M) return fail(QMM_EINVAL, "qmm_mul_mat: ldd < M");
    }
    if (c->chain_on) {
        // recording (qmm_chain_begin): one-token groups are collected; anything else goes out behind what was collected
        int rec = N == 1 ? chain_record(c, st, ws, nw, K, x, ldx, ex) : 0;
        if (rec < 0) return rec;
        if (rec == 1) return QMM_OK;
        int rc = chain_launch(c);
        if (rc) return rc;
    }
    if (N <= QMM_MATVEC_MAX_N && nw >= 2 && nw <= MV_MAX_GROUP && c->mv_kmix) {
        // K-quant matrices of different types share the Q8_K activations: one mixed-type launch for the whole group
        bool kq = true, mixed = false;
        int n_k = 0, n_80 = 0;                                    // K-quant matrices (Q8_K activations), Q8_0 matrices (Q8_0 activations)
        for (int i = 0; i < nw; ++i) {
            const bool k = ws[i].type == T_Q4_K || ws[i].type == T_Q5_K || ws[i].type == T_Q6_K || ws[i].type == T_Q6_KP;
            const bool q = ws[i].type == T_Q8_0 || ws[i].type == T_Q8_0P;
            n_k += k;  n_80 += q;
            kq = kq && (k || q) && ws[i].M > 0;
            mixed = mixed || ws[i].type != ws[0].type;
        }
        // both activation formats in one group (Mixtral's q in Q4_K with k / v in Q8_0): one launch that stages the row twice
        const bool q80 = n_80 > 0;
        if (q80) kq = kq && n_k > 0 && c->mv_kmix > 1 && N <= 4 && !norm_w && !(ex && ex->swiglu) && (size_t) N * K * 21 / 8 + 4096 <= 150 * 1024;
        if (kq && mixed && (size_t) N * K * 11 / 8 + 4096 <= 150 * 1024) {
            MatvecGroup g;
            memset(&g, 0, sizeof(g));
            int rows = 0, idx[MV_MAX_GROUP];
            for (int i = 0; i < nw; ++i) {
                g.w[i] = (const uint8_t *) ws[i].w;  g.dst[i] = ws[i].dst;  g.row_bytes[i] = ws[i].w_row_bytes;  g.ldd[i] = ws[i].ldd;
                g.type[i] = ws[i].type;
                rows += (int) ws[i].M;
                g.row_end[i] = rows;
                idx[i] = i;
            }
            g.n = nw;
            extras(g, idx, 0);
            return launch_kmix(c, st, g, x, ldx, (int) K, (int) N, q80);
        }
    }
    if (N <= QMM_MATVEC_MAX_N) {
        // one launch per run of same-type weights (they share the in-kernel activation quantization)
        int i = 0;
        while (i < nw) {
            MatvecGroup g;
            memset(&g, 0, sizeof(g));
            int rows = 0, j = i, idx[MV_MAX_GROUP];
            while (j < nw && ws[j].type == ws[i].type && g.n < MV_MAX_GROUP) {
                if (ws[j].M > 0) {
                    idx[g.n] = j;
                    g.w[g.n] = (const uint8_t *) ws[j].w;
                    g.dst[g.n] = ws[j].dst;
                    g.row_bytes[g.n] = ws[j].w_row_bytes;
                    g.ldd[g.n] = ws[j].ldd;
                    rows += (int) ws[j].M;
                    g.row_end[g.n] = rows;
                    g.n++;
                }
                ++j;
            }
            if (g.n > 0) {
                // tokens that do not fit LDS together are processed in sub-batches
                int n_at_once = (int) N;
                while (n_at_once > 1 && ((size_t) n_at_once * K * 5 / 4 + 4096) > 150 * 1024) n_at_once = (n_at_once + 1) / 2;
                for (int64_t n0 = 0; n0 < N; n0 += n_at_once) {
                    const int nn = (int) ((N - n0) < n_at_once ? (N - n0) : n_at_once);
                    MatvecGroup gg = g;
                    for (int k = 0; k < gg.n; ++k) gg.dst[k] = g.dst[k] + n0 * g.ldd[k];
                    extras(gg, idx, n0);
                    int rc = matvec_any(c, st, ws[i].type, gg, x + n0 * ldx, ldx, (int) K, nn);
                    if (rc) return rc;
                }
            }
            i = j;
        }
        return QMM_OK;
    }
    c->mfma_calls++;
    // The group shares src1: its 16-bit operand is prepared once per activation format (key).  Runs of same-type matrices are one tiled
    // launch each where the shapes allow (mfma_mul_mat_group).  A run with a NEW key has nothing in common with what was issued before
    // it except src1: it goes to a side stream (own workspace slice) between a fork and a join event, so that its prep / MFMA / reduce
    // launches overlap the earlier runs' instead of queueing behind them; a run that reuses a prep stays on that prep's stream.
    struct Run { int i, j, key, lane; };
    Run runs[MV_MAX_GROUP * 2];
    int nruns = 0;
    for (int i = 0; i < nw;) {
        if (ws[i].M == 0) { ++i; continue; }
        int j = i + 1;
        while (j < nw && j - i < 4 && ws[j].type == ws[i].type && ws[j].M > 0) ++j;
        const int key = mfma_prep_key(c, ws[i].type, N, ws[i].M, K);
        bool same_key = true;
        for (int k = i + 1; k < j; ++k) same_key = same_key && mfma_prep_key(c, ws[k].type, N, ws[k].M, K) == key;
        if (!same_key) j = i + 1;
        if (nruns == (int) (sizeof(runs) / sizeof(runs[0]))) return fail(QMM_EUNSUPPORTED, "qmm_mul_mat_group: too many runs in one group");
        runs[nruns++] = { i, j, key, 0 };
        i = j;
    }
    int nlanes = 1;                                              // lane 0 = the caller's stream, 1..3 = side streams
    const bool side = c->side_on && nruns > 1 && !c->prep_x2;
    for (int r = 1; r < nruns; ++r) {
        int found = -1;
        for (int q = 0; q < r; ++q) if (runs[q].key == runs[r].key) found = runs[q].lane;      // reuses that run's prep: same stream, behind it
        runs[r].lane = found >= 0 ? found : (side && nlanes < 4 ? nlanes++ : 0);
    }
    if (nlanes > 1) {
        for (int l = 1; l < nlanes; ++l) {
            if (!c->side[l - 1]) HIP_TRY(hipStreamCreateWithFlags(&c->side[l - 1], hipStreamNonBlocking));
            if (!c->ev_join[l - 1]) HIP_TRY(hipEventCreateWithFlags(&c->ev_join[l - 1], hipEventDisableTiming));
        }
        if (!c->ev_fork) HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(c->ev_fork, st));
        for (int l = 1; l < nlanes; ++l) HIP_TRY(hipStreamWaitEvent(c->side[l - 1], c->ev_fork, 0));
    }
    // disjoint workspace slices, sized up front by what a lane's runs can take at most (operand + scales + 8 split-K slabs of its rows)
    size_t lane_base[4] = { 0, 0, 0, 0 };
    {
        const size_t fixed = (size_t) mfma_npad(N) * mfma_kpad(K) * 2 + (((size_t) mfma_npad(N) * 4 + 255) & ~(size_t) 255) + 1024;
        size_t rows[4] = { 0, 0, 0, 0 };
        for (int r = 0; r < nruns; ++r)
            for (int k = runs[r].i; k < runs[r].j; ++k) rows[runs[r].lane] += (size_t) ws[k].M;
        size_t base = 0;
        for (int l = 0; l < nlanes; ++l) { lane_base[l] = base; base += (fixed + 8 * (size_t) N * rows[l] * sizeof(float) + 255) & ~(size_t) 255; }
        if (nlanes == 1) lane_base[0] = 0;
    }
    int lane_key[4] = { -1, -1, -1, -1 };
    int rc = QMM_OK;
    // side lanes first: their (small) launches are in the queues when the caller's stream starts on the big run
    for (int pass = 0; pass < 2 && rc == QMM_OK; ++pass)
        for (int r = 0; r < nruns && rc == QMM_OK; ++r) {
            const Run & R = runs[r];
            if ((pass == 0) != (R.lane != 0)) continue;
            c->ws_base = lane_base[R.lane];
            hipStream_t ls = R.lane ? c->side[R.lane - 1] : st;
            const bool reuse = R.key == lane_key[R.lane];
            rc = R.j - R.i > 1 ? mfma_mul_mat_group(c, ls, ws[R.i].type, ws + R.i, R.j - R.i, K, x, N, ldx, reuse)
                               : mfma_mul_mat(c, ls, ws[R.i].type, ws[R.i].w, ws[R.i].w_row_bytes, K, ws[R.i].M, x, N, ldx, ws[R.i].dst, ws[R.i].ldd, reuse);
            lane_key[R.lane] = R.key;
        }
    c->ws_base = 0;
    for (int l = 1; l < nlanes; ++l) {                           // join on every path: a capture must not end with a dangling side stream
        const hipError_t e1 = hipEventRecord(c->ev_join[l - 1], c->side[l - 1]);
        const hipError_t e2 = hipStreamWaitEvent(st, c->ev_join[l - 1], 0);
        if (rc == QMM_OK && (e1 != hipSuccess || e2 != hipSuccess)) rc = fail(QMM_EHIP, "qmm_mul_mat_group: joining the side stream failed");
    }
    return rc;
}

// dst = W * (silu(gate) .* up) for a prompt batch: ffn_down with the SwiGLU product formed by the activation prep of the MFMA path
// (one workgroup per token row reads both rows), so the product never exists in HBM.  Few-token batches fold it on the
// producer side instead (qmm_mv_extra.swiglu).
int qmm_mul_mat_swiglu_in(qmm_ctx * c, int type, const void * w, int64_t w_row_bytes, int64_t K, int64_t M, const float * gate, int64_t ld_gate,
                          const float * up, int64_t ld_up, int64_t N, float * dst, int64_t ldd, void * stream) {
    if (!c) return fail(QMM_EINVAL, "qmm_mul_mat_swiglu_in: NULL context");
    if (N <= QMM_MATVEC_MAX_N) return fail(QMM_EUNSUPPORTED, "qmm_mul_mat_swiglu_in: batches above %d tokens (below: qmm_mv_extra.swiglu)", QMM_MATVEC_MAX_N);
    if (c->prec != QMM_PREC_F16_Q8) return fail(QMM_EUNSUPPORTED, "qmm_mul_mat_swiglu_in: only in the default prefill mode");
    if ((uintptr_t) up % 16 || ld_up % 4 || ld_up < K) return fail(QMM_EINVAL, "qmm_mul_mat_swiglu_in: up must be 16-byte aligned, ld_up %% 4 == 0, ld_up >= K");
    const qmm_weight ws = { w, w_row_bytes, M, dst, ldd, type };
    c->prep_x2 = up;
    c->prep_ldx2 = ld_up;
    const int rc = mul_mat_group_impl(c, &ws, 1, K, gate, N, ld_gate, stream, nullptr);
    c->prep_x2 = nullptr;
    c->prep_ldx2 = 0;
    return rc;
}

int qmm_mul_mat_group(qmm_ctx * c, const qmm_weight * ws, int nw, int64_t K, const float * x, int64_t N, int64_t ldx, void * stream) {
    return mul_mat_group_impl(c, ws, nw, K, x, N, ldx, stream, nullptr);
}

int qmm_mul_mat_group_norm_supported(qmm_ctx * c, const qmm_weight * ws, int nw, int64_t K, int64_t N) {
    if (!c || !ws || nw < 1 || nw > MV_MAX_GROUP || N <= QMM_MATVEC_MAX_N || K <= 0 || K % 1024 || K > 16384) return 0;
    if (c->prec != QMM_PREC_F16_Q8 || !c->prep_reg) return 0;
    for (int i = 0; i < nw; ++i) {
        const int t = type_base(ws[i].type);
        if (!type_known(ws[i].type) || type_act(t) != T_Q8_K || t == T_IQ4_XS || !mfma_regb_supports(c, t)) return 0;
    }
    return 1;
}

int qmm_mul_mat_group_ex(qmm_ctx * c, const qmm_weight * ws, int nw, int64_t K, const float * x, int64_t N, int64_t ldx, const qmm_mv_extra * ex,
                         void * stream) {
    if (c && ex && N > QMM_MATVEC_MAX_N) {
        // a prompt batch: only the norm in front of the group (the activation prep forms it); everything else is the few-token kernels'
        if (!ex->norm_w || ex->swiglu || ex->residual[0] || ex->residual[1] || ex->residual[2] || ex->residual[3])
            return fail(QMM_EUNSUPPORTED, "qmm_mul_mat_group_ex: batches of more than %d tokens take the norm only", QMM_MATVEC_MAX_N);
        if (!qmm_mul_mat_group_norm_supported(c, ws, nw, K, N)) return fail(QMM_EUNSUPPORTED, "qmm_mul_mat_group_ex: this group does not take a fused norm at %lld tokens", (long long) N);
        if (ex->norm_eps < 0.0f || (uintptr_t) ex->norm_w % 16 || (uintptr_t) x % 16 || ldx % 4 || (ex->norm_add && ((uintptr_t) ex->norm_add % 16 || ex->norm_add_ld % 4 || !ex->norm_sum)) ||
            (ex->norm_sum && ((uintptr_t) ex->norm_sum % 16 || ex->norm_sum_ld % 4)))
            return fail(QMM_EINVAL, "qmm_mul_mat_group_ex: norm operands must be 16-byte aligned rows, eps >= 0, a sum buffer with norm_add");
        c->prep_norm = qmm_ctx::prep_norm_t{};
        c->prep_norm.w = ex->norm_w;  c->prep_norm.eps = ex->norm_eps;
        c->prep_norm.add = ex->norm_add;  c->prep_norm.ld_add = ex->norm_add_ld;
        c->prep_norm.sum = ex->norm_add ? ex->norm_sum : nullptr;  c->prep_norm.ld_sum = ex->norm_sum_ld;
        const int rc = mul_mat_group_impl(c, ws, nw, K, x, N, ldx, stream, nullptr);
        c->prep_norm = qmm_ctx::prep_norm_t{};
        return rc;
    }
    return mul_mat_group_impl(c, ws, nw, K, x, N, ldx, stream, ex);
}

int qmm_mul_mat(qmm_ctx * c, int type, const void * w, int64_t rb, int64_t K, int64_t M,
                const float * x, int64_t N, int64_t ldx, float * dst, int64_t ldd, void * stream) {
    qmm_weight ws = { w, rb, M, dst, ldd, type };
    return qmm_mul_mat_group(c, &ws, 1, K, x, N, ldx, stream);
}

int qmm_mul_mat_id(qmm_ctx * c, int type, const void * as, int64_t rb, int64_t expert_bytes,
                   int64_t K, int64_t M, int64_t n_expert,
                   const float * b, int64_t ne11, int64_t b_nb1, int64_t b_nb2,
                   const int32_t * ids, int64_t n_used, int64_t n_tokens, int64_t ids_nb1,
                   float * dst, int64_t d_nb1, int64_t d_nb2, void * stream) {
    if (!c) return fail(QMM_EINVAL, "null ctx");
    int rc = check_mm(type, as, rb, K, b, K, "qmm_mul_mat_id");
    if (rc) return rc;
    if (b_nb1 % 16 || b_nb2 % 16 || ids_nb1 % 4 || d_nb1 % 4 || d_nb2 % 4 || expert_bytes % 2 || (ne11 != 1 && ne11 != n_used))
        return fail(QMM_EINVAL, "qmm_mul_mat_id: strides / ne11");
    if (n_tokens <= 0 || n_used <= 0 || M <= 0) return QMM_OK;
    HIP_TRY(hipSetDevice(c->device));
    QMM_CHAIN_FLUSH(c);
    c->mfma_calls++;
    c->id_calls++;
    return moe_mul_mat_id(c, c->s(stream), type, as, rb, expert_bytes, K, M, n_expert, b, ne11, b_nb1, b_nb2,
                          ids, n_used, n_tokens, ids_nb1, dst, d_nb1, d_nb2);
}

int qmm_mul_mat_id_pair(qmm_ctx * c, int type, const void * as0, const void * as1, int64_t rb, int64_t expert_bytes,
                        int64_t K, int64_t M, int64_t n_expert,
                        const float * b, int64_t ne11, int64_t b_nb1, int64_t b_nb2,
                        const int32_t * ids, int64_t n_used, int64_t n_tokens, int64_t ids_nb1,
                        float * dst0, float * dst1, int64_t d_nb1, int64_t d_nb2, void * stream) {
    if (!c) return fail(QMM_EINVAL, "null ctx");
    if (!as1 || !dst1) return fail(QMM_EINVAL, "qmm_mul_mat_id_pair: second tensor missing");
    int rc = check_mm(type, as0, rb, K, b, K, "qmm_mul_mat_id_pair");
    if (rc) return rc;
    if (b_nb1 % 16 || b_nb2 % 16 || ids_nb1 % 4 || d_nb1 % 4 || d_nb2 % 4 || expert_bytes % 2 || (ne11 != 1 && ne11 != n_used))
        return fail(QMM_EINVAL, "qmm_mul_mat_id_pair: strides / ne11");
    if (n_tokens <= 0 || n_used <= 0 || M <= 0) return QMM_OK;
    HIP_TRY(hipSetDevice(c->device));
    QMM_CHAIN_FLUSH(c);
    c->mfma_calls++;
    c->id_calls++;
    return moe_mul_mat_id(c, c->s(stream), type, as0, rb, expert_bytes, K, M, n_expert, b, ne11, b_nb1, b_nb2,
                          ids, n_used, n_tokens, ids_nb1, dst0, d_nb1, d_nb2, as1, dst1);
}

int qmm_mul_mat_id_swiglu_supported(int64_t n_used, int64_t n_tokens) {
    return n_used > 0 && n_tokens > 0 && n_used * n_tokens <= MOE_MATVEC_MAX_PAIRS;
}

int qmm_mul_mat_id_swiglu(qmm_ctx * c, int type, const void * as_gate, const void * as_up, int64_t rb, int64_t expert_bytes,
                          int64_t K, int64_t M, int64_t n_expert,
                          const float * b, int64_t ne11, int64_t b_nb1, int64_t b_nb2,
                          const int32_t * ids, int64_t n_used, int64_t n_tokens, int64_t ids_nb1,
                          float * dst, int64_t d_nb1, int64_t d_nb2, void * stream) {
    if (!c) return fail(QMM_EINVAL, "null ctx");
    if (!as_up || !dst) return fail(QMM_EINVAL, "qmm_mul_mat_id_swiglu: second tensor missing");
    int rc = check_mm(type, as_gate, rb, K, b, K, "qmm_mul_mat_id_swiglu");
    if (rc) return rc;
    if (b_nb1 % 16 || b_nb2 % 16 || ids_nb1 % 4 || d_nb1 % 4 || d_nb2 % 4 || expert_bytes % 2 || (ne11 != 1 && ne11 != n_used))
        return fail(QMM_EINVAL, "qmm_mul_mat_id_swiglu: strides / ne11");
    if (!qmm_mul_mat_id_swiglu_supported(n_used, n_tokens)) return fail(QMM_EUNSUPPORTED, "qmm_mul_mat_id_swiglu: more than %d (token, slot) pairs", MOE_MATVEC_MAX_PAIRS);
    if (M <= 0) return QMM_OK;
    HIP_TRY(hipSetDevice(c->device));
    QMM_CHAIN_FLUSH(c);
    c->id_calls++;
    hipStream_t st = c->s(stream);
#define QMM_MVIDS(TT)                                                                                                                  \
    return launch_matvec_id_swiglu<TT>(c, st, as_gate, as_up, rb, expert_bytes, (int) K, (int) M, (int) n_expert, b, (int) ne11, b_nb1 / 4, b_nb2 / 4, ids, \
                                       (int) n_used, (int) n_tokens, ids_nb1 / 4, dst, d_nb1 / 4, d_nb2 / 4)
    QMM_FOR_TYPE(type, QMM_MVIDS)
#undef QMM_MVIDS
}

int qmm_chain_begin(qmm_ctx * c) {
    if (!c) return fail(QMM_EINVAL, "qmm_chain_begin: NULL context");
    if (c->chain_on) return fail(QMM_EINVAL, "qmm_chain_begin: already recording");
    c->chain_on = c->chain_enabled != 0;
    return QMM_OK;
}
int qmm_chain_flush(qmm_ctx * c) {
    if (!c) return fail(QMM_EINVAL, "qmm_chain_flush: NULL context");
    if (!c->chain_on) return QMM_OK;
    HIP_TRY(hipSetDevice(c->device));
    return chain_launch(c);
}
int qmm_chain_end(qmm_ctx * c) {
    if (!c) return fail(QMM_EINVAL, "qmm_chain_end: NULL context");
    int rc = QMM_OK;
    if (c->chain_on) {
        HIP_TRY(hipSetDevice(c->device));
        rc = chain_launch(c);
    }
    c->chain_on = false;
    return rc;
}
int qmm_chain_debug(qmm_ctx * c, void * stamps) {
    if (!c) return fail(QMM_EINVAL, "qmm_chain_debug: NULL context");
    c->chain_dbg = (uint64_t *) stamps;
    return QMM_OK;
}
int qmm_trace_begin(qmm_ctx * c) {
    if (!c) return fail(QMM_EINVAL, "qmm_trace_begin: NULL context");
    if (!c->trace) c->trace = new std::string();
    c->trace->clear();
    return QMM_OK;
}

int qmm_trace_end(qmm_ctx * c, char * buf, size_t len) {
    if (!c || !c->trace) return fail(QMM_EINVAL, "qmm_trace_end: no trace in progress");
    int n = 0;
    for (char ch : *c->trace) n += ch == ';';
    if (buf && len) {
        snprintf(buf, len, "%s", c->trace->c_str());
        if (c->trace->size() >= len) n = fail(QMM_EINVAL, "qmm_trace_end: %zu bytes of labels do not fit the buffer", c->trace->size());
    }
    delete c->trace;
    c->trace = nullptr;
    return n;
}

int qmm_chain_stats(const qmm_ctx * c, int * launches, int * steps) {
    if (!c) return fail(QMM_EINVAL, "qmm_chain_stats: NULL context");
    if (launches) *launches = c->chain_launches;
    if (steps) *steps = c->chain_steps;
    return QMM_OK;
}

} // extern "C"
