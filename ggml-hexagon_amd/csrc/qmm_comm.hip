// RCCL exchange for a row-split MUL_MAT inside ONE process that owns several devices (the ggml plugin with -sm row): the root's
// activations go to every device with ncclBroadcast, the dst slices come back with grouped ncclSend / ncclRecv.  This is the RCCL
// form of what ggml-hexagon_amd/csrc/ggml-mi355x.cpp does with peer copies + events (reference: ggml_cuda_op_mul_mat's per-device
// cudaMemcpyPeerAsync, ggml/src/ggml-cuda/ggml-cuda.cu:1365-1673, its dst placement :1603-1625).  One-process-per-GPU callers
// (bench.py) use torch.distributed's RCCL instead.
//
// librccl.so is opened on first use, not linked: a process that never splits pays nothing, and a Python process that already holds
// torch's own copy of RCCL does not get a second one mapped at load time.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <mutex>
#include <vector>

#include "../../include/ggml_mi355x_qmm.h"
#include "qmm_host.h"

namespace {

// the subset of rccl.h this file calls (rccl/rccl.h:236-720); ncclComm_t is an opaque pointer, the enums are ints
typedef void * comm_t;
enum { NCCL_SUCCESS = 0, NCCL_INT8 = 0 };
struct Rccl {
    void * lib = nullptr;
    int (*CommInitAll)(comm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    const char * (*GetErrorString)(int) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, comm_t, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, comm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, comm_t, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, comm_t, hipStream_t) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

int load_rccl() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.lib) return QMM_OK;
    const char * names[] = { getenv("GGML_MI355X_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    void * h = nullptr;
    for (const char * n : names)
        if (n && *n && (h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) return qmm::fail(QMM_EUNSUPPORTED, "qmm_comm: librccl.so not found (%s)", dlerror());
#define QMM_SYM(field, name)                                                                                   \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name));                                   \
    if (!g_rccl.field) { dlclose(h); return qmm::fail(QMM_EUNSUPPORTED, "qmm_comm: %s missing in librccl", name); }
    QMM_SYM(CommInitAll, "ncclCommInitAll")
    QMM_SYM(CommDestroy, "ncclCommDestroy")
    QMM_SYM(GetErrorString, "ncclGetErrorString")
    QMM_SYM(GroupStart, "ncclGroupStart")
    QMM_SYM(GroupEnd, "ncclGroupEnd")
    QMM_SYM(Broadcast, "ncclBroadcast")
    QMM_SYM(Send, "ncclSend")
    QMM_SYM(Recv, "ncclRecv")
    QMM_SYM(AllGather, "ncclAllGather")
#undef QMM_SYM
    g_rccl.lib = h;
    return QMM_OK;
}

#define RCCL_TRY(expr)                                                                                          \
    do {                                                                                                        \
        const int r_ = (expr);                                                                                  \
        if (r_ != NCCL_SUCCESS) return qmm::fail(QMM_EHIP, "%s failed: %s", #expr, g_rccl.GetErrorString(r_)); \
    } while (0)

}  // namespace

struct qmm_comm {
    std::vector<qmm_ctx *> ctx;      // rank r = ctx[r]
    std::vector<comm_t>    comm;
};

extern "C" {

QMM_API int qmm_comm_create(qmm_ctx * const * ctxs, int n, qmm_comm ** out) {
    if (!ctxs || !out || n < 2) return qmm::fail(QMM_EINVAL, "qmm_comm_create: need >= 2 contexts");
    std::vector<int> devs(n);
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i]) return qmm::fail(QMM_EINVAL, "qmm_comm_create: null context");
        devs[i] = qmm_device(ctxs[i]);
        for (int j = 0; j < i; ++j)
            if (devs[j] == devs[i]) return qmm::fail(QMM_EINVAL, "qmm_comm_create: device %d twice (RCCL wants one rank per device)", devs[i]);
    }
    const int rc = load_rccl();
    if (rc) return rc;
    auto * c = new qmm_comm;
    c->ctx.assign(ctxs, ctxs + n);
    c->comm.assign(n, nullptr);
    const int r = g_rccl.CommInitAll(c->comm.data(), n, devs.data());
    if (r != NCCL_SUCCESS) {
        delete c;
        return qmm::fail(QMM_EHIP, "ncclCommInitAll over %d devices failed: %s", n, g_rccl.GetErrorString(r));
    }
    *out = c;
    return QMM_OK;
}

QMM_API void qmm_comm_destroy(qmm_comm * c) {
    if (!c) return;
    for (comm_t cm : c->comm)
        if (cm) g_rccl.CommDestroy(cm);
    delete c;
}

QMM_API int qmm_comm_size(const qmm_comm * c) { return c ? (int) c->ctx.size() : 0; }

// bufs[r] on rank r's device, `bytes` each; rank `root`'s buffer is the source.  Enqueued on each rank's context stream
// (streams[r], or the context's own when streams is NULL); stream-ordered like every other call of this library.
QMM_API int qmm_comm_broadcast(qmm_comm * c, int root, void * const * bufs, size_t bytes, void * const * streams) {
    if (!c || !bufs || root < 0 || root >= (int) c->ctx.size()) return qmm::fail(QMM_EINVAL, "qmm_comm_broadcast: bad arguments");
    RCCL_TRY(g_rccl.GroupStart());
    for (size_t r = 0; r < c->ctx.size(); ++r) {
        if (hipSetDevice(qmm_device(c->ctx[r])) != hipSuccess) { g_rccl.GroupEnd(); return qmm::fail(QMM_EHIP, "qmm_comm_broadcast: hipSetDevice failed"); }
        hipStream_t st = (hipStream_t) (streams && streams[r] ? streams[r] : qmm_stream(c->ctx[r]));
        const int rr = g_rccl.Broadcast(bufs[root], bufs[r], bytes, NCCL_INT8, root, c->comm[r], st);
        if (rr != NCCL_SUCCESS) { g_rccl.GroupEnd(); return qmm::fail(QMM_EHIP, "ncclBroadcast failed: %s", g_rccl.GetErrorString(rr)); }
    }
    RCCL_TRY(g_rccl.GroupEnd());
    return QMM_OK;
}

// Every rank r != root sends `bytes[r]` from send[r]; the root receives them into recv[r] (its own slice needs no exchange:
// send[root] / recv[root] are ignored).  One grouped launch per rank.
QMM_API int qmm_comm_gather(qmm_comm * c, int root, const void * const * send, void * const * recv, const size_t * bytes,
                            void * const * streams) {
    if (!c || !send || !recv || !bytes || root < 0 || root >= (int) c->ctx.size()) return qmm::fail(QMM_EINVAL, "qmm_comm_gather: bad arguments");
    RCCL_TRY(g_rccl.GroupStart());
    for (size_t r = 0; r < c->ctx.size(); ++r) {
        if ((int) r == root || bytes[r] == 0) continue;
        hipStream_t sr = (hipStream_t) (streams && streams[r] ? streams[r] : qmm_stream(c->ctx[r]));
        hipStream_t s0 = (hipStream_t) (streams && streams[root] ? streams[root] : qmm_stream(c->ctx[root]));
        if (hipSetDevice(qmm_device(c->ctx[r])) != hipSuccess) { g_rccl.GroupEnd(); return qmm::fail(QMM_EHIP, "qmm_comm_gather: hipSetDevice failed"); }
        int rr = g_rccl.Send(send[r], bytes[r], NCCL_INT8, root, c->comm[r], sr);
        if (rr == NCCL_SUCCESS) {
            if (hipSetDevice(qmm_device(c->ctx[root])) != hipSuccess) { g_rccl.GroupEnd(); return qmm::fail(QMM_EHIP, "qmm_comm_gather: hipSetDevice failed"); }
            rr = g_rccl.Recv(recv[r], bytes[r], NCCL_INT8, (int) r, c->comm[root], s0);
        }
        if (rr != NCCL_SUCCESS) { g_rccl.GroupEnd(); return qmm::fail(QMM_EHIP, "ncclSend/ncclRecv failed: %s", g_rccl.GetErrorString(rr)); }
    }
    RCCL_TRY(g_rccl.GroupEnd());
    return QMM_OK;
}

// send[r] (`bytes` on rank r) -> recv[r] (`size * bytes` on every rank, rank order)
QMM_API int qmm_comm_all_gather(qmm_comm * c, const void * const * send, void * const * recv, size_t bytes, void * const * streams) {
    if (!c || !send || !recv) return qmm::fail(QMM_EINVAL, "qmm_comm_all_gather: bad arguments");
    RCCL_TRY(g_rccl.GroupStart());
    for (size_t r = 0; r < c->ctx.size(); ++r) {
        if (hipSetDevice(qmm_device(c->ctx[r])) != hipSuccess) { g_rccl.GroupEnd(); return qmm::fail(QMM_EHIP, "qmm_comm_all_gather: hipSetDevice failed"); }
        hipStream_t st = (hipStream_t) (streams && streams[r] ? streams[r] : qmm_stream(c->ctx[r]));
        const int rr = g_rccl.AllGather(send[r], recv[r], bytes, NCCL_INT8, c->comm[r], st);
        if (rr != NCCL_SUCCESS) { g_rccl.GroupEnd(); return qmm::fail(QMM_EHIP, "ncclAllGather failed: %s", g_rccl.GetErrorString(rr)); }
    }
    RCCL_TRY(g_rccl.GroupEnd());
    return QMM_OK;
}

}  // extern "C"
