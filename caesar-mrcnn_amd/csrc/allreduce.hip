// Data-parallel gradient exchange over RCCL / xGMI behind the C-ABI: the MI355X replacement of the
// reference's in-graph tower aggregation (mrcnn/parallel_model.py:54-104 -- shared variables, implicit gradient sum).
// One process per GPU; every rank holds the flat float32 gradient buffer of params.ParamLayout and sums contiguous
// ranges of it with its peers as soon as the backward pass has finalised them, on a side stream.
//
// RCCL is bound at run time (dlopen + dlsym): the kernel library has no link-time dependency on it, loads on a box
// without RCCL, and -- when the caller passes the path of the librccl the process already uses (PyTorch's) -- shares that
// one instance instead of bringing a second copy into the process.
//
// Two exchange algorithms (same result up to float32 summation order; both leave every rank with identical bits):
//   MRCNN_ALLREDUCE_RCCL    ncclAllReduce(sum) in place -- RCCL picks ring / tree / direct by message size.
//   MRCNN_ALLREDUCE_DIRECT  reduce-scatter + all-gather written out as grouped point-to-point transfers: the range is cut
//                           into `world` chunks, rank r receives chunk r from every peer (one ncclSend/ncclRecv pair per
//                           peer, all in one group: xGMI is point-to-point, 7 links per GPU, so the 7 transfers run on 7
//                           different links at once), sums the `world` copies in RANK ORDER with one kernel (fixed order:
//                           bitwise reproducible, and the owner is the only rank that adds), and sends the result back
//                           the same way.  Per link this moves 2*S/world bytes instead of a ring's 2*S*(world-1)/world
//                           (63.6 MB vs 445 MB for ResNet-101's 254.5 MB).  Needs (world-1) * chunk floats
//                           of scratch.  Not exercised on more than one GPU in this pool: opt-in.
#include "common.h"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

namespace {
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
} g_rccl;

char g_last_error[256] = "";

void set_error(const char* what, const char* detail) { snprintf(g_last_error, sizeof(g_last_error), "%s: %s", what, detail ? detail : ""); }

int check(ncclResult_t r, const char* what) {
    if (r == ncclSuccess) return MRCNN_OK;
    set_error(what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
    return MRCNN_ERR_LAUNCH;
}

struct Comm {
    ncclComm_t comm;
    int rank, world;
};
}  // namespace

// chunk c of the range [start, end) cut into `world` pieces of `chunk` floats (the last ones may be short or empty)
static inline void chunk_of(int64_t start, int64_t end, int64_t chunk, int c, int64_t* off, int64_t* len) {
    int64_t o = start + (int64_t)c * chunk;
    if (o > end) o = end;
    int64_t e = o + chunk;
    if (e > end) e = end;
    *off = o; *len = e - o;
}

// own[i] = sum over ranks r = 0 .. world-1 of copy_r[i], in rank order; copy_rank = own (in place), the others sit in
// scratch slots (slot s = peer index with the own rank left out)
__global__ void allreduce_sum_chunks_kernel(float* own, const float* __restrict__ scratch, int64_t slot_stride, int64_t n, int rank,
                                            int world) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        bool first = true;
        for (int r = 0; r < world; ++r) {
            const float v = r == rank ? own[i] : scratch[(int64_t)(r < rank ? r : r - 1) * slot_stride + i];
            s = first ? v : s + v;
            first = false;
        }
        own[i] = s;
    }
}

extern "C" const char* mrcnn_allreduce_last_error(void) { return g_last_error; }

extern "C" int mrcnn_allreduce_load(const char* librccl_path) {
    if (g_rccl.ok) return MRCNN_OK;
    const char* names[] = {librccl_path, "librccl.so.1", "librccl.so"};
    void* h = nullptr;
    for (const char* n : names) {
        if (!n || !n[0]) continue;
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);          // the instance the process already has, if any
        if (!h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) { set_error("dlopen(librccl)", dlerror()); return MRCNN_ERR_UNSUPPORTED; }
    g_rccl.handle = h;
#define MRCNN_SYM(field, name)                                                   \
    *(void**)(&g_rccl.field) = dlsym(h, name);                                   \
    if (!g_rccl.field) { set_error("dlsym", name); return MRCNN_ERR_UNSUPPORTED; }
    MRCNN_SYM(GetUniqueId, "ncclGetUniqueId")
    MRCNN_SYM(CommInitRank, "ncclCommInitRank")
    MRCNN_SYM(CommDestroy, "ncclCommDestroy")
    MRCNN_SYM(AllReduce, "ncclAllReduce")
    MRCNN_SYM(Send, "ncclSend")
    MRCNN_SYM(Recv, "ncclRecv")
    MRCNN_SYM(GroupStart, "ncclGroupStart")
    MRCNN_SYM(GroupEnd, "ncclGroupEnd")
    MRCNN_SYM(GetErrorString, "ncclGetErrorString")
#undef MRCNN_SYM
    g_rccl.ok = true;
    return MRCNN_OK;
}

extern "C" int mrcnn_allreduce_unique_id(void* id) {
    if (!id) return MRCNN_ERR_ARG;
    if (!g_rccl.ok && mrcnn_allreduce_load(nullptr) != MRCNN_OK) return MRCNN_ERR_UNSUPPORTED;
    static_assert(sizeof(ncclUniqueId) == MRCNN_UNIQUE_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId u;
    const int rc = check(g_rccl.GetUniqueId(&u), "ncclGetUniqueId");
    if (rc == MRCNN_OK) memcpy(id, &u, sizeof(u));
    return rc;
}

extern "C" int mrcnn_allreduce_init(void** comm, const void* id, int rank, int world) {
    if (!comm || !id || world < 1 || rank < 0 || rank >= world) return MRCNN_ERR_ARG;
    if (!g_rccl.ok && mrcnn_allreduce_load(nullptr) != MRCNN_OK) return MRCNN_ERR_UNSUPPORTED;
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    Comm* c = new Comm;
    c->rank = rank; c->world = world;
    const int rc = check(g_rccl.CommInitRank(&c->comm, world, u, rank), "ncclCommInitRank");   // on the current HIP device
    if (rc != MRCNN_OK) { delete c; return rc; }
    *comm = c;
    return MRCNN_OK;
}

extern "C" int mrcnn_allreduce_destroy(void* comm) {
    if (!comm) return MRCNN_ERR_ARG;
    Comm* c = (Comm*)comm;
    const int rc = check(g_rccl.CommDestroy(c->comm), "ncclCommDestroy");
    delete c;
    return rc;
}

static inline int64_t direct_chunk(int64_t n, int world) {
    int64_t chunk = (n + world - 1) / world;
    return (chunk + 63) / 64 * 64;                              // 256-byte granules: every chunk starts 16-byte aligned
}

// The direct form as a PLAN: what rank `me` sends to / receives from peer p in each of the two phases, as offsets and
// lengths in floats.  Phase 0 (reduce-scatter): send chunk p of my gradients, receive my chunk of p's gradients into scratch
// slot (p < me ? p : p - 1).  Phase 1 (all-gather): send my summed chunk, receive p's summed chunk in place.  The real
// exchange, the single-GPU simulation and the host-side plan query all walk this one function, so the pairing of every
// ncclSend with the matching ncclRecv (equal lengths on both ends, or RCCL hangs / corrupts) is checked without a second GPU.
struct DirectOp {
    int64_t send_off, send_len;      // in grads (floats from the buffer's start)
    int64_t recv_off, recv_len;      // phase 0: in scratch; phase 1: in grads
};

static inline DirectOp direct_op(int64_t start, int64_t end, int world, int me, int p, int phase) {
    const int64_t chunk = direct_chunk(end - start, world);
    int64_t my_off, my_len, off, len;
    chunk_of(start, end, chunk, me, &my_off, &my_len);
    chunk_of(start, end, chunk, p, &off, &len);
    DirectOp o;
    if (phase == 0) {
        o.send_off = off; o.send_len = len;
        o.recv_off = (int64_t)(p < me ? p : p - 1) * chunk; o.recv_len = my_len;
    } else {
        o.send_off = my_off; o.send_len = my_len;
        o.recv_off = off; o.recv_len = len;
    }
    return o;
}

extern "C" int mrcnn_allreduce_direct_plan(int world, int rank, int64_t start, int64_t end, int phase, int64_t* plan,
                                           int64_t* own_off, int64_t* own_len, int64_t* slot_stride) {
    if (world < 1 || rank < 0 || rank >= world || start < 0 || end < start || phase < 0 || phase > 1 || !plan) return MRCNN_ERR_ARG;
    for (int p = 0; p < world; ++p) {
        DirectOp o = {0, 0, 0, 0};
        if (p != rank && end > start) o = direct_op(start, end, world, rank, p, phase);
        plan[p * 4 + 0] = o.send_off; plan[p * 4 + 1] = o.send_len; plan[p * 4 + 2] = o.recv_off; plan[p * 4 + 3] = o.recv_len;
    }
    const int64_t chunk = end > start ? direct_chunk(end - start, world) : 0;
    int64_t o = start, l = 0;
    if (end > start) chunk_of(start, end, chunk, rank, &o, &l);
    if (own_off) *own_off = o;
    if (own_len) *own_len = l;
    if (slot_stride) *slot_stride = chunk;
    return MRCNN_OK;
}

static int launch_sum_chunks(float* own, const float* scratch, int64_t chunk, int64_t my_len, int me, int W, hipStream_t s) {
    int64_t blocks = (my_len + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(allreduce_sum_chunks_kernel, dim3((unsigned)blocks), dim3(256), 0, s, own, scratch, chunk, my_len, me, W);
    return mrcnn_launch_status();
}

// The direct exchange of `world` FABRICATED ranks on one GPU: grads[r] / scratch[r] are rank r's buffers (host arrays of
// device pointers), every planned send is paired with the receive its peer planned (lengths must agree: a mismatch is the
// error a real node would turn into a hang) and carried out as a device-to-device copy; the owners' sums run the real
// kernel.  Test entry: tests/test_dp_gpu.py::test_direct_allreduce_simulated_ranks.
extern "C" int mrcnn_allreduce_direct_simulate(float* const* grads, float* const* scratch, size_t scratch_bytes, int world,
                                               int64_t start, int64_t end, void* stream) {
    if (!grads || !scratch || world < 2 || world > 64 || start < 0 || end < start) return MRCNN_ERR_ARG;
    if (end == start) return MRCNN_OK;
    hipStream_t s = (hipStream_t)stream;
    const int64_t chunk = direct_chunk(end - start, world);
    if (scratch_bytes < (size_t)(world - 1) * (size_t)chunk * sizeof(float)) return MRCNN_ERR_WORKSPACE;
    for (int phase = 0; phase < 2; ++phase) {
        for (int a = 0; a < world; ++a)
            for (int b = 0; b < world; ++b) {
                if (a == b) continue;
                const DirectOp sa = direct_op(start, end, world, a, b, phase);        // a's send to b ...
                const DirectOp rb = direct_op(start, end, world, b, a, phase);        // ... meets b's receive from a
                if (sa.send_len != rb.recv_len) { set_error("direct plan", "send / receive lengths of a pair differ"); return MRCNN_ERR_LAUNCH; }
                if (sa.send_len == 0) continue;
                float* dst = (phase == 0 ? scratch[b] : grads[b]) + rb.recv_off;
                if (hipMemcpyAsync(dst, grads[a] + sa.send_off, (size_t)sa.send_len * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess)
                    return MRCNN_ERR_LAUNCH;
            }
        if (phase == 0)
            for (int r = 0; r < world; ++r) {
                int64_t my_off, my_len;
                chunk_of(start, end, chunk, r, &my_off, &my_len);
                if (my_len > 0 && launch_sum_chunks(grads[r] + my_off, scratch[r], chunk, my_len, r, world, s) != MRCNN_OK) return MRCNN_ERR_LAUNCH;
            }
    }
    return MRCNN_OK;
}

extern "C" size_t mrcnn_allreduce_scratch(int world, int64_t max_range_floats, int algo) {
    if (algo != MRCNN_ALLREDUCE_DIRECT || world <= 1 || max_range_floats <= 0) return 0;
    return (size_t)(world - 1) * (size_t)direct_chunk(max_range_floats, world) * sizeof(float);
}

extern "C" int mrcnn_allreduce_grad(void* comm, float* grads, int64_t start, int64_t end, int algo, float* scratch,
                                    size_t scratch_bytes, void* stream) {
    if (!comm || !grads || start < 0 || end < start) return MRCNN_ERR_ARG;
    Comm* c = (Comm*)comm;
    const int64_t n = end - start;
    if (n == 0) return MRCNN_OK;
    hipStream_t s = (hipStream_t)stream;
    if (algo == MRCNN_ALLREDUCE_RCCL || c->world == 1) {
        if (c->world == 1 && algo == MRCNN_ALLREDUCE_DIRECT) return MRCNN_OK;       // sum over one rank
        return check(g_rccl.AllReduce(grads + start, grads + start, (size_t)n, ncclFloat, ncclSum, c->comm, s), "ncclAllReduce");
    }
    if (algo != MRCNN_ALLREDUCE_DIRECT) return MRCNN_ERR_ARG;
    const int W = c->world, me = c->rank;
    const int64_t chunk = direct_chunk(n, W);
    if (!scratch || scratch_bytes < (size_t)(W - 1) * (size_t)chunk * sizeof(float)) return MRCNN_ERR_WORKSPACE;
    int64_t my_off, my_len;
    chunk_of(start, end, chunk, me, &my_off, &my_len);
    for (int phase = 0; phase < 2; ++phase) {
        // phase 0, reduce-scatter: my chunk of every peer arrives in scratch, their chunks of mine leave;
        // phase 1, all-gather: the summed chunk goes to every peer, theirs come back in place
        int rc = check(g_rccl.GroupStart(), "ncclGroupStart");
        for (int p = 0; p < W && rc == MRCNN_OK; ++p) {
            if (p == me) continue;
            const DirectOp o = direct_op(start, end, W, me, p, phase);
            if (o.send_len > 0) rc = check(g_rccl.Send(grads + o.send_off, (size_t)o.send_len, ncclFloat, p, c->comm, s), "ncclSend");
            if (rc == MRCNN_OK && o.recv_len > 0)
                rc = check(g_rccl.Recv((phase == 0 ? scratch : grads) + o.recv_off, (size_t)o.recv_len, ncclFloat, p, c->comm, s), "ncclRecv");
        }
        {
            const int rc2 = check(g_rccl.GroupEnd(), "ncclGroupEnd");
            if (rc == MRCNN_OK) rc = rc2;
        }
        if (rc != MRCNN_OK) return rc;
        if (phase == 0 && my_len > 0 && launch_sum_chunks(grads + my_off, scratch, chunk, my_len, me, W, s) != MRCNN_OK) return MRCNN_ERR_LAUNCH;
    }
    return MRCNN_OK;
}
