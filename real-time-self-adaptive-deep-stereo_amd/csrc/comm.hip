// comm.hip -- the collective of the shared-model mode behind the C-ABI (round 6).
//
// streams are independent -- private models need NO collective.  When the streams of several GPUs adapt ONE model, each step sums the flat fp32
// gradient buffer (+ the 4 loss floats behind it) over the ranks: one RCCL all-reduce over xGMI per contiguous range, between the backward pass and the momentum
// update (the reference itself is single-GPU: Stereo_Online_Adaptation.py:39,114-128 -- the update it applies is the one every rank applies here).
//
// The host side used to issue that all-reduce through torch.distributed between two captured graphs (an extra graph boundary per step, and outside the library).
// Here it is an entry point like any other: plain pointers, an explicit stream, an int status -- and a PLAN OP (MH_OP_ALLREDUCE), so mh_plan_run records it
// where it belongs and a captured step is ONE hipGraph with the collective as a node on a side lane (the [estimators + context + loss] range leaves while the
// pyramid's backward pass still runs).  torch.distributed is left with what it is good at: carrying the 128-byte unique id from rank 0 to the others.
//
// RCCL is resolved at run time (dlopen of librccl.so): the library keeps loading on a box without RCCL, where the private-model path never needs it, and the
// entry points answer MH_ERR_UNSUPPORTED with the loader's message.
#include "mh_common.h"
#include <dlfcn.h>
#include <mutex>
#include <string.h>

namespace {

// the five RCCL entry points used (rccl.h: ncclResult_t = int, ncclComm_t = opaque pointer, ncclUniqueId = 128 opaque bytes passed BY VALUE)
struct UniqueId { char internal[MH_COMM_ID_BYTES]; };
typedef int (*fn_get_version)(int*);
typedef int (*fn_get_unique_id)(UniqueId*);
typedef int (*fn_comm_init_rank)(void**, int, UniqueId, int);
typedef int (*fn_comm_destroy)(void*);
typedef int (*fn_all_reduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_group)(void);
typedef const char* (*fn_error_string)(int);
constexpr int RCCL_FLOAT32 = 7, RCCL_SUM = 0;        // ncclFloat32 / ncclSum (rccl.h: ncclDataType_t, ncclRedOp_t)

struct Rccl {
    void* so = nullptr;
    fn_get_version get_version = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_all_reduce all_reduce = nullptr;
    fn_group group_start = nullptr, group_end = nullptr;
    fn_error_string error_string = nullptr;
    char why[256] = {0};
};
Rccl g_rccl;
std::once_flag g_rccl_once;

const Rccl* rccl() {
    std::call_once(g_rccl_once, [] {
        Rccl& R = g_rccl;
        const char* env = getenv("MADNET_HIP_RCCL");            // a full path, for installations that keep RCCL outside the loader's search path
        const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            R.so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (R.so) break;
            snprintf(R.why, sizeof(R.why), "%s", dlerror());
        }
        if (!R.so) return;
        R.get_version = (fn_get_version)dlsym(R.so, "ncclGetVersion");
        R.get_unique_id = (fn_get_unique_id)dlsym(R.so, "ncclGetUniqueId");
        R.comm_init_rank = (fn_comm_init_rank)dlsym(R.so, "ncclCommInitRank");
        R.comm_destroy = (fn_comm_destroy)dlsym(R.so, "ncclCommDestroy");
        R.all_reduce = (fn_all_reduce)dlsym(R.so, "ncclAllReduce");
        R.group_start = (fn_group)dlsym(R.so, "ncclGroupStart");
        R.group_end = (fn_group)dlsym(R.so, "ncclGroupEnd");
        R.error_string = (fn_error_string)dlsym(R.so, "ncclGetErrorString");
        if (!(R.get_version && R.get_unique_id && R.comm_init_rank && R.comm_destroy && R.all_reduce && R.group_start && R.group_end)) {
            snprintf(R.why, sizeof(R.why), "librccl.so lacks one of ncclGetVersion / ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce / ncclGroupStart / ncclGroupEnd");
            R.so = nullptr;
        }
    });
    return g_rccl.so ? &g_rccl : nullptr;
}

struct Comm {
    unsigned magic;
    void* nccl;
    int rank, world, device;
};
constexpr unsigned COMM_MAGIC = 0x6d68636fu;        // 'mhco'

#define MH_RCCL(R, call, what)                                                                                                  \
    do {                                                                                                                        \
        const int rc_ = (call);                                                                                                 \
        if (rc_ != 0) {                                                                                                         \
            mh_set_error("%s: RCCL error %d (%s)", what, rc_, (R)->error_string ? (R)->error_string(rc_) : "?");               \
            return MH_ERR_COLLECTIVE;                                                                                           \
        }                                                                                                                       \
    } while (0)

}  // namespace

extern "C" int mh_comm_available(void) { return rccl() ? 1 : 0; }

extern "C" int mh_comm_unique_id(void* id) {
    MH_REQUIRE(id, MH_ERR_ARG, "mh_comm_unique_id: null id buffer (MH_COMM_ID_BYTES bytes)");
    const Rccl* R = rccl();
    MH_REQUIRE(R, MH_ERR_UNSUPPORTED, "mh_comm_unique_id: RCCL is not available (%s)", g_rccl.why);
    UniqueId u;
    MH_RCCL(R, R->get_unique_id(&u), "ncclGetUniqueId");
    memcpy(id, u.internal, MH_COMM_ID_BYTES);
    return 0;
}

extern "C" int mh_comm_init(const void* id, int32_t rank, int32_t world, void** comm) {
    MH_REQUIRE(id && comm, MH_ERR_ARG, "mh_comm_init: null argument");
    MH_REQUIRE(world >= 1 && rank >= 0 && rank < world, MH_ERR_ARG, "mh_comm_init: rank %d of %d", rank, world);
    const Rccl* R = rccl();
    MH_REQUIRE(R, MH_ERR_UNSUPPORTED, "mh_comm_init: RCCL is not available (%s)", g_rccl.why);
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev)) { mh_set_error("mh_comm_init: %s", hipGetErrorString(e)); return (int)e; }
    UniqueId u;
    memcpy(u.internal, id, MH_COMM_ID_BYTES);
    void* nc = nullptr;
    MH_RCCL(R, R->comm_init_rank(&nc, world, u, rank), "ncclCommInitRank");      // collective: returns when every rank of `world` has called it with the same id
    Comm* c = new Comm{COMM_MAGIC, nc, rank, world, dev};
    *comm = c;
    return 0;
}

extern "C" int mh_comm_destroy(void* comm) {
    Comm* c = (Comm*)comm;
    MH_REQUIRE(c && c->magic == COMM_MAGIC, MH_ERR_ARG, "mh_comm_destroy: not a communicator of mh_comm_init");
    const Rccl* R = rccl();
    int rc = 0;
    if (R && c->nccl) rc = R->comm_destroy(c->nccl);
    c->magic = 0;
    delete c;
    if (rc != 0) { mh_set_error("ncclCommDestroy: RCCL error %d", rc); return MH_ERR_COLLECTIVE; }
    return 0;
}

extern "C" int mh_comm_info(void* comm, int32_t* rank, int32_t* world, int32_t* version) {
    Comm* c = (Comm*)comm;
    MH_REQUIRE(c && c->magic == COMM_MAGIC, MH_ERR_ARG, "mh_comm_info: not a communicator of mh_comm_init");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (version) {
        const Rccl* R = rccl();
        int v = 0;
        if (R) R->get_version(&v);
        *version = v;
    }
    return 0;
}

// In-place fp32 sum over the ranks of `comm` of n buffers (one fused RCCL group: the ranges of a MAD block + the loss tail travel as ONE launch).  Recordable
// inside a stream capture (RCCL's kernels become nodes of the graph); every rank must issue the same sequence of calls with the same counts.
extern "C" int mh_allreduce_sum(float* const* bufs, const int64_t* counts, int32_t n, void* comm, void* stream) {
    Comm* c = (Comm*)comm;
    MH_REQUIRE(c && c->magic == COMM_MAGIC, MH_ERR_ARG, "mh_allreduce_sum: not a communicator of mh_comm_init");
    MH_REQUIRE(bufs && counts && n >= 1 && n <= MH_ALLREDUCE_MAX_BUFS, MH_ERR_ARG, "mh_allreduce_sum: 1 .. %d buffers", MH_ALLREDUCE_MAX_BUFS);
    for (int i = 0; i < n; ++i) MH_REQUIRE(bufs[i] && counts[i] > 0, MH_ERR_ARG, "mh_allreduce_sum: buffer %d: null pointer or non-positive count", i);
    const Rccl* R = rccl();
    MH_REQUIRE(R, MH_ERR_UNSUPPORTED, "mh_allreduce_sum: RCCL is not available (%s)", g_rccl.why);
    if (n > 1) MH_RCCL(R, R->group_start(), "ncclGroupStart");
    int bad = 0;
    for (int i = 0; i < n && !bad; ++i) bad = R->all_reduce(bufs[i], bufs[i], (size_t)counts[i], RCCL_FLOAT32, RCCL_SUM, c->nccl, (hipStream_t)stream);
    if (n > 1) { const int ge = R->group_end(); if (!bad) bad = ge; }
    if (bad) { mh_set_error("ncclAllReduce: RCCL error %d (%s)", bad, R->error_string ? R->error_string(bad) : "?"); return MH_ERR_COLLECTIVE; }
    mh_note_kernel("rccl all-reduce (sum, fp32) of %d range%s, world %d", n, n == 1 ? "" : "s", c->world);
    return 0;
}
