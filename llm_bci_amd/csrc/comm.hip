// comm.hip — the data-parallel exchange step behind the C-ABI (SURVEY §8(b) nbci_allreduce_bucket): a thin veneer over RCCL,
// for hosts that are not PyTorch (the Python host keeps using torch.distributed, whose "nccl" backend IS RCCL). librccl is
// opened lazily with dlopen — libnbci.so has no link-time dependency on it, so the library loads (and every non-DP entry point
// works) on a box without RCCL — and when the process already has an RCCL loaded (PyTorch's) that copy is reused.
// Replaces: accelerator.backward's DDP bucket all-reduce (models/trainer.py:260-262,339). Mean = SUM here + the 1/W folded into
// nbci_adamw's grad_scale.
//
// Also here: nbci_debug_occupy_cus, a measurement aid that parks workgroups on k CUs for a given time on a second stream — what
// an overlapped RCCL all-reduce does to the CUs the step's GEMM grids were sized for (tools/dp_cu_footprint.py).
#include <dlfcn.h>

#include <cstring>
#include <mutex>

#include "kernels.h"
#include "nbci_common.h"
#include "../../include/nbci.h"

namespace nbci {
namespace {
struct UniqueId { char internal[128]; };   // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void* Comm;
struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
std::once_flag g_once;
int g_load_rc = NBCI_OK;

void load_rccl() {
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
        g_rccl.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.h) break;
    }
    if (!g_rccl.h) { g_load_rc = fail(NBCI_EHIP, std::string("comm: cannot open librccl.so: ") + dlerror()); return; }
    g_rccl.GetUniqueId = (int (*)(UniqueId*))dlsym(g_rccl.h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(Comm*, int, UniqueId, int))dlsym(g_rccl.h, "ncclCommInitRank");
    g_rccl.CommDestroy = (int (*)(Comm))dlsym(g_rccl.h, "ncclCommDestroy");
    g_rccl.AllReduce = (int (*)(const void*, void*, size_t, int, int, Comm, hipStream_t))dlsym(g_rccl.h, "ncclAllReduce");
    g_rccl.GetErrorString = (const char* (*)(int))dlsym(g_rccl.h, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllReduce)
        g_load_rc = fail(NBCI_EHIP, "comm: librccl.so lacks an expected symbol");
}
int rccl() {
    std::call_once(g_once, load_rccl);
    return g_load_rc;
}
int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(NBCI_EHIP, std::string(what) + " launch: " + hipGetErrorString(e));
    return NBCI_OK;
}
int check_nccl(int rc, const char* what) {
    if (rc == 0) return NBCI_OK;
    return fail(NBCI_EHIP, std::string("comm: ") + what + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error"));
}

// parks one workgroup per requested CU slot until `ticks` of the 100 MHz wall clock have passed (bounded: every wave leaves)
__global__ __launch_bounds__(256) void occupy_kernel(unsigned long long ticks, unsigned* sink) {
    extern __shared__ char lds[];
    const unsigned long long t0 = wall_clock64();
    unsigned n = 0;
    while (wall_clock64() - t0 < ticks && n < (1u << 26)) {
        __builtin_amdgcn_s_sleep(32);
        ++n;
    }
    if (ticks == ~0ull) { lds[threadIdx.x] = (char)n; sink[0] = lds[(threadIdx.x + 1) & 255]; }   // (keeps the LDS allocation alive; never taken)
}
}  // namespace
}  // namespace nbci

extern "C" {

int nbci_comm_unique_id(void* id128) {
    if (!id128) return nbci::fail(NBCI_EINVAL, "comm: null id buffer");
    int rc = nbci::rccl();
    if (rc != NBCI_OK) return rc;
    return nbci::check_nccl(nbci::g_rccl.GetUniqueId((nbci::UniqueId*)id128), "ncclGetUniqueId");
}

int nbci_comm_create(nbci_comm* comm, int32_t world_size, int32_t rank, const void* id128) {
    if (!comm || !id128 || world_size < 1 || rank < 0 || rank >= world_size) return nbci::fail(NBCI_EINVAL, "comm: bad arguments");
    int rc = nbci::rccl();
    if (rc != NBCI_OK) return rc;
    nbci::UniqueId id;
    memcpy(&id, id128, sizeof(id));
    nbci::Comm c = nullptr;
    rc = nbci::check_nccl(nbci::g_rccl.CommInitRank(&c, world_size, id, rank), "ncclCommInitRank");
    if (rc != NBCI_OK) return rc;
    *comm = c;
    return NBCI_OK;
}

void nbci_comm_destroy(nbci_comm comm) {
    if (comm && nbci::g_rccl.CommDestroy) (void)nbci::g_rccl.CommDestroy(comm);
}

int nbci_allreduce_bucket(nbci_comm comm, void* buf, int64_t n, int32_t dtype, nbci_stream_t stream) {
    if (!comm || !buf || n < 0 || (dtype != NBCI_F32 && dtype != NBCI_BF16)) return nbci::fail(NBCI_EINVAL, "allreduce: bad arguments");
    if (n == 0) return NBCI_OK;
    int rc = nbci::rccl();
    if (rc != NBCI_OK) return rc;
    // in place, SUM (ncclSum = 0); ncclFloat32 = 7, ncclBfloat16 = 9 (rccl.h)
    return nbci::check_nccl(nbci::g_rccl.AllReduce(buf, buf, (size_t)n, dtype == NBCI_F32 ? 7 : 9, 0, comm, (hipStream_t)stream), "ncclAllReduce");
}

int nbci_debug_occupy_cus(int32_t n_workgroups, int32_t lds_bytes, double microseconds, nbci_stream_t stream) {
    if (n_workgroups <= 0 || n_workgroups > 1024 || lds_bytes < 0 || lds_bytes > 160 * 1024 || microseconds < 0 || microseconds > 5e5)
        return nbci::fail(NBCI_EINVAL, "occupy: bad arguments");
    { const int r = nbci::ensure_dyn_lds((const void*)nbci::occupy_kernel, 160 * 1024, "occupy"); if (r != NBCI_OK) return r; }
    hipLaunchKernelGGL(nbci::occupy_kernel, dim3(n_workgroups), dim3(256), lds_bytes, (hipStream_t)stream,
                       (unsigned long long)(microseconds * 100.0), (unsigned*)nullptr);
    return nbci::check_launch("occupy");
}

int nbci_set_available_cus(int32_t cus) {
    if (cus < 1 || cus > 256) return nbci::fail(NBCI_EINVAL, "available_cus must be in 1..256");
    nbci::set_available_cus(cus);
    return NBCI_OK;
}

}  // extern "C"
