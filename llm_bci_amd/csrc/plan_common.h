// plan_common.h — pieces shared by the model-level orchestration files (ndt1.hip, itransformer.hip):
// flat-parameter bookkeeping, workspace carving, GEMM descriptor helpers, weight-gradient queue.
#pragma once
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "kernels.h"

namespace nbci {

struct PInfo {
    std::string name;
    int64_t off, numel;
    int rows, cols, seg;
};

constexpr int NREP = 32;       // replicated accumulators for the 1-D parameters' gradients
constexpr int64_t PALIGN = 8;  // elements: 32 B in f32, 16 B in bf16

static inline size_t bump(size_t& cur, size_t bytes) {
    cur = (cur + 255) / 256 * 256;
    const size_t o = cur;
    cur += bytes;
    return o;
}

// ---- small helpers for describing GEMMs ------------------------------------------------------
static inline nbci_operand op(const void* base, size_t es, int64_t elem_off, int64_t ld, int kmajor, int rpb = 0,
                       int64_t gstride = 0, int64_t zs1 = 0, int64_t zs2 = 0) {
    nbci_operand o;
    o.ptr = (const char*)base + elem_off * (int64_t)es;
    o.ld = ld; o.kmajor = kmajor; o.rpb = rpb; o.gstride = gstride; o.zs1 = zs1; o.zs2 = zs2;
    return o;
}

static inline nbci_gemm_desc gd(int M, int N, int K, int dtype, nbci_operand A, nbci_operand B, void* C, int64_t ldc,
                         int c_dtype) {
    nbci_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.M = M; d.N = N; d.K = K; d.in_dtype = dtype; d.A = A; d.B = B; d.C = C; d.ldc = ldc; d.c_dtype = c_dtype;
    d.batch = 1; d.zdiv = 1; d.splitk = 1; d.alpha = 1.f;
    return d;
}

// weight-gradient GEMMs have K = tokens (huge) and few output tiles: split K so ~2 blocks/CU are
// busy; partials are combined with f32 atomics straight into the (accumulating) grad buffer.
static inline int wgrad_splitk(int M, int N, int K, int dtype) {
    const int tiles = ((M + 127) / 128) * ((N + 127) / 128);
    const int bk = dtype == NBCI_BF16 ? 64 : 16;
    const int ktiles = (K + bk - 1) / bk;
    int sk = (512 + tiles - 1) / tiles;
    const int max_by_k = std::max(1, ktiles / 8);
    sk = std::max(1, std::min(sk, max_by_k));
    return std::min(sk, 64);
}

// dW[M][N] += A^T B over tokens, both operands row-major-in-k. grads accumulate (zero_grad is the
// caller's memset), so splitk==1 runs with beta=1 and splitk>1 with atomics.
static inline int wgrad(hipStream_t s, int dtype, int M, int N, int K, nbci_operand A, nbci_operand B, float* dW, int64_t ldw) {
    nbci_gemm_desc d = gd(M, N, K, dtype, A, B, dW, ldw, NBCI_F32);
    d.splitk = wgrad_splitk(M, N, K, dtype);
    if (d.splitk == 1) d.beta = 1.f;
    return gemm_launch_timed(d, s);
}

// A layer's weight gradients are queued and issued as ONE grouped launch (full-K tiles, beta = 1, no
// split-K atomics) once all their operands exist; in f32 mode they run one by one as before.
struct WgradQueue {
    nbci_gemm_desc d[6];
    int n = 0;
    int dtype;
    hipStream_t s;
    int push(int M, int N, int K, nbci_operand A, nbci_operand B, float* dW, int64_t ldw) {
        if (dtype != NBCI_BF16 || n >= 6) return wgrad(s, dtype, M, N, K, A, B, dW, ldw);
        d[n] = gd(M, N, K, dtype, A, B, dW, ldw, NBCI_F32);
        d[n].beta = 1.f;
        ++n;
        return NBCI_OK;
    }
    int flush() {
        if (n == 0) return NBCI_OK;
        // full-K tiles only pay off when the group fills the chip: with few output tiles and a very long K (narrow
        // models, huge row counts) split-K launches are the better shape
        int tiles = 0;
        for (int i = 0; i < n; ++i) tiles += ((d[i].M + 127) / 128) * ((d[i].N + 127) / 128);
        int rc = NBCI_OK;
        if (tiles >= 192) {
            rc = gemm_grouped_launch_timed(d, n, s);
        } else {
            for (int i = 0; i < n && rc == NBCI_OK; ++i)
                rc = wgrad(s, dtype, d[i].M, d[i].N, d[i].K, d[i].A, d[i].B, (float*)d[i].C, d[i].ldc);
        }
        n = 0;
        return rc;
    }
};

// TRY with a profiling scope (kernels.h ProfScope): name = the kernel's symbol (a launcher may refine it), algorithmic flops / bytes
#define TRYP(name, flops, bytes, s, call)                              \
    do {                                                               \
        ::nbci::ProfScope _ps((name), (double)(flops), (double)(bytes), (s)); \
        int _r = (call);                                               \
        if (_r != NBCI_OK) return _r;                                  \
    } while (0)

#define TRY(x)                    \
    do {                          \
        int _r = (x);             \
        if (_r != NBCI_OK) return _r; \
    } while (0)


}  // namespace nbci
