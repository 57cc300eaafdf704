// rs_finish.h — shared by resstack.hip and graph.hip: SELU, and the "every block adds fp64 partials, the last block finishes" pattern
// behind every BatchNorm of the AASIST back-end (model/wav2vec2_aasist.py:62-155, 158-332, 377-433).
#pragma once
#include "common.h"

namespace {

constexpr float SELU_ALPHA = 1.6732632423543772848170429916717f;
constexpr float SELU_SCALE = 1.0507009873554804934193349852946f;

__device__ __forceinline__ float selu_f(float v) { return v > 0.f ? SELU_SCALE * v : SELU_SCALE * SELU_ALPHA * (__expf(v) - 1.0f); }
__device__ __forceinline__ float selu_grad_from_y(float y) { return y > 0.f ? SELU_SCALE : y + SELU_SCALE * SELU_ALPHA; }

// Every block adds its n fp64 partials to ONE OF RS_NSLOT accumulator rows (row = block index mod RS_NSLOT: 512 blocks on one row would
// serialise 512 atomics per address at the L2) with hardware fp64 atomics; the LAST block to arrive (ticket) sums the rows in index
// order, hands the totals to `fin` and leaves rows and ticket zeroed for the next launch.  fp64 addition order inside a row varies from
// run to run: an order-dependent error of ~1e-16 relative, invisible after the rounding to fp32 that every consumer applies.
constexpr int RS_NSLOT = SCL_RS_NSLOT;
template <class F>
__device__ __forceinline__ void rs_finish(double* acc, unsigned* ticket, int n, const double* mine, double* lds_tot, F fin) {
    __shared__ int is_last;
    const int tid = threadIdx.x;
    // No __threadfence(): at agent scope it writes the XCD's whole L2 back (this kernel has just stored its output map there) — 40 us per
    // launch with 512 blocks doing it.  The partials travel as RETURNING atomics instead: a thread has its old value back only once the
    // addition has been performed at the device-coherent level, the barrier collects all of them, then the ticket goes out.
    if (tid < n) {
        const double old = __hip_atomic_fetch_add(&acc[(blockIdx.x % RS_NSLOT) * n + tid], mine[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("" ::"v"(old));
    }
    __syncthreads();
    if (tid == 0) is_last = (__hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) ? 1 : 0;
    __syncthreads();
    if (!is_last) return;
    if (tid < n) {
        double t = 0.0;
#pragma unroll
        for (int sl = 0; sl < RS_NSLOT; ++sl) {
            t += __hip_atomic_load(&acc[sl * n + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&acc[sl * n + tid], 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        lds_tot[tid] = t;
    }
    if (tid == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    fin(lds_tot);
}

}  // namespace
