// gemm_bench.cpp — stand-alone A/B of the bf16 GEMM kernel variants on the encoder's shapes, through the C ABI only
// (no Python, no torch: starts in a second on a fresh GPU box).
//   hipcc --offload-arch=gfx950 -O2 tools/gemm_bench.cpp -Iinclude -L scl-deepfake-audio-detection_amd -lscl_hip \
//         -Wl,-rpath,'$ORIGIN/../scl-deepfake-audio-detection_amd' -o tools/gemm_bench
//   tools/gemm_bench [B=64] [reps=20]
// For every case the variants run interleaved in one process (median of `rounds`), operands rotate over 3 buffer sets so
// that no launch finds its inputs in L2, and the outputs are compared bit for bit.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "scl_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (uint16_t)(u >> 16); }

struct Case { std::string name; int M, N, K; bool at, bt; int splitk; int extra; };   // extra: 1 = bias+gelu+c2 epilogue, 2 = f32 C + residual

static const int NSETS = 3;

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const char* only = argc > 3 ? argv[3] : nullptr;
    const int M = B * 199;
    std::vector<Case> cases;
    const char* nm[4] = {"qkv", "out", "fc1", "fc2"};
    const int Ns[4] = {3072, 1024, 4096, 1024}, Ks[4] = {1024, 1024, 1024, 4096};
    for (int i = 0; i < 4; ++i) {
        cases.push_back({std::string(nm[i]) + " fwd", M, Ns[i], Ks[i], false, false, 1, i == 2 ? 1 : (i == 0 ? 0 : 2)});
        cases.push_back({std::string(nm[i]) + " dgrad", M, Ks[i], Ns[i], false, true, 1, i == 3 ? 3 : 0});      // fc2's data gradient: x gelu'(R)
        for (int sk : {2, 4}) cases.push_back({std::string(nm[i]) + " wgrad sk" + std::to_string(sk), Ns[i], Ks[i], M, true, true, sk, 0});
    }
    cases.push_back({"conv2 fwd", B * 6399, 512, 1536, false, false, 1, 0});
    cases.push_back({"conv4 fwd", B * 1599, 512, 1536, false, false, 1, 0});
    cases.push_back({"conv2 dgrad-like", B * 6399, 512, 1024, false, true, 1, 0});
    cases.push_back({"square 4096", 4096, 4096, 4096, false, false, 1, 0});
    cases.push_back({"square 8192", 8192, 8192, 8192, false, false, 1, 0});

    if (only) { std::vector<Case> f; for (auto& c : cases) if (c.name.find(only) != std::string::npos) f.push_back(c); cases = f; }
    size_t maxA = 0, maxB = 0, maxC = 0;
    for (auto& c : cases) {
        maxA = std::max(maxA, (size_t)c.M * c.K); maxB = std::max(maxB, (size_t)c.N * c.K);
        maxC = std::max(maxC, (size_t)c.M * c.N * (c.splitk > 1 ? c.splitk : 1));
    }
    std::vector<uint16_t> h(std::max(maxA, maxB) + 8192);
    uint32_t s = 12345u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = f2bf(((int)(s >> 8) - (1 << 23)) * (0.3f / (1 << 23))); }
    uint16_t *dA[NSETS], *dB[NSETS]; void* dC[NSETS]; void* dC2; float *dBias, *dR;
    for (int i = 0; i < NSETS; ++i) {
        CK(hipMalloc(&dA[i], maxA * 2 + 65536)); CK(hipMalloc(&dB[i], maxB * 2 + 65536)); CK(hipMalloc(&dC[i], maxC * 4));
        CK(hipMemcpy(dA[i], h.data() + (i * 977) % 4096, maxA * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB[i], h.data() + (i * 1931 + 7) % 4096, maxB * 2, hipMemcpyHostToDevice));
    }
    CK(hipMalloc(&dC2, maxC * 2)); CK(hipMalloc(&dBias, 65536 * 4)); CK(hipMalloc(&dR, maxC * 4));
    CK(hipMemset(dBias, 0, 65536 * 4)); CK(hipMemset(dR, 0, maxC * 4)); CK(hipMemset(dC2, 0x3F, maxC * 2));      // bf16 0x3F3F = 0.746
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<char> out0, out1;

    struct Var { const char* name; int flags; };
    const Var vars3[] = {{"t128", SCL_GEMM_NO_W8 | SCL_GEMM_NO_P8 | SCL_GEMM_NO_BIG}, {"p8", SCL_GEMM_FORCE_P8 | SCL_GEMM_NO_W8}, {"w8", SCL_GEMM_FORCE_W8}};
    const Var vars1[] = {{"w8", SCL_GEMM_FORCE_W8}, {"", 0}, {"", 0}};
    const Var* vars = getenv("GEMM_BENCH_W8_ONLY") ? vars1 : vars3;
    const int NV = getenv("GEMM_BENCH_W8_ONLY") ? 1 : 3;      // W8_ONLY: only the automatic choice (vars[0] below), e.g. under rocprofv3 --pmc
    printf("%-18s %6s %5s %5s | %s\n", "case", "M", "N", "K", "variant: us TFLOP/s ... | bitwise vs t128");
    for (auto& c : cases) {
        if (only && c.name.find(only) == std::string::npos) continue;
        auto desc = [&](int set, int vflags) {
            SclGemmDesc d; memset(&d, 0, sizeof d);
            d.A.ptr = dA[set]; d.A.rpb = 0x7fffffff; d.A.cin = 0x7fffffff; d.A.ld = c.at ? c.M : c.K;
            d.B.ptr = dB[set]; d.B.rpb = 0x7fffffff; d.B.cin = 0x7fffffff; d.B.ld = c.bt ? c.N : c.K;
            d.C = dC[set]; d.c_rpb = 0x7fffffff; d.ldc = c.N; d.M = c.M; d.N = c.N; d.K = c.K; d.nb1 = d.nb2 = 1; d.splitk = c.splitk;
            d.alpha = 1.f;
            d.flags = vflags | (c.at ? SCL_GEMM_A_T : 0) | (c.bt ? SCL_GEMM_B_T : 0);
            if (c.splitk > 1) { d.flags |= SCL_GEMM_C_F32; d.c_split_stride = (int64_t)c.M * c.N; }
            if (c.extra == 1) { d.flags |= SCL_GEMM_HAS_BIAS | SCL_GEMM_HAS_C2 | (5 << SCL_GEMM_ACT_SHIFT); d.bias = dBias; d.C2 = dC2; }      // fc1 forward as the encoder launches it
            if (c.extra == 3) { d.flags |= (2 << SCL_GEMM_RMODE_SHIFT) | (4 << SCL_GEMM_RACT_SHIFT); d.R = dC2; }                              // fc2 data gradient x stored gelu' (bf16)
            if (c.extra == 2) { d.flags |= SCL_GEMM_HAS_BIAS | SCL_GEMM_C_F32 | SCL_GEMM_R_F32 | (1 << SCL_GEMM_RMODE_SHIFT); d.bias = dBias; d.R = dR; }
            return d;
        };
        const size_t cbytes = (size_t)c.M * c.N * (c.splitk > 1 ? c.splitk * 4 : (c.extra == 2 ? 4 : 2));
        std::vector<float> med(NV);
        bool same[NV];
        for (int v = 0; v < NV; ++v) {
            CK(hipMemset(dC[0], 0xFF, cbytes));
            SclGemmDesc d = desc(0, vars[v].flags);
            if (scl_gemm_bf16(&d, st) != 0) { printf("launch failed: %s\n", scl_last_error()); return 1; }
            CK(hipStreamSynchronize(st));
            std::vector<char>& o = v == 0 ? out0 : out1;
            o.resize(cbytes);
            CK(hipMemcpy(o.data(), dC[0], cbytes, hipMemcpyDeviceToHost));
            same[v] = v == 0 || memcmp(out0.data(), out1.data(), cbytes) == 0;
        }
        std::vector<std::vector<float>> t(NV);
        for (int round = 0; round < 5; ++round)
            for (int v = 0; v < NV; ++v) {
                CK(hipEventRecord(e0, st));
                for (int r = 0; r < reps; ++r) { SclGemmDesc d = desc(r % NSETS, vars[v].flags); scl_gemm_bf16(&d, st); }
                CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                t[v].push_back(ms * 1e3f / reps);
            }
        const double fl = 2.0 * c.M * c.N * (double)c.K;
        printf("%-18s %6d %5d %5d |", c.name.c_str(), c.M, c.N, c.K);
        for (int v = 0; v < NV; ++v) {
            std::sort(t[v].begin(), t[v].end());
            med[v] = t[v][t[v].size() / 2];
            printf(" %s %7.1f us %6.0f TF |", vars[v].name, med[v], fl / med[v] / 1e6);
        }
        if (NV == 3) printf(" w8/t128 x%.2f  bitwise p8:%d w8:%d\n", med[0] / med[2], (int)same[1], (int)same[2]);
        else printf("\n");
        if (getenv("STAMPS")) {   // one stamped launch of the wide kernel: where a block's time goes
            SclGemmDesc d = desc(1, SCL_GEMM_FORCE_W8 | SCL_GEMM_STAMPS);
            scl_gemm_bf16(&d, st); CK(hipStreamSynchronize(st));
            const long long tn_ = (c.N + 255) / 256, z_ = c.splitk;
            const long long n208 = (c.M + 207) / 208, n256 = (c.M + 255) / 256;
            const long long c208 = ((n208 * tn_ * z_ + 255) / 256) * 13, c256 = ((n256 * tn_ * z_ + 255) / 256) * 16;
            const int nb = (int)std::min(4096ll, (c208 <= c256 ? n208 : n256) * tn_);
            std::vector<unsigned long long> sp(8 * (size_t)nb);
            if (scl_debug_gemm_stamps(sp.data(), nb) == 0) {
                unsigned long long t0 = ~0ull, t1 = 0;
                for (int b = 0; b < nb; ++b) { t0 = std::min(t0, sp[8 * b]); t1 = std::max(t1, sp[8 * b + 6]); }
                std::vector<double> st0, pro, loop, epi, clk;
                for (int b = 0; b < nb; ++b) {
                    st0.push_back((sp[8 * b] - t0) * 0.01); pro.push_back((sp[8 * b + 2] - sp[8 * b]) * 0.01);
                    loop.push_back((sp[8 * b + 4] - sp[8 * b + 2]) * 0.01); epi.push_back((sp[8 * b + 6] - sp[8 * b + 4]) * 0.01);
                    clk.push_back((double)(sp[8 * b + 5] - sp[8 * b + 3]) / std::max(1.0, (double)(sp[8 * b + 4] - sp[8 * b + 2])) * 0.1);
                }
                auto q = [](std::vector<double>& v, double f) { std::sort(v.begin(), v.end()); return v[(size_t)(f * (v.size() - 1))]; };
                printf("    stamps(%d blocks): span %.1f us | start med %.1f max %.1f | prologue med %.1f max %.1f | loop med %.1f max %.1f | epilogue med %.1f max %.1f | loop clock %.2f GHz\n",
                       nb, (t1 - t0) * 0.01, q(st0, .5), q(st0, 1), q(pro, .5), q(pro, 1), q(loop, .5), q(loop, 1), q(epi, .5), q(epi, 1), q(clk, .5));
            }
        }
        fflush(stdout);
    }
    return 0;
}
