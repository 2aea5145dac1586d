// Back-to-back timing of libick_amd.so entry points from C++ (no Python launch overhead).
// Usage: probe_ops [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "ick_amd.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static hipStream_t st;
template <typename F> float timeit(F f, int iters) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 10; ++i) f();
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.f / iters;
}

int main(int argc, char** argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 200;
    CK(hipStreamCreate(&st));
    float *A, *B, *C, *bias;
    size_t na = (size_t)64 * 2048 * 196, nb = (size_t)64 * 6 * 10 * 288 * 32 /* >= KV of 64 samples, S <= 288 */, nc = (size_t)13824 * 2048;
    setvbuf(stdout, nullptr, _IONBF, 0);
    CK(hipMalloc(&A, na * 4)); CK(hipMalloc(&B, nb * 4)); CK(hipMalloc(&C, nc * 4)); CK(hipMalloc(&bias, 65536 * 4));
    std::vector<float> h(na);
    for (size_t i = 0; i < na; ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
    CK(hipMemcpy(A, h.data(), na * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, h.data(), std::min(na, nb) * 4, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, 65536 * 4));
    auto gemm = [&](const char* name, int M, int N, int K, bool feat, int hs) {
        ick_gemm_args a; memset(&a, 0, sizeof(a));
        a.A = A; a.B = B; a.C = C; a.bias = bias; a.M = M; a.N = N; a.K = K; a.alpha = 1.f; a.split_k = 1;
        a.a_extent = (int64_t)na; a.b_extent = (int64_t)nb;
        if (feat) { a.a_rs = 1; a.a_ks = 196; a.a_grp = 196; a.a_gs = (int64_t)K * 196; }
        else { a.a_rs = K; a.a_ks = 1; }
        a.b_rs = K; a.b_ks = 1; a.c_rs = N;
        if (hs) { a.hs_dh = 30; a.hs_dhp = 32; a.hs_H = 10; a.hs_S = hs; a.hs_s0 = 0; a.c_grp = hs; a.c_gs = (int64_t)(N / 300) * 10 * hs * 32; }
        int rc = ick_gemm(&a, st);
        if (rc) { printf("%s rc=%d\n", name, rc); return; }
        float us = timeit([&] { ick_gemm(&a, st); }, iters);
        printf("%-28s M=%5d N=%5d K=%4d : %8.2f us  %6.1f TFLOP/s\n", name, M, N, K, us, 2.0 * M * N * K / us * 1e-6);
    };
    gemm("feat_proj (k-major A)", 12544, 300, 2048, true, 0);
    gemm("feat_proj B=32", 6272, 300, 2048, true, 0);
    gemm("feat_proj B=16", 3136, 300, 2048, true, 0);
    gemm("cross KV image rows B=32", 6272, 1800, 300, false, 196);
    gemm("cross KV image rows", 12544, 1800, 300, false, 196);
    gemm("cross KV entity rows", 1280, 1800, 300, false, 20);
    gemm("vocab", 1280, 10000, 300, false, 0);
    gemm("qkv (head split)", 1280, 900, 300, false, 20);
    gemm("q-proj (head split)", 1280, 300, 300, false, 20);
    gemm("out-proj", 1280, 300, 300, false, 0);
    gemm("ffn1", 1280, 512, 300, false, 0);
    gemm("ffn2", 1280, 300, 512, false, 0);
    gemm("latency: 1 tile", 32, 32, 300, false, 0);
    gemm("latency: K=32", 1280, 300, 32, false, 0);
    gemm("latency: K=64", 1280, 300, 64, false, 0);
    gemm("latency: K=128", 1280, 300, 128, false, 0);
    gemm("latency: K=1200", 1280, 300, 1200, false, 0);
    gemm("latency: 256 tiles", 512, 512, 300, false, 0);
    gemm("latency: 1024 tiles", 1024, 1024, 300, false, 0);
    gemm("decode-step qkv B=64", 64, 900, 300, false, 0);
    gemm("decode-step vocab B=64", 64, 10000, 300, false, 0);
    gemm("square 4096", 4096, 4096, 2048, false, 0);
    // backward shapes: dgrad (B operand k-major) and wgrad (both k-major, split-K atomics)
    auto gemm_bwd = [&](const char* name, int M, int N, int K, bool wgrad, int split) {
        ick_gemm_args a; memset(&a, 0, sizeof(a));
        a.A = A; a.B = B; a.C = C; a.M = M; a.N = N; a.K = K; a.alpha = 1.f; a.split_k = split;
        a.a_extent = (int64_t)na; a.b_extent = (int64_t)nb;
        if (wgrad) { a.a_rs = 1; a.a_ks = M; a.b_rs = 1; a.b_ks = N; }
        else { a.a_rs = K; a.a_ks = 1; a.b_rs = 1; a.b_ks = N; }
        a.c_rs = N;
        if (split > 1 || wgrad) a.flags = ICK_GEMM_ATOMIC;
        int rc = ick_gemm(&a, st);
        if (rc) { printf("%s rc=%d\n", name, rc); return; }
        float us = timeit([&] { ick_gemm(&a, st); }, iters);
        printf("%-28s M=%5d N=%5d K=%5d s=%2d: %8.2f us  %6.1f TFLOP/s\n", name, M, N, K, split, us, 2.0 * M * N * K / us * 1e-6);
    };
    gemm_bwd("dgrad out-proj", 1280, 300, 300, false, 1);
    gemm_bwd("dgrad ffn2->f", 1280, 512, 300, false, 1);
    gemm_bwd("dgrad ffn1->x", 1280, 300, 512, false, 1);
    gemm_bwd("dgrad qkv->x", 1280, 300, 900, false, 1);
    gemm_bwd("dgrad vocab->h", 1280, 300, 10000, false, 1);
    gemm_bwd("dgrad vocab->h split8", 1280, 300, 10000, false, 8);
    gemm_bwd("wgrad out-proj", 300, 300, 1280, true, 5);
    gemm_bwd("wgrad qkv", 900, 300, 1280, true, 5);
    gemm_bwd("wgrad vocab", 10000, 300, 1280, true, 5);
    gemm_bwd("wgrad vocab s=1", 10000, 300, 1280, true, 1);
    gemm_bwd("wgrad vocab s=2", 10000, 300, 1280, true, 2);
    gemm_bwd("wgrad vocab s=3", 10000, 300, 1280, true, 3);
    gemm_bwd("wgrad cross kv s=8", 600, 300, 13824, true, 8);
    gemm_bwd("wgrad cross kv s=24", 600, 300, 13824, true, 24);
    gemm_bwd("wgrad cross kv x3 s=8", 1800, 300, 13824, true, 8);
    gemm_bwd("wgrad cross kv x3 s=16", 1800, 300, 13824, true, 16);
    gemm_bwd("wgrad cross kv", 600, 300, 13824, true, 16);
    // attention
    float* Q = A; float* KV = B; float* O = C;
    auto attn = [&](const char* name, int Bn, int T, int S, int causal) {
        ick_attn_args a; memset(&a, 0, sizeof(a));
        a.Q = Q; a.K = KV; a.V = KV + (size_t)10 * S * 32; a.O = O; a.B = Bn; a.H = 10; a.T = T; a.S = S; a.dh = 30;
        a.q_bs = (int64_t)10 * T * 32; a.q_hs = T * 32; a.q_ts = 32;
        a.k_bs = (int64_t)6 * 10 * S * 32; a.k_hs = S * 32; a.k_ss = 32; a.v_bs = a.k_bs; a.v_hs = a.k_hs; a.v_ss = 32;
        a.o_bs = T * 300; a.o_ts = 300; a.scale = 0.18f; a.causal = causal;
        int rc = ick_attention(&a, st);
        if (rc) { printf("%s rc=%d\n", name, rc); return; }
        float us = timeit([&] { ick_attention(&a, st); }, iters);
        printf("%-28s B=%3d T=%3d S=%3d       : %8.2f us\n", name, Bn, T, S, us);
    };
    attn("cross attention", 64, 20, 216, 0);
    attn("self attention", 64, 20, 20, 1);
    attn("decode cross", 64, 1, 216, 0);
    attn("fact self-attention", 64, 51, 51, 0);
    attn("cross attention S=267", 64, 20, 267, 0);
    float us = timeit([&] { ick_add_layernorm(A, B, bias, bias, C, 1280, 300, 1e-5f, 300, 300, 300, nullptr, nullptr, 0.f, 0, 0, nullptr, st); }, iters);
    printf("%-28s rows=1280               : %8.2f us\n", "add_layernorm", us);
    // row-resident chains (rowchain.hip): GEMM + add & norm (+ GEMM) in one launch
    auto chain = [&](const char* name, int M, int K1, int N2, int hs, int slim = 0) {
        ick_rowchain_args a; memset(&a, 0, sizeof(a));
        a.A = A; a.a_rs = K1; a.M = M; a.K1 = K1; a.d = 300; a.w1p = B; a.b1 = bias; a.res = A + 4000000; a.res_rs = 300;
        a.gamma = bias; a.beta = bias; a.eps = 1e-5f; a.o = C; a.o_rs = 300; a.x = C + 1000000; a.x_rs = 300;
        a.mean = bias + 4096; a.rstd = bias + 8192;
        if (N2) {
            a.w2p = B + 1000000; a.b2 = bias; a.N2 = N2; a.y2 = C + 2000000; a.y2_rs = N2;
            if (hs) { a.hs_dh = 30; a.hs_dhp = 32; a.hs_H = 10; a.hs_S = hs; a.y2_grp = hs; a.y2_gs = (int64_t)(N2 / 300) * 10 * hs * 32; }
        }
        if (slim) a.flags |= ICK_CHAIN_SLIM;
        int rc = ick_rowchain_fwd(&a, st);
        if (rc) { printf("%s rc=%d\n", name, rc); return; }
        float t = timeit([&] { ick_rowchain_fwd(&a, st); }, iters);
        printf("%-28s M=%5d K1=%4d N2=%4d  : %8.2f us\n", name, M, K1, N2, t);
    };
    chain("chain out+LN", 1280, 300, 0, 0);
    chain("chain ffn2+LN", 1280, 512, 0, 0);
    chain("chain out+LN+q", 1280, 300, 300, 20);
    chain("chain out+LN+ffn1", 1280, 300, 512, 0);
    chain("chain ffn2+LN+in_proj", 1280, 512, 900, 20);
    chain("chain out+LN+q B=16", 320, 300, 300, 20);
    chain("chain out+LN+ffn1 slim", 1280, 300, 512, 0, 1);
    chain("chain ffn2+LN+in_proj slim", 1280, 512, 900, 20, 1);
    chain("chain ffn2+LN slim", 1280, 512, 0, 0, 1);
    // backward probes
    float *mean = bias + 4096, *rstd = bias + 8192;
    us = timeit([&] { ick_layernorm_bwd(A, B, C, bias, mean, rstd, C + 4000000, bias + 1024, bias + 2048, 1280, 300, nullptr, 0.f, 0, 0, nullptr, nullptr, st); }, iters);
    printf("%-28s rows=1280               : %8.2f us\n", "layernorm_bwd", us);
    us = timeit([&] { ick_layernorm_bwd(A, B, C, bias, mean, rstd, C + 4000000, bias + 1024, bias + 2048, 1280, 300, nullptr, 0.f, 0, 0, nullptr, C + 8000000, st); }, iters);
    printf("%-28s rows=1280               : %8.2f us\n", "layernorm_bwd (partials)", us);
    us = timeit([&] { ick_colsum(A, 1280, 900, 900, bias + 1024, st); }, iters);
    printf("%-28s 1280x900                : %8.2f us\n", "colsum", us);
    auto attn_bwd = [&](const char* name, int Bn, int T, int S, int causal) {
        ick_attn_bwd_args a; memset(&a, 0, sizeof(a));
        a.Q = Q; a.K = KV; a.V = KV + (size_t)10 * S * 32; a.O = O; a.dO = O + 2000000; a.lse = bias;
        a.dQ = O + 4000000; a.dK = O + 6000000; a.dV = O + 12000000;
        a.B = Bn; a.H = 10; a.T = T; a.S = S; a.dh = 30;
        a.q_bs = (int64_t)10 * T * 32; a.q_hs = T * 32; a.q_ts = 32;
        a.k_bs = (int64_t)6 * 10 * S * 32; a.k_hs = S * 32; a.k_ss = 32; a.v_bs = a.k_bs; a.v_hs = a.k_hs; a.v_ss = 32;
        a.o_bs = T * 300; a.o_ts = 300; a.dq_bs = T * 300; a.dq_ts = 300; a.dk_bs = S * 300; a.dk_ss = 300; a.dv_bs = S * 300; a.dv_ss = 300;
        a.scale = 0.18f; a.causal = causal;
        int rc = ick_attention_bwd(&a, st);
        if (rc) { printf("%s rc=%d\n", name, rc); return; }
        float t = timeit([&] { ick_attention_bwd(&a, st); }, iters);
        printf("%-28s B=%3d T=%3d S=%3d       : %8.2f us\n", name, Bn, T, S, t);
    };
    attn_bwd("cross attention bwd", 64, 20, 216, 0);
    attn_bwd("self attention bwd", 64, 20, 20, 1);
    attn_bwd("fact self-attention bwd", 64, 51, 51, 0);
    attn_bwd("cross attention bwd S=267", 64, 20, 267, 0);
    return 0;
}
