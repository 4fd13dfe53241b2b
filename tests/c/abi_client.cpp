// Stand-alone client of the C ABI (include/dldkd_hip.h): no Python, no torch.  What a C/C++ host (or a cgo/JNI/ctypes
// stub) would do: allocate device buffers with the HIP runtime, pack, score, finish, copy back; then a one-rank RCCL
// communicator through the same ABI (dldkd_comm_*).
// The check is a scalar loop in this file over the SAME bf16-rounded, L2-normalised operands (fp64 accumulation):
//   fused[q][v] = 0.7 * max_{l < len_v} <q0, g0[v][l]> + 0.3 * max_l <q1, g1[v][l]>       (model.py:318-327, eval.py:254)
// Build:  hipcc --offload-arch=gfx950 -I include tests/c/abi_client.cpp -L dl-dkd_amd/dldkd_amd -ldldkd_hip -o tests/c/abi_client
// Exit code 0 and "abi_client ok" on success.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "dldkd_hip.h"

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define ABICHK(x) do { int r_ = (x); if (r_ != 0) { std::fprintf(stderr, "%s -> %d: %s\n", #x, r_, dldkd_last_error()); return 3; } } while (0)

static float bf16_round(float x) {   // round-to-nearest-even to bf16, back to float
    uint32_t u;
    std::memcpy(&u, &x, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    u &= 0xFFFF0000u;
    std::memcpy(&x, &u, 4);
    return x;
}
static uint32_t rng_state = 12345u;
static float rnd() {   // xorshift -> roughly N(0,1) by summing uniforms
    float s = 0.f;
    for (int i = 0; i < 4; ++i) {
        rng_state ^= rng_state << 13; rng_state ^= rng_state >> 17; rng_state ^= rng_state << 5;
        s += (rng_state >> 8) * (1.0f / 16777216.0f);
    }
    return (s - 2.0f) * 1.7320508f;
}

int main(int argc, char** argv) {
    const bool with_comm = !(argc > 1 && std::strcmp(argv[1], "nocomm") == 0);
    const int NQ = 45, NV = 19, L = 40, D = DLDKD_HIDDEN, NB = 2;
    if (dldkd_abi_version() <= 0) { std::fprintf(stderr, "bad abi version\n"); return 1; }
    std::vector<float> q[2], g[2], mask((size_t)NV * L, 0.f);
    std::vector<int> lens(NV);
    for (int v = 0; v < NV; ++v) {
        lens[v] = v == 3 ? L : 1 + (int)((v * 7 + 5) % L);
        for (int l = 0; l < lens[v]; ++l) mask[(size_t)v * L + l] = 1.f;
    }
    for (int b = 0; b < NB; ++b) {
        q[b].resize((size_t)NQ * D);
        g[b].resize((size_t)NV * L * D);
        for (auto& x : q[b]) x = rnd();
        for (auto& x : g[b]) x = rnd();
    }
    // host expectation on normalised + bf16-rounded operands
    auto normalise = [&](const float* src, std::vector<float>& dst) {
        double ss = 0;
        for (int k = 0; k < D; ++k) ss += (double)src[k] * src[k];
        const float inv = 1.0f / std::fmax(std::sqrt((float)ss), 1e-12f);
        dst.resize(D);
        for (int k = 0; k < D; ++k) dst[k] = bf16_round(src[k] * inv);
    };
    std::vector<double> expect((size_t)NQ * NV, 0.0);
    const double w[2] = {0.7, 0.3};
    for (int b = 0; b < NB; ++b) {
        std::vector<std::vector<float>> qn(NQ), gn((size_t)NV * L);
        for (int i = 0; i < NQ; ++i) normalise(&q[b][(size_t)i * D], qn[i]);
        for (int v = 0; v < NV; ++v)
            for (int l = 0; l < lens[v]; ++l) normalise(&g[b][((size_t)v * L + l) * D], gn[(size_t)v * L + l]);
        for (int i = 0; i < NQ; ++i)
            for (int v = 0; v < NV; ++v) {
                double best = -1e30;
                for (int l = 0; l < lens[v]; ++l) {
                    double s = 0;
                    const auto& gg = gn[(size_t)v * L + l];
                    for (int k = 0; k < D; ++k) s += (double)qn[i][k] * gg[k];
                    best = s > best ? s : best;
                }
                expect[(size_t)i * NV + v] += w[b] * best;
            }
    }
    // device side through the C ABI
    float *dq[2], *dg[2], *dmask, *dfused;
    void *pq[2], *pg[2], *ws;
    int32_t *dlens, *dorder, *dinv;
    HIPCHK(hipMalloc(&dmask, mask.size() * 4));
    HIPCHK(hipMemcpy(dmask, mask.data(), mask.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc(&dlens, NV * 4));
    HIPCHK(hipMalloc(&dorder, NV * 4));
    HIPCHK(hipMalloc(&dinv, NV * 4));
    HIPCHK(hipMalloc(&dfused, (size_t)NQ * NV * 4));
    for (int b = 0; b < NB; ++b) {
        HIPCHK(hipMalloc(&dq[b], q[b].size() * 4));
        HIPCHK(hipMalloc(&dg[b], g[b].size() * 4));
        HIPCHK(hipMemcpy(dq[b], q[b].data(), q[b].size() * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(dg[b], g[b].data(), g[b].size() * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMalloc(&pq[b], dldkd_packed_queries_bytes(NQ)));
        HIPCHK(hipMalloc(&pg[b], dldkd_packed_gallery_bytes(NV, L)));
        ABICHK(dldkd_pack_queries_bf16(dq[b], NQ, 1, pq[b], nullptr, nullptr));
        // the streaming packer, two chunks with different padded lengths (the eval driver's usage)
        const int v_split = 4, l0 = L, l1 = L;
        ABICHK(dldkd_pack_gallery_chunk_bf16(dg[b], dmask, v_split, l0, 1, pg[b], dlens, 0, NV, L, nullptr));
        ABICHK(dldkd_pack_gallery_chunk_bf16(dg[b] + (size_t)v_split * L * D, dmask + (size_t)v_split * L, NV - v_split, l1, 1, pg[b],
                                             dlens, v_split, NV, L, nullptr));
    }
    std::vector<int32_t> hl(NV), order(NV), inv(NV);
    HIPCHK(hipMemcpy(hl.data(), dlens, NV * 4, hipMemcpyDeviceToHost));
    for (int v = 0; v < NV; ++v) {
        if (hl[v] != lens[v]) { std::fprintf(stderr, "lens[%d] = %d, expected %d\n", v, hl[v], lens[v]); return 4; }
        order[v] = NV - 1 - v;   // any permutation is a valid visiting order
        inv[order[v]] = v;
    }
    HIPCHK(hipMemcpy(dorder, order.data(), NV * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dinv, inv.data(), NV * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc(&ws, dldkd_simpool_eval_workspace_bytes(NQ, NV, NB)));
    ABICHK(dldkd_simpool_eval_bf16(pq, pg, dlens, dorder, NQ, NV, L, NB, 0, nullptr, ws, nullptr));
    ABICHK(dldkd_simpool_finish(ws, dinv, NQ, NV, NB, 0.7f, 0.3f, dfused, nullptr, nullptr, nullptr));
    HIPCHK(hipDeviceSynchronize());
    std::vector<float> fused((size_t)NQ * NV);
    HIPCHK(hipMemcpy(fused.data(), dfused, fused.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (size_t i = 0; i < fused.size(); ++i) worst = std::fmax(worst, std::fabs(fused[i] - expect[i]));
    // error paths: sizes are validated, the message is retrievable
    if (dldkd_simpool_eval_bf16(pq, pg, dlens, dorder, NQ, NV, DLDKD_MAX_CLIPS + 1, NB, 0, nullptr, ws, nullptr) == 0 ||
        std::strlen(dldkd_last_error()) == 0) { std::fprintf(stderr, "bad-size call was accepted\n"); return 5; }
    if (worst > 2e-5) { std::fprintf(stderr, "max |HIP - host| = %g\n", worst); return 6; }
    // Collectives (one rank): the library finds RCCL by itself (no torch in this process: the system librccl.so.1), a communicator is
    // an explicit object, every collective is one enqueue on the caller's stream.  One rank: sum / gather / broadcast are identities.
    if (with_comm) {
        std::fprintf(stderr, "comm: loading RCCL\n");
        if (dldkd_comm_rccl_version() < 20000) { std::fprintf(stderr, "rccl version %d: %s\n", dldkd_comm_rccl_version(), dldkd_last_error()); return 7; }
        unsigned char id[DLDKD_COMM_ID_BYTES];
        void* comm = nullptr;
        std::fprintf(stderr, "comm: version %d, unique id\n", dldkd_comm_rccl_version());
        ABICHK(dldkd_comm_unique_id(id));
        std::fprintf(stderr, "comm: init\n");
        ABICHK(dldkd_comm_init(&comm, 1, 0, id));
        std::fprintf(stderr, "comm: collectives\n");
        int world = -1, rank = -1;
        ABICHK(dldkd_comm_info(comm, &world, &rank));
        if (world != 1 || rank != 0) { std::fprintf(stderr, "comm_info: %d of %d\n", rank, world); return 7; }
        hipStream_t st;
        HIPCHK(hipStreamCreate(&st));
        float *dgather = nullptr;
        HIPCHK(hipMalloc(&dgather, fused.size() * 4));
        ABICHK(dldkd_comm_all_reduce(comm, dfused, dfused, fused.size(), DLDKD_F32, DLDKD_SUM, st));        // in place, like the gradient buffer
        ABICHK(dldkd_comm_all_gather(comm, dfused, dgather, fused.size(), DLDKD_F32, st));                  // like a block of scores
        ABICHK(dldkd_comm_broadcast(comm, dgather, fused.size(), DLDKD_F32, 0, st));
        HIPCHK(hipStreamSynchronize(st));
        ABICHK(dldkd_comm_async_error(comm));
        std::vector<float> back(fused.size());
        HIPCHK(hipMemcpy(back.data(), dgather, back.size() * 4, hipMemcpyDeviceToHost));
        if (std::memcmp(back.data(), fused.data(), back.size() * 4) != 0) { std::fprintf(stderr, "one-rank collectives changed the data\n"); return 7; }
        if (dldkd_comm_all_reduce(comm, dfused, dfused, 4, 99, DLDKD_SUM, st) != DLDKD_EINVAL) { std::fprintf(stderr, "bad dtype accepted\n"); return 7; }
        std::fprintf(stderr, "comm: destroy\n");
        ABICHK(dldkd_comm_destroy(comm));
        std::fprintf(stderr, "comm: done\n");
        HIPCHK(hipStreamDestroy(st));
        HIPCHK(hipFree(dgather));
    }
    std::printf("abi_client ok: %d x %d x <=%d clips, 2 branches, max |HIP - host scalar loop| = %.2e\n", NQ, NV, L, worst);
    return 0;
}
