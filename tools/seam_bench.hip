// Micro-benchmark of an in-launch hand-off between the workgroups of a cluster (the seam of a persistent decode step).
//
// 256 workgroups of 512 threads, clusters of 16.  Every phase each workgroup stores a 24 KiB slice tagged with the
// phase, arrives on its cluster's counter, waits until all 16 have arrived and reads 48 KiB of the slices back,
// checking every word.  Variants: how the payload is stored (plain + release fence / write-through 8-byte / write-through
// 16-byte), how it is read (acquire fence + plain loads / write-through-aware loads with no fence), and whether a
// cluster's workgroups share blockIdx % 8.  Every spin is bounded; a timeout or a stale word is reported, never waited out.
//
//   hipcc --offload-arch=gfx950 -O3 -o seam_bench tools/seam_bench.hip && ./seam_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x)                                                                              \
    do {                                                                                      \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess) {                                                               \
            fprintf(stderr, "%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            exit(2);                                                                          \
        }                                                                                     \
    } while (0)

constexpr int CL = 16;               // workgroups per cluster
constexpr int NT = 512;              // threads per workgroup
constexpr int SLICE_V4 = 3 * NT;     // uint4 per workgroup slice (24 KiB)
constexpr int READ_V4 = 6;           // uint4 per thread read back (48 KiB per workgroup)

struct Args {
    uint4* buf;        // [2][clusters][CL][SLICE_V4]
    unsigned* cnt;     // [clusters] arrival counters, zeroed before the launch
    unsigned* err;     // [0] stale words, [1] timeouts
    unsigned* flags;   // [clusters][CL] per-producer flags, 64 bytes apart
    unsigned* go;      // [clusters] release words, 256 bytes apart
    int phases, st_mode, ld_mode, same_xcd, payload, bar_mode;
};

__device__ __forceinline__ void store_v4(uint4* p, uint4 v, int mode, __amdgpu_buffer_rsrc_t rsrc, unsigned off) {
    if (mode == 0) {
        *p = v;
    } else if (mode == 1) {
        unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
        __hip_atomic_store(q, ((unsigned long long)v.y << 32) | v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(q + 1, ((unsigned long long)v.w << 32) | v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        __builtin_amdgcn_raw_buffer_store_b128(v4u{v.x, v.y, v.z, v.w}, rsrc, off, 0, 16);
    }
}

__device__ __forceinline__ uint4 load_v4(const uint4* p, int mode, __amdgpu_buffer_rsrc_t rsrc, unsigned off) {
    if (mode == 0) return *p;
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const v4u r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 16);
    return make_uint4(r[0], r[1], r[2], r[3]);
}

__global__ __launch_bounds__(NT) void seam_kernel(const Args a) {
    __shared__ int s_abort;
    const int tid = threadIdx.x, b = blockIdx.x;
    int cluster, j;
    if (a.same_xcd) {
        cluster = (b & 7) * (gridDim.x / 8 / CL) + (b >> 3) / CL;
        j = (b >> 3) % CL;
    } else {
        cluster = b / CL;
        j = b % CL;
    }
    const int clusters = gridDim.x / CL;
    if (tid == 0) s_abort = 0;
    __syncthreads();
    const size_t total_bytes = (size_t)2 * clusters * CL * SLICE_V4 * sizeof(uint4);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.buf, 0, (int)total_bytes, 0x00020000);
    unsigned bad = 0;
    for (int ph = 0; ph < a.phases; ++ph) {
        uint4* mine = a.buf + (((size_t)(ph & 1) * clusters + cluster) * CL + j) * SLICE_V4;
        const unsigned tag = (unsigned)(ph + 1);
        if (a.payload) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                uint4* p = mine + tid + i * NT;
                store_v4(p, make_uint4(tag, (unsigned)j, (unsigned)(tid + i * NT), tag ^ 0x5a5a5a5au), a.st_mode, rsrc,
                         (unsigned)((const char*)p - (const char*)a.buf));
            }
        }
        // ---- arrive ----
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (a.bar_mode == 1) {
            // per-producer flags: one write-through store each, the first 16 lanes of the workgroup poll one flag each
            if (tid == 0) {
                if (a.st_mode == 0 && a.payload) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __hip_atomic_store(a.flags + ((size_t)cluster * CL + j) * 16, (unsigned)(ph + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (tid < 64) {
                const unsigned want = (unsigned)(ph + 1);
                unsigned spins = 0;
                for (;;) {
                    const unsigned got = tid < CL ? __hip_atomic_load(a.flags + ((size_t)cluster * CL + tid) * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : want;
                    if (__all(got >= want)) break;
                    if (++spins > (1u << 22)) {
                        if (tid == 0) { atomicAdd(a.err + 1, 1u); s_abort = 1; }
                        break;
                    }
                }
                if (tid == 0 && a.ld_mode == 0 && a.payload) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            }
        } else if (tid == 0) {
            if (a.st_mode == 0 && a.payload) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            const unsigned tag = (unsigned)(ph + 1);
            const unsigned want = (unsigned)CL * tag;
            unsigned spins = 0;
            if (a.bar_mode == 3) {
                // returning add: the last arriver raises a release word on a line of its own, the others poll that word
                const unsigned old = __hip_atomic_fetch_add(a.cnt + cluster * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (old + 1 == want) {
                    __hip_atomic_store(a.go + cluster * 64, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    while (__hip_atomic_load(a.go + cluster * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < tag) {
                        if (++spins > (1u << 22)) { atomicAdd(a.err + 1, 1u); s_abort = 1; break; }
                    }
                }
            } else {
                __hip_atomic_fetch_add(a.cnt + cluster * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(a.cnt + cluster * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    if (a.bar_mode == 0) __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1u << 22)) { atomicAdd(a.err + 1, 1u); s_abort = 1; break; }
                }
            }
            if (a.ld_mode == 0 && a.payload) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        __syncthreads();
        if (s_abort) break;
        if (a.payload) {
            // ---- read 3 KiB of each producer's slice and check every word ----
            const uint4* base = a.buf + ((size_t)(ph & 1) * clusters + cluster) * CL * SLICE_V4;
            uint4 v[READ_V4];
#pragma unroll
            for (int i = 0; i < READ_V4; ++i) {
                const int idx = tid + i * NT;           // 0 .. 3071
                const int prod = idx / 192, w = idx % 192;
                const uint4* p = base + (size_t)prod * SLICE_V4 + w;
                v[i] = load_v4(p, a.ld_mode, rsrc, (unsigned)((const char*)p - (const char*)a.buf));
            }
#pragma unroll
            for (int i = 0; i < READ_V4; ++i) {
                const int idx = tid + i * NT;
                const int prod = idx / 192, w = idx % 192;
                bad += v[i].x != tag || v[i].y != (unsigned)prod || v[i].z != (unsigned)w || v[i].w != (tag ^ 0x5a5a5a5au);
            }
        }
    }
    if (bad) atomicAdd(a.err, bad);
}

int main(int argc, char** argv) {
    const int phases = argc > 1 ? atoi(argv[1]) : 400;
    const int grid = 256;
    const int clusters = grid / CL;
    Args a{};
    CHECK(hipMalloc(&a.buf, (size_t)2 * clusters * CL * SLICE_V4 * sizeof(uint4)));
    CHECK(hipMalloc(&a.cnt, 4096));
    CHECK(hipMalloc(&a.flags, 65536));
    CHECK(hipMalloc(&a.go, 65536));
    CHECK(hipMalloc(&a.err, 256));
    CHECK(hipMemset(a.buf, 0, (size_t)2 * clusters * CL * SLICE_V4 * sizeof(uint4)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    a.phases = phases;
    const char* st_names[] = {"plain+release", "sc1 8-byte", "sc1 16-byte"};
    const char* ld_names[] = {"acquire+plain", "sc1 loads"};
    const char* bar_names[] = {"counter+sleep", "flags", "counter", "ticket+go"};
    for (int bar = 0; bar < 4; ++bar)
    for (int payload = 0; payload <= 1; ++payload)
        for (int same = 1; same <= 1; ++same)
            for (int st = 0; st < 3; ++st)
                for (int ld = 0; ld < 2; ++ld) {
                    if (!payload && (st || ld)) continue;
                    if (st == 0 && ld == 1) continue;  // plain stores are not a write-through hand-off
                    a.st_mode = st; a.ld_mode = ld; a.same_xcd = same; a.payload = payload; a.bar_mode = bar;
                    if (payload && st == 0) continue;
                    float best = 1e30f;
                    unsigned err[2] = {0, 0};
                    for (int rep = 0; rep < 5; ++rep) {
                        CHECK(hipMemsetAsync(a.cnt, 0, 4096, 0));
                        CHECK(hipMemsetAsync(a.flags, 0, 65536, 0));
                        CHECK(hipMemsetAsync(a.go, 0, 65536, 0));
                        CHECK(hipMemsetAsync(a.err, 0, 256, 0));
                        CHECK(hipEventRecord(e0, 0));
                        hipLaunchKernelGGL(seam_kernel, dim3(grid), dim3(NT), 0, 0, a);
                        CHECK(hipGetLastError());
                        CHECK(hipEventRecord(e1, 0));
                        CHECK(hipEventSynchronize(e1));
                        float ms = 0.f;
                        CHECK(hipEventElapsedTime(&ms, e0, e1));
                        unsigned e[2];
                        CHECK(hipMemcpy(e, a.err, 8, hipMemcpyDeviceToHost));
                        err[0] += e[0]; err[1] += e[1];
                        if (ms < best) best = ms;
                        if (e[1]) break;
                    }
                    printf("bar=%-13s payload=%d same_xcd=%d store=%-14s load=%-14s  %.2f us/phase  stale=%u timeouts=%u\n", bar_names[bar], payload, same,
                           st_names[st], ld_names[ld], best * 1000.f / phases, err[0], err[1]);
                    fflush(stdout);
                    if (err[1]) { printf("timeout: stopping\n"); return 1; }
                }
    return 0;
}
