// Dev probe: can two PROCESSES on one device exchange data through hipIpc-mapped memory with in-kernel flags?
// (The transport csrc/comm.inc builds on; run through gpurun.)  Two children (fork before any HIP call), each
// exports one fine-grained buffer [flags | payload]; ping-pong: the sender's kernel stores payload + flag into the
// PEER's buffer, the receiver's kernel polls its own flag (bounded), checks the payload, answers.
//   hipcc --offload-arch=gfx950 -O2 tools/ipc_probe.cpp -o tools/ipc_probe
#include <hip/hip_runtime.h>
#include <sys/socket.h>
#include <sys/wait.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "[%d] %s -> %s (line %d)\n", me, #x, hipGetErrorString(e_), __LINE__); _exit(3); } } while (0)

constexpr int PAYLOAD = 4096;  // doubles
struct Box { unsigned long long flag[8]; unsigned long long err; double data[PAYLOAD]; };

__global__ void k_push(Box *peer, unsigned long long seq, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) peer->data[i] = (double)seq + i;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&peer->flag[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_wait_check(Box *mine, unsigned long long seq, int n, long long max_cycles) {
    __shared__ int ok;
    if (threadIdx.x == 0) {
        const long long t0 = wall_clock64();
        ok = 0;
        while (wall_clock64() - t0 < max_cycles) {
            if (__hip_atomic_load(&mine->flag[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) >= seq) { ok = 1; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (!ok) atomicAdd(&mine->err, 1ull << 32);
    }
    __syncthreads();
    __threadfence_system();
    if (ok) {
        int bad = 0;
        for (int i = threadIdx.x; i < n; i += blockDim.x) bad += (mine->data[i] != (double)seq + i);
        if (bad) atomicAdd(&mine->err, (unsigned long long)bad);
    }
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    const int finegrained = argc > 2 ? atoi(argv[2]) : 1;
    int sv[2];
    socketpair(AF_UNIX, SOCK_STREAM, 0, sv);
    int me = 0;
    pid_t pid = fork();
    me = pid == 0 ? 1 : 0;
    const int fd = sv[me];
    CK(hipSetDevice(0));
    Box *mine = nullptr, *peer = nullptr;
    if (finegrained) CK(hipExtMallocWithFlags((void **)&mine, sizeof(Box), hipDeviceMallocFinegrained));
    else CK(hipMalloc((void **)&mine, sizeof(Box)));
    CK(hipMemset(mine, 0, sizeof(Box)));
    CK(hipDeviceSynchronize());
    hipIpcMemHandle_t hm, hp;
    CK(hipIpcGetMemHandle(&hm, mine));
    if (write(fd, &hm, sizeof(hm)) != sizeof(hm) || read(fd, &hp, sizeof(hp)) != sizeof(hp)) { fprintf(stderr, "pipe\n"); _exit(4); }
    CK(hipIpcOpenMemHandle((void **)&peer, hp, hipIpcMemLazyEnablePeerAccess));
    char c = 'r';
    if (write(fd, &c, 1) != 1 || read(fd, &c, 1) != 1) _exit(4);
    hipStream_t s;
    CK(hipStreamCreate(&s));
    const long long max_cycles = 100000000LL * 5;  // wall_clock64 ticks at 100 MHz: 5 s
    auto t0 = std::chrono::steady_clock::now();
    // even sequence numbers: 0 -> 1, odd: 1 -> 0; everything queued up front on one stream per process
    for (int k = 1; k <= iters; ++k) {
        const unsigned long long seq = k;
        const bool sender = (k & 1) == (me == 0 ? 1 : 0);
        if (sender) hipLaunchKernelGGL(k_push, dim3(1), dim3(256), 0, s, peer, seq, PAYLOAD);
        else hipLaunchKernelGGL(k_wait_check, dim3(1), dim3(256), 0, s, mine, seq, PAYLOAD, max_cycles);
    }
    CK(hipStreamSynchronize(s));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    Box h;
    CK(hipMemcpy(&h, mine, sizeof(Box), hipMemcpyDeviceToHost));
    printf("[%d] finegrained=%d iters=%d  %.2f us per one-way hand-off  timeouts=%llu  bad values=%llu\n", me, finegrained, iters,
           us / iters, h.err >> 32, h.err & 0xffffffffull);
    fflush(stdout);
    if (write(fd, &c, 1) != 1 || read(fd, &c, 1) != 1) _exit(4);
    CK(hipIpcCloseMemHandle(peer));
    CK(hipFree(mine));
    if (me == 0) { int st; waitpid(pid, &st, 0); }
    return 0;
}
