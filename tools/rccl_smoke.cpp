// rccl_smoke.cpp -- (tools) single-process RCCL exactly as fsgpu_db_broadcast drives it: dlopen, ncclCommInitAll over
// the visible devices, one grouped in-place ncclBroadcast of a byte buffer per device, verify, destroy.  On a 1-GPU box
// this checks the library loading and call signatures; on a multi-GPU node it is the real xGMI broadcast.
// build: hipcc -O2 -o rccl_smoke rccl_smoke.cpp -ldl
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
int main() {
    void *lib = nullptr;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) if ((lib = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
    if (!lib) { printf("librccl not loadable: %s\n", dlerror()); return 2; }
    auto CommInitAll = (int (*)(void **, int, const int *)) dlsym(lib, "ncclCommInitAll");
    auto CommDestroy = (int (*)(void *)) dlsym(lib, "ncclCommDestroy");
    auto GroupStart = (int (*)()) dlsym(lib, "ncclGroupStart");
    auto GroupEnd = (int (*)()) dlsym(lib, "ncclGroupEnd");
    auto Broadcast = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t)) dlsym(lib, "ncclBroadcast");
    if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Broadcast) { printf("missing symbol\n"); return 3; }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) { printf("no device\n"); return 4; }
    std::vector<int> devs(n);
    for (int i = 0; i < n; i++) devs[i] = i;
    std::vector<void *> comms(n, nullptr);
    int rc = CommInitAll(comms.data(), n, devs.data());
    if (rc) { printf("ncclCommInitAll rc=%d\n", rc); return 5; }
    const size_t bytes = 64u << 20;
    std::vector<void *> buf(n);
    std::vector<hipStream_t> st(n);
    std::vector<unsigned char> h(bytes);
    for (size_t i = 0; i < bytes; i++) h[i] = (unsigned char) (i * 2654435761u >> 24);
    for (int i = 0; i < n; i++) {
        hipSetDevice(i); hipStreamCreate(&st[i]); hipMalloc(&buf[i], bytes);
        if (i == 0) hipMemcpy(buf[0], h.data(), bytes, hipMemcpyHostToDevice); else hipMemset(buf[i], 0, bytes);
    }
    hipEvent_t e0, e1; hipSetDevice(0); hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, st[0]);
    rc = GroupStart();
    for (int i = 0; i < n && !rc; i++) { hipSetDevice(i); rc = Broadcast(buf[0], buf[i], bytes, 1 /*ncclUint8*/, 0, comms[i], st[i]); }
    rc = GroupEnd() || rc;
    hipSetDevice(0); hipEventRecord(e1, st[0]);
    for (int i = 0; i < n; i++) { hipSetDevice(i); hipStreamSynchronize(st[i]); }
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    if (rc) { printf("broadcast rc=%d\n", rc); return 6; }
    std::vector<unsigned char> back(bytes);
    for (int i = 0; i < n; i++) {
        hipSetDevice(i); hipMemcpy(back.data(), buf[i], bytes, hipMemcpyDeviceToHost);
        if (memcmp(back.data(), h.data(), bytes)) { printf("device %d: data mismatch\n", i); return 7; }
    }
    for (void *c : comms) CommDestroy(c);
    printf("rccl smoke ok: %d device(s), 64 MiB broadcast in %.3f ms\n", n, ms);
    return 0;
}
