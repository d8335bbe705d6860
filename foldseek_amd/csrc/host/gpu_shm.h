// gpu_shm.h -- the shared-memory block a resident `gpuserver` and its clients talk through.  Byte layout and state
// machine of the reference: GPUSharedMemory, M/src/commons/GpuUtil.h:9-49 and GpuUtil.cpp:18-121 (Marv::Result:
// M/lib/libmarv/src/marv.h:10-15), so that either side can be the reference's binary.
#pragma once
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <string>

namespace fsh {

struct GpuShmResult {            // Marv::Result
    unsigned int id;
    int score;
    int qEndPos;
    int dbEndPos;
};

struct GpuShm {
    enum State { IDLE, RESERVED, READY, DONE };
    unsigned int maxSeqLen;                   // longest query the block can carry
    unsigned int maxResListLen;               // capacity of the result list
    std::atomic<int> state;
    std::atomic<bool> serverExit;
    unsigned int queryOffset;                 // query residue codes   [maxSeqLen]
    unsigned int queryLen;
    unsigned int resultsOffset;               // GpuShmResult          [maxResListLen]
    unsigned int resultLen;
    unsigned int profileOffset;               // int8 profile          [21][queryLen], row-major by residue code

    int8_t *query() { return reinterpret_cast<int8_t *>(this) + queryOffset; }
    GpuShmResult *results() { return reinterpret_cast<GpuShmResult *>(reinterpret_cast<char *>(this) + resultsOffset); }
    int8_t *profile() { return reinterpret_cast<int8_t *>(this) + profileOffset; }
    static size_t bytes(unsigned int maxSeqLen, unsigned int maxResListLen) {
        return sizeof(GpuShm) + (size_t) maxSeqLen + sizeof(GpuShmResult) * (size_t) maxResListLen + (size_t) 21 * maxSeqLen;
    }
};
static_assert(sizeof(GpuShmResult) == 16, "Marv::Result layout");
static_assert(sizeof(GpuShm) == 36 && offsetof(GpuShm, state) == 8 && offsetof(GpuShm, serverExit) == 12 &&
              offsetof(GpuShm, queryOffset) == 16 && offsetof(GpuShm, profileOffset) == 32, "GPUSharedMemory layout");

// name of the block = decimal Util::hash (h = 31 h + c) of realpath(db without .idx/.linidx) + visible devices + version
std::string gpuShmName(const std::string &db, const char *visibleDevices, const char *version);
GpuShm *gpuShmCreate(const std::string &name, unsigned int maxSeqLen, unsigned int maxResListLen, std::string &err);
void gpuShmDestroy(GpuShm *shm, const std::string &name);     // unmap + unlink (server)
GpuShm *gpuShmOpen(const std::string &name, std::string &err);  // map an existing block, size taken from its header
void gpuShmUnmap(GpuShm *shm);
bool gpuShmExists(const std::string &name);

} // namespace fsh
