// ref_misc_harness.cpp -- TEST ONLY.  Small facts taken straight from the reference's headers, compiled where they lie:
// the gpuserver shared-memory layout (M/src/commons/GpuUtil.h:9-49, M/lib/libmarv/src/marv.h) and Util::hash
// (M/src/commons/Util.h:368-377) that names the block.  Nothing here is linked into the product.
#include <cstddef>
#include <cstdint>
#include "GpuUtil.h"
#include "Util.h"

extern "C" {

void ref_gpu_shm_layout(unsigned int out[11]) {
    out[0] = (unsigned int) sizeof(GPUSharedMemory);
    out[1] = (unsigned int) offsetof(GPUSharedMemory, maxSeqLen); out[2] = (unsigned int) offsetof(GPUSharedMemory, maxResListLen);
    out[3] = (unsigned int) offsetof(GPUSharedMemory, state); out[4] = (unsigned int) offsetof(GPUSharedMemory, serverExit);
    out[5] = (unsigned int) offsetof(GPUSharedMemory, queryOffset); out[6] = (unsigned int) offsetof(GPUSharedMemory, queryLen);
    out[7] = (unsigned int) offsetof(GPUSharedMemory, resultsOffset); out[8] = (unsigned int) offsetof(GPUSharedMemory, resultLen);
    out[9] = (unsigned int) offsetof(GPUSharedMemory, profileOffset); out[10] = (unsigned int) sizeof(Marv::Result);
}

size_t ref_gpu_shm_bytes(unsigned int maxSeqLen, unsigned int maxResListLen) { return GPUSharedMemory::calculateSize(maxSeqLen, maxResListLen); }

int ref_gpu_shm_states(int out[4]) {
    out[0] = GPUSharedMemory::IDLE; out[1] = GPUSharedMemory::RESERVED; out[2] = GPUSharedMemory::READY; out[3] = GPUSharedMemory::DONE;
    return 4;
}

size_t ref_util_hash(const char *s, size_t n) { return Util::hash(s, n); }

} // extern "C"
