// gpu_shm.cpp -- POSIX shared-memory plumbing of the gpuserver protocol (see gpu_shm.h for the reference citations).
#include "gpu_shm.h"
#include "fshost.h"

#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <new>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace fsh {

static std::string withoutIndexSuffix(const std::string &db) {      // PrefilteringIndexReader::dbPathWithoutIndex
    auto strip = [](const std::string &s, const char *suf) {
        const size_t n = strlen(suf), p = s.rfind(suf);
        return (p != std::string::npos && s.size() - p == n) ? s.substr(0, p) : s;
    };
    return strip(strip(db, ".idx"), ".linidx");
}

std::string gpuShmName(const std::string &db, const char *visibleDevices, const char *version) {
    std::string path = withoutIndexSuffix(db);
    char real[PATH_MAX];
    if (realpath(path.c_str(), real)) path = real;                   // FileUtil::getRealPathFromSymLink
    if (visibleDevices) path.append(visibleDevices);
    if (version) path.append(version);
    size_t h = 0;
    for (char c : path) h = h * 31 + (size_t) c;                     // Util::hash: plain char, sign-extended like the reference
    return std::to_string(h);
}

GpuShm *gpuShmCreate(const std::string &name, unsigned int maxSeqLen, unsigned int maxResListLen, std::string &err) {
    const size_t size = GpuShm::bytes(maxSeqLen, maxResListLen);
    int fd = shm_open(name.c_str(), O_CREAT | O_RDWR, 0666);
    if (fd == -1) { err = "Failed to open shared memory"; return nullptr; }
    if (ftruncate(fd, (off_t) size) == -1) { close(fd); err = "Failed to size shared memory"; return nullptr; }
    void *ptr = mmap(nullptr, size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (ptr == MAP_FAILED) { err = "Failed to map shared memory"; return nullptr; }
    GpuShm *shm = new (ptr) GpuShm;
    shm->state.store(GpuShm::IDLE);
    shm->serverExit.store(false);
    shm->queryLen = 0; shm->resultLen = 0;
    shm->maxSeqLen = maxSeqLen;
    shm->maxResListLen = maxResListLen;
    shm->queryOffset = sizeof(GpuShm);
    shm->resultsOffset = shm->queryOffset + maxSeqLen;
    shm->profileOffset = shm->resultsOffset + (unsigned int) (sizeof(GpuShmResult) * maxResListLen);
    return shm;
}

void gpuShmUnmap(GpuShm *shm) {
    if (shm) munmap(shm, GpuShm::bytes(shm->maxSeqLen, shm->maxResListLen));
}

void gpuShmDestroy(GpuShm *shm, const std::string &name) {
    gpuShmUnmap(shm);
    shm_unlink(name.c_str());
}

GpuShm *gpuShmOpen(const std::string &name, std::string &err) {
    int fd = shm_open(name.c_str(), O_RDWR, 0666);
    if (fd == -1) { err = "Failed to open shared memory"; return nullptr; }
    void *ptr = mmap(nullptr, sizeof(GpuShm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    if (ptr == MAP_FAILED) { close(fd); err = "Failed to map shared memory"; return nullptr; }
    const unsigned int maxSeqLen = reinterpret_cast<unsigned int *>(ptr)[0], maxResListLen = reinterpret_cast<unsigned int *>(ptr)[1];
    munmap(ptr, sizeof(GpuShm));
    const size_t size = GpuShm::bytes(maxSeqLen, maxResListLen);
    struct stat st;
    if (fstat(fd, &st) != 0 || (size_t) st.st_size < size) { close(fd); err = "shared memory block is smaller than its header says"; return nullptr; }
    ptr = mmap(nullptr, size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (ptr == MAP_FAILED) { err = "Failed to remap shared memory"; return nullptr; }
    // the three areas must lie inside the block (a block of another program / version under the same name is refused, not written into)
    GpuShm *shm = reinterpret_cast<GpuShm *>(ptr);
    const auto inside = [&](unsigned int off, size_t len) { return off >= sizeof(GpuShm) && off <= size && len <= size - off; };
    if (!inside(shm->queryOffset, maxSeqLen) || !inside(shm->resultsOffset, sizeof(GpuShmResult) * (size_t) maxResListLen) ||
        !inside(shm->profileOffset, (size_t) 21 * maxSeqLen)) {          // (the result area is NOT aligned in general: it starts at 36 + maxSeqLen)
        munmap(ptr, size);
        err = "shared memory block has an unexpected layout (not a gpuserver block of this database?)";
        return nullptr;
    }
    return shm;
}

bool gpuShmExists(const std::string &name) {
    struct stat st;                                                  // the client polls /dev/shm/<hash> (ungappedprefilter.cpp:74-83)
    return stat(("/dev/shm/" + name).c_str(), &st) == 0 && st.st_size > 0;
}

} // namespace fsh

extern "C" {

int fshost_gpu_shm_name(const char *db, const char *visibleDevices, const char *version, char *out, size_t cap) {
    if (!db || !out || cap == 0) return -1;
    const std::string s = fsh::gpuShmName(db, visibleDevices, version);
    if (s.size() + 1 > cap) return -1;
    memcpy(out, s.c_str(), s.size() + 1);
    return (int) s.size();
}

size_t fshost_gpu_shm_bytes(unsigned int maxSeqLen, unsigned int maxResListLen) { return fsh::GpuShm::bytes(maxSeqLen, maxResListLen); }

void fshost_gpu_shm_layout(unsigned int out[11]) {
    using fsh::GpuShm;
    out[0] = (unsigned int) sizeof(GpuShm);
    out[1] = (unsigned int) offsetof(GpuShm, maxSeqLen); out[2] = (unsigned int) offsetof(GpuShm, maxResListLen);
    out[3] = (unsigned int) offsetof(GpuShm, state); out[4] = (unsigned int) offsetof(GpuShm, serverExit);
    out[5] = (unsigned int) offsetof(GpuShm, queryOffset); out[6] = (unsigned int) offsetof(GpuShm, queryLen);
    out[7] = (unsigned int) offsetof(GpuShm, resultsOffset); out[8] = (unsigned int) offsetof(GpuShm, resultLen);
    out[9] = (unsigned int) offsetof(GpuShm, profileOffset); out[10] = (unsigned int) sizeof(fsh::GpuShmResult);
}

} // extern "C"
