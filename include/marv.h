/*
 * marv.h -- link-compatible `class Marv` over the fsgpu_* C ABI.
 *
 * The reference's in-process GPU plugin for the gapless prefilter IS this class (M/lib/libmarv/src/marv.h:6-58); its call
 * sites are M/src/prefiltering/ungappedprefilter.cpp:139-158,207 (runFilterOnGpu), M/src/util/gpuserver.cpp:58-82 and
 * M/src/commons/GpuUtil.{h,cpp} (sizeof(Marv::Result) in the shared-memory layout).  This header declares the same
 * class -- same member functions, same nested types, same data members in the same order -- so that the reference's own
 * translation units compile against it and link to foldseek_amd/csrc/host/marv_shim.cpp + libfsgpu.so with ZERO source
 * hunks (oracle/build_ref_full.sh gpu builds exactly that binary; tests/test_marv_dropin.py runs it on a GPU).
 *
 * Differences in behaviour, all on the side of the CPU path the hit lists must equal (BASELINE.json north_star):
 *   - scan() returns the uint8-saturated scores of SmithWaterman::ungapped_alignment (capped at 255 - bias,
 *     M/src/alignment/StripedSmithWaterman.cpp:1817-1876); libmarv's half2 kernels do not saturate.
 *   - results are ordered (score desc, id asc), the order hit_t::compareHitsByScoreAndId gives the CPU path.
 *   - AlignmentType GAPLESS only; the other two types (gapped end-position scan) terminate with a message, the way libmarv
 *     terminates on a CUDA error (CUERR).
 *   - like libmarv, one Marv object drives every visible device and shards the TARGETS over them (device k holds targets
 *     k, k + N, ...; per-device top lists merged in the CPU path's order) because its caller hands it one query at a time; the
 *     throughput path shards QUERIES over replicated DBs instead (fsgpu_db_broadcast / fsgpu-modules --gpus).
 */
#ifndef MARV_H
#define MARV_H

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

class Marv {
public:
    enum AlignmentType {
        GAPLESS,
        SMITH_WATERMAN,
        GAPLESS_SMITH_WATERMAN
    };

    Marv(size_t dbEntries, int alphabetSize, int maxSeqLength, size_t maxSeqs, AlignmentType alignmentType = AlignmentType::GAPLESS);
    ~Marv();

    static std::vector<int> getDeviceIds();
    /* data = the padded GPU database file (makepaddedseqdb), offset[dbEntries + 1], length[dbEntries]; returns a handle */
    void* loadDb(char* data, size_t* offset, int32_t* length, size_t dbByteSize);
    void* loadDb(char* data, size_t dbByteSize, void* otherdb);
    void setDb(void* dbhandle);
    void setDbWithAllocation(void* dbhandle, const std::string& allocationinfo);
    std::string getDbMemoryHandle();

    void printInfo();
    void prefetch();

    void startTimer();
    void stopTimer();

    struct Stats {
        size_t results;
        int numOverflows;
        double seconds;
        double gcups;
    };

    struct Result {
        unsigned int id;
        int score;
        int qEndPos;
        int dbEndPos;

        Result(unsigned int id, int score, int qEndPos, int dbEndPos) :
            id(id), score(score), qEndPos(qEndPos), dbEndPos(dbEndPos) {};
    };

    /* sequence: numeric codes [sequenceLength]; pssm: int8 [alphabetSize][sequenceLength] (row a, column i);
     * results: caller-owned, capacity maxSeqs of the constructor */
    Stats scan(const char* sequence, size_t sequenceLength, int8_t* pssm, Result* results);

private:
    size_t dbEntries;
    int alphabetSize;

    void* cudasw;        /* here: the shim's state (device context, staging buffers) */
    // void* db;
    void* dbmanager;     /* here: the database handle that is current */
    AlignmentType alignmentType;
};

#endif
