// mmseqs_db.h -- minimal reader / writer for the MMseqs2 on-disk database layout the hot-path modules exchange
// (reference: DBReader / DBWriter, M/src/commons/DBReader.{h,cpp}, DBWriter.cpp:331-428; layout summary SURVEY.md 8b):
//   <db>         entries back to back, each terminated by '\0' (sequence entries: residues + '\n' + '\0')
//   <db>.index   one line per entry: key \t offset \t length   (length counts the terminator bytes), sorted by key
//   <db>.dbtype  int32: low 16 bits = type (0 amino acids, 5 alignment results, 7 prefilter results),
//                high 16 bits = extended flags (8 = padded GPU database)
// Only what the two modules need: uncompressed, single data file (or <db>.0, <db>.1 ... concatenated by offset).
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

namespace fsh {

enum { DBTYPE_AMINO_ACIDS = 0, DBTYPE_ALIGNMENT_RES = 5, DBTYPE_PREFILTER_RES = 7 };
enum { DBTYPE_EXTENDED_COMPRESSED = 1, DBTYPE_EXTENDED_GPU = 8 };

class DbReader {
public:
    struct Entry { uint32_t key; uint64_t offset; uint32_t length; };
    ~DbReader();
    // `path` may name a precomputed index (<db>.idx, what the workflows pass once `createindex` has run:
    // F/data/structuresearch.sh "${TARGET_PREFILTER}${INDEXEXT}"): the sequence database is then taken from <db> if it still
    // exists, else from the copy INSIDE the index (entries DBR1INDEX = 5 / DBR1DATA = 6 of the index DB,
    // M/src/prefiltering/PrefilteringIndexReader.cpp:15-18,116-128: a serialised DBReader index + the data blob).  The
    // persisted k-mer table itself is not read: the device rebuilds it from the sequences in 0.02-0.07 s.
    bool open(const std::string &path, std::string &err);
    // the header database of `db` (which may name a precomputed index): <db>_h if it exists, else the copy inside <db>.idx (entries
    // HDR1INDEX = 18 / HDR1DATA = 19, what the reference's IndexReader(..., SRC_HEADERS) reads, M/src/commons/IndexReader.h:55-75)
    bool openHeaders(const std::string &db, std::string &err);
    size_t size() const { return entries.size(); }
    uint32_t key(size_t id) const { return entries[id].key; }
    const char *data(size_t id) const { return base + entries[id].offset; }
    uint32_t entryLen(size_t id) const { return entries[id].length; }        // as in the index
    uint32_t seqLen(size_t id) const { return entries[id].length >= 2 ? entries[id].length - 2 : 0; }   // DBReader::getSeqLen
    uint64_t offset(size_t id) const { return entries[id].offset; }
    int64_t idOf(uint32_t key) const;                                         // DBReader::getId, -1 if absent
    int dbtype() const { return type & 0xffff; }
    int rawDbtype() const { return type; }                                   // the int32 of <db>.dbtype, extended bits included
    int extended() const { return (int) ((uint32_t) type >> 16); }
    const char *dataBase() const { return base; }
    uint64_t dataSize() const { return bytes; }
    uint64_t residues() const;                                                // getAminoAcidDBSize: sum of seqLen
private:
    std::vector<Entry> entries;                                               // sorted by key
    const char *base = nullptr;
    uint64_t bytes = 0;
    int type = 0;
    bool mapped = false;
    std::vector<char> owned;                                                  // multi-file DBs are read into memory
    const char *mapBase = nullptr;                                            // what to munmap (base may point into it)
    uint64_t mapBytes = 0;
    bool openInsideIndex(const std::string &idxPath, std::string &err, uint32_t indexKey = 5, uint32_t dataKey = 6);
};

// <db>[.idx] + suffix the way StructureUtil::getIndexWithSuffix does (F/src/commons/StructureUtil.h:9-21): "db.idx" + "_ss" ->
// "db_ss.idx" if that index exists, else "db_ss"; "db" + "_ss" -> "db_ss"
std::string dbPathWithSuffix(const std::string &db, const std::string &suffix);

class DbWriter {
public:
    bool open(const std::string &path, int dbtype, std::string &err);
    // appends data + '\0' and remembers (key, offset, size + 1)   (DBWriter::writeData with addNullByte)
    void write(uint32_t key, const char *data, size_t size);
    bool close(std::string &err);                                             // index sorted by key + .dbtype
private:
    std::string path;
    int type = 0;
    FILE *f = nullptr;
    uint64_t off = 0;
    bool failed = false;                                                      // an fwrite / fclose failed: close() reports it
    std::vector<DbReader::Entry> entries;
};

} // namespace fsh
