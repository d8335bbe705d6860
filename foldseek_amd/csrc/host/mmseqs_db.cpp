// mmseqs_db.cpp -- see mmseqs_db.h
#include "mmseqs_db.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace fsh {

static bool fileSize(const std::string &p, uint64_t &sz) {
    struct stat st;
    if (stat(p.c_str(), &st) != 0) return false;
    sz = (uint64_t) st.st_size;
    return true;
}

DbReader::~DbReader() {
    if (mapped && mapBase) munmap((void *) mapBase, mapBytes);
    else if (mapped && base) munmap((void *) base, bytes);
}

static bool endsWith(const std::string &s, const std::string &suf) { return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0; }

std::string dbPathWithSuffix(const std::string &db, const std::string &suffix) {
    if (!endsWith(db, ".idx")) return db + suffix;
    const std::string plain = db.substr(0, db.size() - 4) + suffix;
    uint64_t sz;
    return fileSize(plain + ".idx.dbtype", sz) ? plain + ".idx" : plain;
}

// sequence DB stored inside a precomputed index: DBR1INDEX (key 5) = DBReader::serialize (DBReader.cpp:824-840: size, dataSize,
// lastKey, dbtype, maxSeqLen, then `size` records {u32 id; u64 offset; u32 length} with natural alignment = 24 bytes),
// DBR1DATA (key 6) = the data file's bytes
bool DbReader::openInsideIndex(const std::string &idxPath, std::string &err, uint32_t indexKey, uint32_t dataKey) {
    DbReader outer;
    // the index DB itself is an ordinary MMseqs DB (its .dbtype is INDEX_DB = 16: no sequence semantics needed here)
    if (!outer.open(idxPath + "\x01raw", err)) return false;
    const int64_t iIdx = outer.idOf(indexKey), iDat = outer.idOf(dataKey);
    if (iIdx < 0 || iDat < 0) {
        err = idxPath + (indexKey == 5 ? ": no sequence database inside the index (DBR1INDEX / DBR1DATA missing)" : ": no header database inside the index (HDR1INDEX / HDR1DATA missing)");
        return false;
    }
    const char *p = outer.data((size_t) iIdx);
    const uint64_t need = 8 + 8 + 4 + 4 + 4;
    if (outer.entryLen((size_t) iIdx) < need) { err = idxPath + (indexKey == 5 ? ": truncated DBR1INDEX" : ": truncated HDR1INDEX"); return false; }
    uint64_t n, dataSize; int32_t dbt;
    memcpy(&n, p, 8); memcpy(&dataSize, p + 8, 8); memcpy(&dbt, p + 20, 4);
    if (n > (outer.entryLen((size_t) iIdx) - need) / 24) { err = idxPath + (indexKey == 5 ? ": truncated DBR1INDEX" : ": truncated HDR1INDEX"); return false; }
    const char *rec = p + 28;
    entries.resize(n);
    for (uint64_t i = 0; i < n; i++) {
        uint32_t id, len; uint64_t off;
        memcpy(&id, rec + i * 24, 4); memcpy(&off, rec + i * 24 + 8, 8); memcpy(&len, rec + i * 24 + 16, 4);
        entries[i] = {id, off, len};
    }
    std::stable_sort(entries.begin(), entries.end(), [](const Entry &a, const Entry &b) { return a.key < b.key; });
    type = dbt;
    // keep the outer mapping alive, point at the data blob
    base = outer.data((size_t) iDat);
    bytes = outer.entryLen((size_t) iDat);
    mapBase = outer.base; mapBytes = outer.bytes; mapped = outer.mapped;
    owned.swap(outer.owned);
    if (!mapped) { base = owned.data() + (base - outer.base); }
    outer.mapped = false; outer.base = nullptr;
    for (const Entry &e : entries)
        if (e.offset > bytes || e.length > bytes + 1 - e.offset) { err = idxPath + ": sequence entry beyond the data blob inside the index"; return false; }
    return true;
}

bool DbReader::openHeaders(const std::string &db, std::string &err) {
    const bool idx = endsWith(db, ".idx");
    const std::string plain = (idx ? db.substr(0, db.size() - 4) : db) + "_h";
    uint64_t sz;
    if (!idx || (fileSize(plain + ".dbtype", sz) && fileSize(plain + ".index", sz))) return open(plain, err);
    return openInsideIndex(db, err, 18, 19);
}

bool DbReader::open(const std::string &pathIn, std::string &err) {
    std::string path = pathIn;
    const bool raw = endsWith(path, "\x01raw");          // internal: open an index DB as the plain DB it is
    if (raw) path.resize(path.size() - 4);
    if (!raw && endsWith(path, ".idx")) {
        const std::string plain = path.substr(0, path.size() - 4);
        uint64_t sz;
        if (fileSize(plain + ".dbtype", sz) && fileSize(plain + ".index", sz)) path = plain;      // the sequence DB is still there
        else return openInsideIndex(path, err);
    }
    // .dbtype
    {
        FILE *f = fopen((path + ".dbtype").c_str(), "rb");
        if (!f) { err = "cannot open " + path + ".dbtype"; return false; }
        int32_t t = 0;
        if (fread(&t, 4, 1, f) != 1) { fclose(f); err = "short read on " + path + ".dbtype"; return false; }
        fclose(f);
        type = t;
        if (extended() & DBTYPE_EXTENDED_COMPRESSED) { err = path + ": compressed databases are not supported by this module"; return false; }
    }
    // .index
    {
        FILE *f = fopen((path + ".index").c_str(), "rb");
        if (!f) { err = "cannot open " + path + ".index"; return false; }
        uint64_t isz = 0;
        fileSize(path + ".index", isz);
        std::vector<char> buf(isz + 1);
        if (isz && fread(buf.data(), 1, isz, f) != isz) { fclose(f); err = "short read on " + path + ".index"; return false; }
        fclose(f);
        buf[isz] = 0;
        const char *p = buf.data(), *end = buf.data() + isz;
        while (p < end) {
            char *q;
            Entry e;
            e.key = (uint32_t) strtoul(p, &q, 10);
            if (q == p) break;
            e.offset = strtoull(q, &q, 10);
            e.length = (uint32_t) strtoul(q, &q, 10);
            entries.push_back(e);
            while (q < end && *q != '\n') q++;
            p = q + 1;
        }
        std::stable_sort(entries.begin(), entries.end(), [](const Entry &a, const Entry &b) { return a.key < b.key; });
    }
    // data: <db> or <db>.0, <db>.1, ...
    uint64_t sz = 0;
    if (fileSize(path, sz)) {
        int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) { err = "cannot open " + path; return false; }
        bytes = sz;
        if (sz) {
            void *m = mmap(NULL, sz, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) { ::close(fd); err = "mmap failed for " + path; return false; }
            base = (const char *) m;
            mapped = true;
        }
        ::close(fd);
    } else {
        for (int k = 0;; k++) {
            const std::string part = path + "." + std::to_string(k);
            uint64_t psz = 0;
            if (!fileSize(part, psz)) break;
            FILE *f = fopen(part.c_str(), "rb");
            if (!f) { err = "cannot open " + part; return false; }
            const size_t old = owned.size();
            owned.resize(old + psz);
            if (psz && fread(owned.data() + old, 1, psz, f) != psz) { fclose(f); err = "short read on " + part; return false; }
            fclose(f);
        }
        if (owned.empty() && !entries.empty()) { err = "cannot find data file for " + path; return false; }
        base = owned.data();
        bytes = owned.size();
    }
    // padded GPU databases store L residues (+ padding) but index L + 2 (makepaddedseqdb.cpp:96-104)
    const bool gpuDb = (extended() & DBTYPE_EXTENDED_GPU) != 0;
    for (const Entry &e : entries) {
        const uint64_t need = gpuDb ? (e.length >= 2 ? e.length - 2 : 0) : e.length;
        if (e.offset > bytes || need > bytes - e.offset) { err = path + ": index entry beyond end of data file"; return false; }      // no sum: a damaged offset must not wrap
    }
    return true;
}

int64_t DbReader::idOf(uint32_t k) const {
    auto it = std::lower_bound(entries.begin(), entries.end(), k, [](const Entry &e, uint32_t v) { return e.key < v; });
    if (it == entries.end() || it->key != k) return -1;
    return (int64_t) (it - entries.begin());
}

uint64_t DbReader::residues() const {
    uint64_t r = 0;
    for (const Entry &e : entries) r += e.length >= 2 ? e.length - 2 : 0;
    return r;
}

bool DbWriter::open(const std::string &p, int dbtype, std::string &err) {
    path = p; type = dbtype; off = 0; entries.clear(); failed = false;
    f = fopen(p.c_str(), "wb");
    if (!f) { err = "cannot create " + p; return false; }
    return true;
}

void DbWriter::write(uint32_t key, const char *data, size_t size) {
    // a failed write (disk full, I/O error) is remembered and reported by close(): a truncated data file with a
    // valid-looking index must never end in EXIT_SUCCESS
    if (size && fwrite(data, 1, size, f) != size) failed = true;
    const char nul = 0;
    if (fwrite(&nul, 1, 1, f) != 1) failed = true;
    DbReader::Entry e; e.key = key; e.offset = off; e.length = (uint32_t) (size + 1);
    entries.push_back(e);
    off += size + 1;
}

bool DbWriter::close(std::string &err) {
    if (f) { if (fclose(f) != 0) failed = true; f = nullptr; }
    if (failed) { err = "write error on " + path; return false; }
    std::stable_sort(entries.begin(), entries.end(), [](const DbReader::Entry &a, const DbReader::Entry &b) { return a.key < b.key; });
    FILE *fi = fopen((path + ".index").c_str(), "wb");
    if (!fi) { err = "cannot create " + path + ".index"; return false; }
    bool ok = true;
    for (const auto &e : entries) ok = (fprintf(fi, "%u\t%llu\t%u\n", e.key, (unsigned long long) e.offset, e.length) > 0) && ok;
    ok = (fclose(fi) == 0) && ok;
    if (!ok) { err = "write error on " + path + ".index"; return false; }
    FILE *ft = fopen((path + ".dbtype").c_str(), "wb");
    if (!ft) { err = "cannot create " + path + ".dbtype"; return false; }
    int32_t t = type;
    ok = fwrite(&t, 4, 1, ft) == 1;
    ok = (fclose(ft) == 0) && ok;
    if (!ok) { err = "write error on " + path + ".dbtype"; return false; }
    return true;
}

} // namespace fsh
