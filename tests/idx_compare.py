"""Entry-by-entry comparison of two precomputed index DBs (<db>.idx of PrefilteringIndexReader::createIndexFile): same keys, offsets and
lengths, same bytes -- except where the reference itself writes indeterminate bytes: the struct padding inside serialised DBReader index
records (keys 5 / 7 / 18 / 20, DBReader.cpp:837), the one byte past the sequence lookup (key 14, SequenceLookup.cpp:13 + :300 of the index
writer) and the GENERATOR string (key 22: the writer's version).  TEST INFRASTRUCTURE."""
import struct

import numpy as np

SERIALISED_READERS = {5, 7, 18, 20, 500, 502}
GENERATOR, SEQINDEXDATA = 22, 14


def read_index(path):
    return [tuple(int(x) for x in l.split()) for l in open(path + ".index") if l.strip()]


def parse_reader(b):
    size, data_size = struct.unpack_from("<QQ", b, 0)
    last_key, dbtype, max_len = struct.unpack_from("<IiI", b, 16)
    rec = np.frombuffer(b, dtype=np.dtype([("id", "<u4"), ("pad0", "<u4"), ("offset", "<u8"), ("length", "<u4"), ("pad1", "<u4")]), count=size, offset=28)
    return (size, data_size, last_key, dbtype, max_len), rec


def compare(ref, mine, chunk=1 << 26):
    """returns a list of human-readable differences (empty = equivalent)"""
    a, b = read_index(ref), read_index(mine)
    bad = []
    if open(ref + ".dbtype", "rb").read() != open(mine + ".dbtype", "rb").read():
        bad.append("dbtype differs")
    if [k for k, _, _ in a] != [k for k, _, _ in b]:
        return bad + [f"key sets differ: {[k for k, _, _ in a]} vs {[k for k, _, _ in b]}"]
    fa, fb = open(ref, "rb"), open(mine, "rb")
    for (k, oa, la), (_, ob, lb) in zip(a, b):
        if k == GENERATOR:
            continue
        if (oa, la) != (ob, lb):
            bad.append(f"key {k}: offset/length {oa}/{la} vs {ob}/{lb}")
            continue
        if k in SERIALISED_READERS:
            fa.seek(oa); fb.seek(ob)
            ha, ra = parse_reader(fa.read(la))
            hb, rb = parse_reader(fb.read(lb))
            if ha != hb or not all((ra[f] == rb[f]).all() for f in ("id", "offset", "length")):
                bad.append(f"key {k}: serialised reader differs ({ha} vs {hb})")
            continue
        pos = 0
        while pos < la:
            n = min(chunk, la - pos)
            fa.seek(oa + pos); fb.seek(ob + pos)
            xa, xb = fa.read(n), fb.read(n)
            if xa != xb:
                va, vb = np.frombuffer(xa, np.uint8), np.frombuffer(xb, np.uint8)
                d = np.flatnonzero(va != vb)
                if k == SEQINDEXDATA:
                    d = d[pos + d != la - 2]                 # the byte past the lookup data (before the entry's terminator)
                if len(d):
                    bad.append(f"key {k}: {len(d)} bytes differ in [{pos}, {pos + n}), first at {pos + int(d[0])}")
                    break
            pos += n
    return bad


if __name__ == "__main__":
    import sys
    out = compare(sys.argv[1], sys.argv[2])
    print("\n".join(out) if out else "equivalent")
    sys.exit(1 if out else 0)
