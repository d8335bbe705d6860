"""Read / write MMseqs2-layout databases from Python (test + tooling helper; the product's DB I/O is the C++ in
csrc/host/mmseqs_db.cpp).  Layout: SURVEY.md section 8b."""
import numpy as np
from .synth import ALPHABET, PaddedDB


def write_seq_db(path, seqs_codes, keys=None, lower_mask=None, dbtype=0):
    """ASCII sequence DB: entry = letters + '\\n' + '\\0'; index length = L + 2."""
    keys = list(range(len(seqs_codes))) if keys is None else list(keys)
    off = 0
    with open(path, "wb") as f, open(path + ".index", "w") as fi:
        for i, (k, s) in enumerate(zip(keys, seqs_codes)):
            txt = "".join(ALPHABET[c] for c in s)
            if lower_mask is not None and lower_mask[i] is not None:
                txt = "".join(ch.lower() if m else ch for ch, m in zip(txt, lower_mask[i]))
            b = txt.encode() + b"\n\0"
            f.write(b)
            fi.write(f"{k}\t{off}\t{len(b)}\n")
            off += len(b)
    np.array([dbtype], dtype=np.int32).tofile(path + ".dbtype")


def write_padded_db(path, db: PaddedDB, which="3di"):
    data = db.data3di if which == "3di" else db.dataaa
    np.ascontiguousarray(data, np.uint8).tofile(path)
    with open(path + ".index", "w") as fi:
        for i in range(db.n):
            fi.write(f"{i}\t{int(db.offsets[i])}\t{int(db.lengths[i]) + 2}\n")
    np.array([0 | (8 << 16)], dtype=np.int32).tofile(path + ".dbtype")


def read_db(path):
    """returns (dbtype, {key: bytes-without-terminator})"""
    t = int(np.fromfile(path + ".dbtype", dtype=np.int32)[0])
    data = open(path, "rb").read()
    out = {}
    for line in open(path + ".index"):
        k, o, l = line.split()
        k, o, l = int(k), int(o), int(l)
        out[k] = data[o:o + l - 1]
    return t, out


def read_padded_db(path):
    t = int(np.fromfile(path + ".dbtype", dtype=np.int32)[0])
    data = np.fromfile(path, dtype=np.uint8)
    rows = [tuple(int(x) for x in line.split()) for line in open(path + ".index")]
    rows.sort()
    offsets = np.array([r[1] for r in rows] + [data.size], dtype=np.int64)
    lengths = np.array([r[2] - 2 for r in rows], dtype=np.int32)
    return t, data, offsets, lengths


def write_seq_db_from_padded(path, db: PaddedDB, which="3di", keys=None):
    """ASCII sequence DB (letters + '\\n' + '\\0' per entry, soft-masked residues in lower case) of every entry of a PaddedDB, vectorised:
    200k entries in well under a second.  keys default to 0..n-1."""
    data = db.data3di if which == "3di" else db.dataaa
    n = db.n
    lens = db.lengths.astype(np.int64)
    out_off = np.zeros(n + 1, np.int64)
    out_off[1:] = np.cumsum(lens + 2)
    # source index of every residue: entry offset + position inside the entry
    ent = np.repeat(np.arange(n), lens)
    pos = np.arange(int(lens.sum())) - np.repeat(np.cumsum(lens) - lens, lens)
    codes = data[db.offsets[:-1][ent] + pos]
    lut = np.frombuffer((ALPHABET + "X" * 11 + ALPHABET.lower() + "x" * 11).encode(), np.uint8)      # code + 32 = soft-masked -> lower case
    out = np.empty(int(out_off[-1]), np.uint8)
    out[out_off[:-1][ent] + pos] = lut[np.minimum(codes, 63)]
    out[out_off[1:] - 2] = ord("\n")
    out[out_off[1:] - 1] = 0
    out.tofile(path)
    keys = np.arange(n) if keys is None else np.asarray(keys)
    with open(path + ".index", "w") as fi:
        fi.write("".join(f"{int(k)}\t{int(o)}\t{int(l) + 2}\n" for k, o, l in zip(keys, out_off[:-1], lens)))
    np.array([0], dtype=np.int32).tofile(path + ".dbtype")
