"""Writer of a Jellyfish 2 "binary/sorted" count file in the layout talc_amd/csrc/talc_jf.h reads (test infrastructure).
The layout is Jellyfish 2.x's generic_file_header + binary_dumper as recollected — there is no `jellyfish` program here
to produce a real one, so this pins the reader to the documented layout, not to the tool."""
import json

import numpy as np

CODE = {"A": 0, "C": 1, "G": 2, "T": 3}


def pack(kmer):
    v = 0
    for ch in kmer.upper():
        v = (v << 2) | CODE[ch]
    return v


def header_bytes(k, counter_len=4, fmt="binary/sorted", alignment=8, extra=None):
    h = {"alignment": alignment, "canonical": False, "cmdline": ["jellyfish", "count", "-m", str(k), "-s", "100M", "reads.fa"],
         "counter_len": counter_len, "format": fmt, "key_len": 2 * k, "max_reprobe": 126, "size": 134217728, "val_len": 7}
    h.update(extra or {})
    js = json.dumps(h, indent=3).encode()
    hlen = len(js)
    pad = (9 + hlen) % alignment if alignment else 0
    if pad:
        hlen += alignment - pad
    return b"%09d" % hlen + js + b"\0" * (hlen - len(js))


def write_jf(path, entries, k, counter_len=4, **kw):
    """entries: iterable of (kmer text or packed int, count), written in the order given (a real file is in hash order)."""
    kb = (2 * k + 7) // 8
    with open(path, "wb") as f:
        f.write(header_bytes(k, counter_len, **kw))
        for km, c in entries:
            v = pack(km) if isinstance(km, str) else int(km)
            f.write(int(v).to_bytes(kb, "little") + int(c).to_bytes(counter_len, "little"))


def dump_to_jf(dump_path, jf_path, k, seed=1, **kw):
    """The text dump's entries as a .jf, in a shuffled order; of a k-mer listed twice the first entry (the one the
    reference's map keeps, Jellyfish.cpp:262) — a count file holds every k-mer once."""
    ent, seen = [], set()
    with open(dump_path) as f:
        for line in f:
            t = line.split()
            if len(t) >= 2 and t[0] not in seen:
                seen.add(t[0])
                ent.append((t[0], int(t[1])))
    order = np.random.default_rng(seed).permutation(len(ent))
    write_jf(jf_path, [ent[i] for i in order], k, **kw)
    return len(ent)
