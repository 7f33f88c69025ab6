"""Test-side WRITER of KenLM binary models (probing, rest-probing, trie and the quantised / Bhiksha-array trie
variants: model types 0-5), following the layout cited in
ppasr_amd/csrc/klm.hip.  KenLM cannot be built here (not vendored, no network), so reader and writer are both written
from the documented format; the writer sets KenLM's flag bits (prob sign = "extends left", back-off -0.0 = "no
extension") the way build_binary does, so that the reader has to strip them.

    write_klm(arpa_path, klm_path, model_type="probing" | "rest_probing" | "trie" | "quant_trie" | "array_trie" |
              "quant_array_trie", multiplier=1.5, prob_bits=.., backoff_bits=.., bhiksha_bits=..)

Quantised models: build_binary trains its bins on the model; here the bins are the model's own distinct values (padded
to 2^bits entries), so a quantised binary still scores exactly like its ARPA source -- what is tested is the LAYOUT
(bin tables in front of the unigrams, [backoff bin][prob bin] records, the reserved back-off bins 0 / 1).
"""
import struct

import numpy as np

MAGIC = b"mmap lm http://kheafield.com/code format version 5\n\x00"
M64 = (1 << 64) - 1


def murmur64a(data, seed=0):
    """util/murmur_hash.cc MurmurHash64A."""
    m, r = 0xc6a4a7935bd1e995, 47
    n = len(data)
    h = (seed ^ (n * m)) & M64
    for i in range(0, n - n % 8, 8):
        k = struct.unpack_from("<Q", data, i)[0]
        k = (k * m) & M64
        k ^= k >> r
        k = (k * m) & M64
        h ^= k
        h = (h * m) & M64
    tail = data[n - n % 8:]
    if tail:
        for i in reversed(range(len(tail))):
            h ^= tail[i] << (8 * i)
        h = (h * m) & M64
    h ^= h >> r
    h = (h * m) & M64
    h ^= h >> r
    return h


def combine(cur, nxt):
    """lm/search_hashed.hh detail::CombineWordHash."""
    return ((cur * 8978948897894561157) & M64) ^ (((1 + nxt) * 17894857484156487943) & M64)


def chain(ids):
    """key of the n-gram ids[0..n-1]: start from the last word, fold the preceding ones in, newest first."""
    h = ids[-1]
    for w in reversed(ids[:-1]):
        h = combine(h, w)
    return h


def read_arpa_grams(path):
    """-> (words list in first-seen order with <unk> first, {n: [(ids tuple, prob, backoff)]})"""
    words = {"<unk>": 0}
    grams, section = {}, 0
    with open(path, encoding="utf-8") as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            if line.startswith("\\"):
                section = int(line[1:line.index("-")]) if line.endswith("-grams:") else 0
                continue
            if section == 0:
                continue
            parts = line.split()
            ws = parts[1:1 + section]
            for w in ws:
                words.setdefault(w, len(words))
            bo = float(parts[1 + section]) if len(parts) > 1 + section else 0.0
            grams.setdefault(section, []).append((tuple(words[w] for w in ws), float(parts[0]), bo))
    return list(words), grams


def _buckets(entries, mult):
    return max(entries + 1, int(np.float32(mult) * np.float32(entries)))


def _f32(v):
    return struct.pack("<f", v)


def _header(order, mult, model_type, counts):
    sanity = MAGIC.ljust(56, b"\x00") + struct.pack("<fffII", 0.0, 1.0, -0.5, 1, 0xFFFFFFFF) + b"\x00" * 4 + struct.pack("<Q", 1)
    assert len(sanity) == 88
    fixed = struct.pack("<B3xfiB3xI", order, mult, model_type, 1, 1)
    assert len(fixed) == 20
    out = sanity + fixed + b"".join(struct.pack("<Q", c) for c in counts)
    return out.ljust((len(out) + 7) // 8 * 8, b"\x00")


def _probing_table(items, buckets, entry_bytes, pack_value):
    """items: [(key, value...)] -> bytes of a linear-probing table, empty key 0 (util/probing_hash_table.hh)."""
    tab = [None] * buckets
    for key, *val in items:
        assert key != 0
        i = key % buckets
        while tab[i] is not None:
            i = (i + 1) % buckets
        tab[i] = (key, val)
    out = bytearray()
    for e in tab:
        if e is None:
            out += b"\x00" * entry_bytes
        else:
            out += (struct.pack("<Q", e[0]) + pack_value(*e[1])).ljust(entry_bytes, b"\x00")
    return bytes(out)


def _flags(grams, order):
    """extends-left / has-extension flags like KenLM's builder: an n-gram "extends left" when some (n+1)-gram has it as
    its suffix, "has an extension" (to the right) when some (n+1)-gram has it as its prefix."""
    suffix_of, prefix_of = set(), set()
    for n in range(2, order + 1):
        for ids, _, _ in grams.get(n, []):
            suffix_of.add(ids[1:])
            prefix_of.add(ids[:-1])
    return suffix_of, prefix_of


def write_klm(arpa_path, klm_path, model_type="probing", multiplier=1.5, prob_bits=12, backoff_bits=12, bhiksha_bits=64):
    words, grams = read_arpa_grams(arpa_path)
    order = max(grams)
    counts = [len(grams[n]) for n in range(1, order + 1)]
    assert counts[0] == len(words)
    suffix_of, prefix_of = _flags(grams, order)

    def stored_prob(ids, p):   # sign bit cleared <=> extends left
        return abs(p) if ids in suffix_of else -abs(p)

    def stored_backoff(ids, b):  # -0.0 = kNoExtensionBackoff
        return b if (ids in prefix_of or b != 0.0) else -0.0

    if model_type in ("probing", "rest_probing"):
        rest = model_type == "rest_probing"
        wsize = 12 if rest else 8
        mt = 1 if rest else 0

        def weights(p, b):
            return _f32(p) + _f32(b) + (_f32(p) if rest else b"")

        body = bytearray()
        # vocabulary: header {version 0, bound} + probing table {hash, index}
        body += struct.pack("<II", 0, len(words))
        vitems = [(murmur64a(w.encode("utf-8")), i) for i, w in enumerate(words) if i > 0]
        body += _probing_table(vitems, _buckets(counts[0], multiplier), 16, lambda i: struct.pack("<I", i))
        uni = {ids[0]: (p, b) for ids, p, b in grams[1]}
        for w in range(counts[0] + 1):
            if w in uni:
                p, b = uni[w]
                body += weights(stored_prob((w,), p), stored_backoff((w,), b))
            else:
                body += b"\x00" * wsize
        for n in range(2, order):
            items = [(chain(ids), stored_prob(ids, p), stored_backoff(ids, b)) for ids, p, b in grams[n]]
            body += _probing_table(items, _buckets(counts[n - 1], multiplier), 8 + wsize, weights)
        items = [(chain(ids), -abs(p)) for ids, p, _ in grams[order]]
        body += _probing_table(items, _buckets(counts[order - 1], multiplier), 16, _f32)
        strings = words
    elif model_type in ("trie", "quant_trie", "array_trie", "quant_array_trie"):
        quant, array = model_type.startswith("quant"), "array" in model_type
        mt = 2 + (1 if quant else 0) + (2 if array else 0)
        # SortedVocabulary: words renumbered by the order of their hashes, <unk> = 0
        hashed = sorted((murmur64a(w.encode("utf-8")), w) for w in words[1:])
        new_index = {"<unk>": 0}
        for i, (_, w) in enumerate(hashed):
            new_index[w] = i + 1
        remap = [new_index[w] for w in words]
        strings = ["<unk>"] + [w for _, w in hashed]
        body = bytearray(struct.pack("<Q", len(hashed)) + b"".join(struct.pack("<Q", h) for h, _ in hashed))
        body += b"\x00" * (8 * (counts[0] - len(hashed)))  # SortedVocabulary::Size = 8 + 8 * entries (entries counts <unk>)
        body = body[:8 + 8 * counts[0]]
        # reverse trie: level n holds the n-grams sorted by (w_n, w_{n-1}, ..., w_1)
        level = {n: sorted(((tuple(remap[w] for w in reversed(ids)), p, b) for ids, p, b in grams[n]), key=lambda t: t[0])
                 for n in range(1, order + 1)}

        def first_child(n, prefix_rev, start_from):
            """index of the first record of level n+1 whose reversed ids start with prefix_rev (records are sorted)."""
            recs = level[n + 1]
            i = start_from
            while i < len(recs) and recs[i][0][:n] < prefix_rev:
                i += 1
            return i

        # ---- SeparatelyQuantize tables (lm/quantize.cc): in FRONT of the unigrams ----
        pbin, bbin = {}, {}
        if quant:
            def f32v(v):
                return struct.unpack("<f", _f32(v))[0]
            qbody = bytearray(struct.pack("<BBB5x", 2, prob_bits, backoff_bits))
            for n in range(2, order + 1):
                pv = sorted({f32v(-abs(p)) for _, p, _ in level[n]})
                assert len(pv) <= (1 << prob_bits), "more distinct probabilities than bins"
                pbin[n] = {v: i for i, v in enumerate(pv)}
                qbody += b"".join(_f32(v) for v in pv) + _f32(float("inf")) * ((1 << prob_bits) - len(pv))
                if n < order:
                    # bins 0 / 1 are reserved: kNoExtensionBackoff (-0.0) and kExtensionBackoff (0.0)
                    bv = sorted({f32v(b) for _, _, b in level[n] if b != 0.0})
                    assert len(bv) + 2 <= (1 << backoff_bits), "more distinct back-offs than bins"
                    bbin[n] = {v: i + 2 for i, v in enumerate(bv)}
                    qbody += _f32(-0.0) + _f32(0.0) + b"".join(_f32(v) for v in bv) + _f32(float("inf")) * ((1 << backoff_bits) - 2 - len(bv))
            assert len(qbody) == (order - 2) * 4 * ((1 << prob_bits) + (1 << backoff_bits)) + 4 * (1 << prob_bits) + 8
            body += qbody

        word_bits = int(counts[0]).bit_length()
        uni = {rev[0]: (p, b) for rev, p, b in level[1]}
        ptr = 0
        urecs = []
        for w in range(counts[0] + 2):
            if order >= 2:
                ptr = first_child(1, (w,), ptr) if w < counts[0] else len(level[2])
            p, b = uni.get(w, (0.0, 0.0))
            urecs.append(struct.pack("<ffQ", -abs(p), b, ptr))
        body += b"".join(urecs)

        def pack_bits(records, total_bits, n_records):
            nbytes = ((1 + n_records) * total_bits + 7) // 8 + 8
            acc = 0
            for r, fields in enumerate(records):
                off = r * total_bits
                for value, bits in fields:
                    acc |= (value & ((1 << bits) - 1)) << off
                    off += bits
            return acc.to_bytes(nbytes, "little")

        def f32bits(v):
            return struct.unpack("<I", _f32(v))[0]

        def quant_fields(n, p, b, last):
            if quant:
                pi = pbin[n][struct.unpack("<f", _f32(-abs(p)))[0]]
                if last:
                    return [(pi, prob_bits)]
                if b == 0.0:
                    bi = 1 if (False) else 0   # either reserved bin decodes to a zero back-off
                else:
                    bi = bbin[n][struct.unpack("<f", _f32(b))[0]]
                return [(bi, backoff_bits), (pi, prob_bits)]
            if last:
                return [(f32bits(abs(p)) & 0x7FFFFFFF, 31)]
            return [(f32bits(abs(p)) & 0x7FFFFFFF, 31), (f32bits(b), 32)]

        for n in range(2, order):
            max_next, max_offset = counts[n], counts[n - 1] + 1
            required = int(max_next).bit_length()
            ptrs, ptr = [], 0
            for rev, p, b in level[n]:
                ptr = first_child(n, rev, ptr)
                ptrs.append(ptr)
            ptrs.append(len(level[n + 1]))  # the final next pointer
            chop = 0
            if array:
                # lm/bhiksha.cc ChopBits / ArrayCount
                lowest = None
                for c in range(0, min(required, bhiksha_bits) + 1):
                    change = (max_next >> (required - c)) * 64 - max_offset * c
                    if lowest is None or change < lowest:
                        lowest, chop = change, c
                inline_bits = required - chop
                n_off = (max_next >> (required - chop)) + 1
                offsets = [0] * n_off
                w_to = 1
                for index, value in enumerate(ptrs):           # ArrayBhiksha::WriteNext
                    enc = value >> inline_bits
                    while w_to <= enc:
                        offsets[w_to] = index
                        w_to += 1
                assert w_to == n_off, (w_to, n_off)
                blk = bytearray(8 * (1 + n_off) + 7)
                start = len(_header(order, multiplier, mt, counts)) + len(body)   # file offset of the block
                a8 = (-start) % 8
                blk[0], blk[1] = 0, bhiksha_bits                # version, configured bits (FinishedLoading)
                struct.pack_into("<%dQ" % n_off, blk, a8 + 8, *offsets)
                body += blk
            else:
                inline_bits = required
            total = word_bits + (prob_bits + backoff_bits if quant else 63) + inline_bits
            recs = []
            for (rev, p, b), pt in zip(level[n], ptrs):
                recs.append([(rev[-1], word_bits)] + quant_fields(n, p, b, False) + [(pt & ((1 << inline_bits) - 1), inline_bits)])
            recs.append([(0, word_bits)] + [(0, prob_bits + backoff_bits if quant else 63)] + [(ptrs[-1] & ((1 << inline_bits) - 1), inline_bits)])
            body += pack_bits(recs, total, counts[n - 1])
        total = word_bits + (prob_bits if quant else 31)
        recs = [[(rev[-1], word_bits)] + quant_fields(order, p, b, True) for rev, p, b in level[order]]
        body += pack_bits(recs, total, counts[order - 1])
    else:
        raise ValueError(model_type)
    with open(klm_path, "wb") as f:
        f.write(_header(order, multiplier, mt, counts))
        f.write(bytes(body))
        f.write(b"".join(w.encode("utf-8") + b"\x00" for w in strings))
    return klm_path


def patch_model_type(klm_path, model_type):
    with open(klm_path, "r+b") as f:
        f.seek(96)
        f.write(struct.pack("<i", model_type))
