// klm.hip -- reader of KenLM binary language models (.klm), the format PPASR ships its scorer model in
// (decoders/beam_search_decoder.py:19-29 downloads lm/zh_giga.no_cna_cmn.prune01244.klm; swig_wrapper.py:18-33 hands the
// path to paddlespeech_ctcdecoders' Scorer, which loads it through KenLM's lm::ngram::LoadVirtual).  Host-only code.
//
// KenLM is not part of /root/reference (un-vendored third-party dependency), so the byte layout below is written from
// the KenLM sources as recalled (file names of https://github.com/kpu/kenlm given per item); the reader is exercised
// against tests/klm_writer.py, a writer that follows the same layout, not against a file produced by KenLM itself.
//
// File layout (lm/binary_format.cc):
//   [0]   Sanity    char magic[56] = "mmap lm http://kheafield.com/code format version 5\n\0" zero-padded,
//                   float 0, 1, -0.5; uint32 1, 0xffffffff; (4 bytes pad) uint64 1                       = 88 bytes
//   [88]  FixedWidthParameters { uint8 order; float probing_multiplier; int32 model_type; bool has_vocabulary;
//                   uint32 search_version }  with natural alignment                                       = 20 bytes
//   [108] uint64 counts[order]
//   header size = ALIGN8(88 + 20 + 8 * order); then the vocabulary memory, the search memory and, if
//   has_vocabulary, every word as a NUL-terminated string in index order ("<unk>" first).
// model_type (lm/model_type.hh): PROBING 0, REST_PROBING 1, TRIE 2, QUANT_TRIE 3, ARRAY_TRIE 4, QUANT_ARRAY_TRIE 5.
//
// PROBING / REST_PROBING (lm/vocab.hh ProbingVocabulary, lm/search_hashed.hh HashedSearch, util/probing_hash_table.hh):
//   vocabulary : { uint32 version; uint32 bound } + probing table of { uint64 MurmurHash64A(word, seed 0); uint32 index }
//                (16-byte entries), buckets = max(n + 1, (uint64)(multiplier * (float)n)), n = counts[0]
//   unigrams   : (counts[0] + 1) x { float prob; float backoff [; float rest] }, indexed by word
//   middle n   : probing table of { uint64 key; float prob; float backoff [; float rest] }, buckets as above with
//                n = counts[n-1]; key = the word-hash chain of lm.h (kenlm_chain), empty slot = key 0
//   longest    : probing table of { uint64 key; float prob } (16-byte entries)
//   The sign bit of a stored prob is a flag ("extends left"), the probability is -|stored|; a back-off of -0.0 means
//   "no extension" and counts as 0.  The tables are taken over as they are: every entry keeps KenLM's key
//   (LmDev::kenlm_keys), because the words of an n-gram cannot be recovered from its hash.
//
// TRIE (lm/vocab.hh SortedVocabulary, lm/search_trie.hh, lm/trie.hh, util/bit_packing.hh), quantisation and Bhiksha
// pointer compression off:
//   vocabulary : uint64 count + sorted uint64 word hashes; word index = position + 1, <unk> = 0
//   unigrams   : (counts[0] + 2) x { float prob; float backoff; uint64 next }
//   middle n   : bit-packed records [word : bits(counts[0])] [prob : 31, sign dropped] [backoff : 32]
//                [next : bits(counts[n])], 1 + counts[n-1] records, ((1 + entries) * bits + 7) / 8 + 8 bytes
//   longest    : bit-packed [word] [prob : 31]
//   A record's children are [next, next of the following record) one order up; the trie is keyed by the n-gram's
//   words from the LAST to the first.  The walk enumerates every n-gram with its word ids, which are re-keyed with
//   lm_key like an ARPA model.
// QUANT_TRIE / QUANT_ARRAY_TRIE (lm/quantize.hh SeparatelyQuantize; `build_binary -q N -b M trie`): the search memory
//   STARTS with the quantiser: { uint8 version = 2; uint8 prob_bits; uint8 backoff_bits; 5 pad } then, per middle order
//   2 .. order-1, 2^prob_bits float prob bins + 2^backoff_bits float back-off bins, then 2^prob_bits float bins of the
//   longest order (Size = (order - 2) * middle_table + longest_table + 8); unigrams are not quantised.  A middle record is
//   [word] [backoff bin : backoff_bits] [prob bin : prob_bits] [next], a longest record [word] [prob bin]; value = bin[i].
// ARRAY_TRIE / QUANT_ARRAY_TRIE (lm/bhiksha.hh ArrayBhiksha; `build_binary -a N trie`): every middle block starts with
//   the pointer-compression table: { uint8 version = 0; uint8 configured max bits }, then at the next 8-byte boundary an
//   8-byte header slot followed by ArrayCount uint64 offsets (Size = 8 * (1 + ArrayCount) + 7), then -- unaligned -- the
//   bit-packed records, whose `next` field keeps only the low InlineBits = RequiredBits(max_next) - chop bits.  chop =
//   argmin over c in [0, min(RequiredBits(max_next), configured)] of (max_next >> (required - c)) * 64 - (entries + 1) * c
//   (first minimum); ArrayCount = (max_next >> (required - chop)) + 1; offsets[e] = index of the first record whose
//   pointer's high part is >= e, so next(r) = ((upper_bound(offsets, r) - offsets - 1) << InlineBits) | inline(r).
//   The configured bits are read from the FIRST middle block (right behind the unigrams) and hold for every order.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <limits>
#include <cstring>
#include <functional>
#include <memory>

#include "lm_host.h"

using namespace ppasr;

namespace {

constexpr size_t kSanityBytes = 88, kFixedBytes = 20;
const char kMagic[] = "mmap lm http://kheafield.com/code format version 5\n";

struct Mapped {
  const uint8_t* p = nullptr;
  size_t n = 0;
  int fd = -1;
  ~Mapped() {
    if (p) munmap(const_cast<uint8_t*>(p), n);
    if (fd >= 0) close(fd);
  }
};

template <class T>
T rd(const uint8_t* p) {
  T v;
  std::memcpy(&v, p, sizeof(T));
  return v;
}
size_t align8(size_t v) { return (v + 7) & ~(size_t)7; }
uint64_t buckets_for(uint64_t entries, float multiplier) {
  const uint64_t a = entries + 1, b = (uint64_t)(multiplier * (float)entries);
  return a > b ? a : b;
}
int required_bits(uint64_t max_value) {
  if (!max_value) return 0;
  int r = 1;
  while (max_value >>= 1) ++r;
  return r;
}
float neg_abs(float stored) {  // probability with the flag bit removed: log10 P <= 0
  uint32_t u;
  std::memcpy(&u, &stored, 4);
  u |= 0x80000000u;
  float f;
  std::memcpy(&f, &u, 4);
  return f == 0.f ? 0.f : f;  // -0.0 -> 0
}
// util/bit_packing.hh, little endian: value = (unaligned u64 at byte (bit_off >> 3)) >> (bit_off & 7), masked
uint64_t read_bits(const uint8_t* base, uint64_t bit_off, int length) {
  if (length == 0) return 0;
  const uint64_t v = rd<uint64_t>(base + (bit_off >> 3)) >> (bit_off & 7);
  return length >= 64 ? v : (v & ((1ull << length) - 1));
}

}  // namespace

ppasr_status klm_load(const char* path, const char* const* vocab_utf8, int V, bool host_only, ppasr_lm_handle* out) {
  if (!path || !vocab_utf8 || V <= 0 || !out) return fail(PPASR_EINVAL, "lm: null argument");
  Mapped m;
  m.fd = open(path, O_RDONLY);
  if (m.fd < 0) return fail(PPASR_EINVAL, std::string("lm: cannot open ") + path);
  struct stat stt;
  if (fstat(m.fd, &stt) != 0 || stt.st_size < (off_t)(kSanityBytes + kFixedBytes + 8))
    return fail(PPASR_EINVAL, "lm: file too short for a KenLM binary");
  m.n = (size_t)stt.st_size;
  void* mp = mmap(nullptr, m.n, PROT_READ, MAP_PRIVATE, m.fd, 0);
  if (mp == MAP_FAILED) return fail(PPASR_EINVAL, "lm: mmap failed");
  m.p = static_cast<const uint8_t*>(mp);
  const uint8_t* f = m.p;
  // ---- Sanity ----
  if (std::memcmp(f, kMagic, sizeof(kMagic) - 1) != 0) {
    if (std::memcmp(f, "mmap lm http://kheafield.com/code format version", 48) == 0)
      return fail(PPASR_EUNSUPPORTED, "lm: KenLM binary of a format version other than 5");
    return fail(PPASR_EINVAL, "lm: not a KenLM binary (bad magic)");
  }
  if (rd<float>(f + 56) != 0.f || rd<float>(f + 60) != 1.f || rd<float>(f + 64) != -0.5f || rd<uint32_t>(f + 68) != 1u ||
      rd<uint32_t>(f + 72) != 0xffffffffu || rd<uint64_t>(f + 80) != 1ull)
    return fail(PPASR_EINVAL, "lm: KenLM sanity block mismatch (file written on an incompatible architecture?)");
  // ---- FixedWidthParameters + counts ----
  const int order = rd<uint8_t>(f + 88);
  const float mult = rd<float>(f + 92);
  const int model_type = rd<int32_t>(f + 96);
  const bool has_vocab = rd<uint8_t>(f + 100) != 0;
  if (order < 2 || order > kLmMaxOrder) return fail(PPASR_EUNSUPPORTED, "lm: model order outside 2..6");
  if (m.n < kSanityBytes + kFixedBytes + 8 * (size_t)order) return fail(PPASR_EINVAL, "lm: truncated header");
  std::vector<uint64_t> counts(order);
  for (int i = 0; i < order; ++i) counts[i] = rd<uint64_t>(f + 108 + 8 * i);
  if (!has_vocab)
    return fail(PPASR_EUNSUPPORTED, "lm: the binary was built without its vocabulary strings (build_binary -v); the scorer "
                                    "needs them, as paddlespeech_ctcdecoders does");
  size_t pos = align8(kSanityBytes + kFixedBytes + 8 * (size_t)order);
  const bool probing = model_type == 0 || model_type == 1;
  const bool trie = model_type >= 2 && model_type <= 5;
  const bool quant = model_type == 3 || model_type == 5, array = model_type == 4 || model_type == 5;
  if (!probing && !trie) return fail(PPASR_EUNSUPPORTED, "lm: unknown KenLM model type " + std::to_string(model_type));
  auto lm = std::make_unique<ppasr_lm_s>();
  lm->order = order;
  std::vector<LmEntry> entries;
  size_t strings_at = 0;

  if (probing) {
    if (!(mult > 1.0f)) return fail(PPASR_EINVAL, "lm: probing multiplier must be > 1");
    const size_t wsize = model_type == 1 ? 12 : 8;  // RestWeights carry a third float
    const size_t vocab_bytes = align8(8) + buckets_for(counts[0], mult) * 16;
    const uint8_t* uni = f + pos + vocab_bytes;
    size_t spos = pos + vocab_bytes + (counts[0] + 1) * wsize;
    std::vector<std::pair<size_t, uint64_t>> tables;  // (offset, buckets) of the middle tables and the longest one
    for (int n = 2; n < order; ++n) {
      const uint64_t b = buckets_for(counts[n - 1], mult);
      tables.emplace_back(spos, b);
      spos += b * (8 + wsize);
    }
    const uint64_t bl = buckets_for(counts[order - 1], mult);
    tables.emplace_back(spos, bl);
    spos += bl * 16;
    if (spos > m.n) return fail(PPASR_EINVAL, "lm: file shorter than its header says (probing tables)");
    strings_at = spos;
    lm->format = model_type == 1 ? "klm-rest-probing" : "klm-probing";
    lm->kenlm_keys = true;
    size_t total = counts[0];
    for (int n = 2; n <= order; ++n) total += counts[n - 1];
    entries.reserve(total);
    for (uint64_t w = 0; w < counts[0]; ++w) {
      const uint8_t* e = uni + w * wsize;
      entries.push_back(LmEntry{lm_key_from_kenlm(w, 1), neg_abs(rd<float>(e)), rd<float>(e + 4) + 0.f});
    }
    for (int n = 2; n <= order; ++n) {
      const auto& t = tables[n - 2];
      const size_t esz = n == order ? 16 : 8 + wsize;
      uint64_t found = 0;
      for (uint64_t b = 0; b < t.second; ++b) {
        const uint8_t* e = f + t.first + b * esz;
        const uint64_t key = rd<uint64_t>(e);
        if (key == 0) continue;
        ++found;
        entries.push_back(LmEntry{lm_key_from_kenlm(key, n), neg_abs(rd<float>(e + 8)), n == order ? 0.f : rd<float>(e + 12) + 0.f});
      }
      if (found != counts[n - 1])
        return fail(PPASR_EINVAL, "lm: order-" + std::to_string(n) + " table holds " + std::to_string(found) + " n-grams, header says " +
                                      std::to_string(counts[n - 1]));
    }
  } else {
    // ---- trie (plain, quantised and / or Bhiksha-array pointer compression) ----
    const size_t vocab_bytes = 8 + 8 * counts[0];
    size_t spos = pos + vocab_bytes;
    int prob_bits = 0, backoff_bits = 0;
    std::vector<const float*> prob_bins(order + 1, nullptr), backoff_bins(order + 1, nullptr);  // by n-gram order
    if (quant) {
      if (spos + 8 > m.n) return fail(PPASR_EINVAL, "lm: truncated quantiser header");
      const int qv = f[spos];
      prob_bits = f[spos + 1];
      backoff_bits = f[spos + 2];
      if (qv != 2) return fail(PPASR_EUNSUPPORTED, "lm: quantiser version " + std::to_string(qv) + " (this reader knows SeparatelyQuantize version 2)");
      if (prob_bits < 1 || prob_bits > 25 || backoff_bits < 1 || backoff_bits > 25)
        return fail(PPASR_EINVAL, "lm: quantiser bit widths outside 1..25");
      const size_t longest_tab = ((size_t)1 << prob_bits) * 4, middle_tab = ((size_t)1 << backoff_bits) * 4 + longest_tab;
      const size_t qsize = (size_t)(order - 2) * middle_tab + longest_tab + 8;
      if (spos + qsize > m.n) return fail(PPASR_EINVAL, "lm: file shorter than its quantiser tables");
      const float* t = reinterpret_cast<const float*>(f + spos + 8);
      for (int n = 2; n < order; ++n) {
        prob_bins[n] = t;
        t += (size_t)1 << prob_bits;
        backoff_bins[n] = t;
        t += (size_t)1 << backoff_bits;
      }
      prob_bins[order] = t;
      spos += qsize;
    }
    const uint8_t* uni = f + spos;
    spos += (counts[0] + 2) * 16;
    if (spos > m.n) return fail(PPASR_EINVAL, "lm: file shorter than its unigram array");
    const int word_bits = required_bits(counts[0]);
    struct Level {
      const uint8_t* base;       // first bit-packed record
      int total_bits, next_bits; // next_bits = inline bits of the pointer
      uint64_t entries;
      const uint8_t* offsets;    // Bhiksha offset array (unaligned reads), or nullptr
      uint64_t n_offsets;
      int quant_bits;
    };
    std::vector<Level> mid;
    int configured_bits = -1;
    for (int n = 2; n < order; ++n) {
      const uint64_t entries = counts[n - 1], max_next = counts[n], max_offset = entries + 1;
      const int required = required_bits(max_next);
      const int qbits = quant ? prob_bits + backoff_bits : 63;
      Level L{};
      L.entries = entries;
      L.quant_bits = qbits;
      if (array) {
        if (spos + 2 > m.n) return fail(PPASR_EINVAL, "lm: truncated pointer-compression header");
        if (f[spos] != 0) return fail(PPASR_EUNSUPPORTED, "lm: ArrayBhiksha version " + std::to_string((int)f[spos]) + " (this reader knows version 0)");
        if (configured_bits < 0) configured_bits = f[spos + 1];  // (KenLM reads the first block's byte and applies it to all)
        int chop = 0;
        long long lowest = std::numeric_limits<long long>::max();
        for (int c = 0; c <= std::min(required, configured_bits); ++c) {
          const long long change = (long long)(max_next >> (required - c)) * 64 - (long long)max_offset * c;
          if (change < lowest) {
            lowest = change;
            chop = c;
          }
        }
        L.n_offsets = (max_next >> (required - chop)) + 1;
        L.next_bits = required - chop;
        L.offsets = f + align8(spos) + 8;
        const size_t bsize = 8 * (1 + L.n_offsets) + 7;
        L.base = f + spos + bsize;
        L.total_bits = word_bits + qbits + L.next_bits;
        spos += bsize + ((1 + entries) * (uint64_t)L.total_bits + 7) / 8 + 8;
      } else {
        L.next_bits = required;
        L.total_bits = word_bits + qbits + required;
        L.base = f + spos;
        spos += ((1 + entries) * (uint64_t)L.total_bits + 7) / 8 + 8;
      }
      if (spos > m.n) return fail(PPASR_EINVAL, "lm: file shorter than its header says (trie arrays)");
      mid.push_back(L);
    }
    const int lqbits = quant ? prob_bits : 31;
    const int ltotal = word_bits + lqbits;
    Level lng{};
    lng.base = f + spos;
    lng.total_bits = ltotal;
    lng.entries = counts[order - 1];
    lng.quant_bits = lqbits;
    spos += ((1 + counts[order - 1]) * (uint64_t)ltotal + 7) / 8 + 8;
    if (spos > m.n) return fail(PPASR_EINVAL, "lm: file shorter than its header says (trie arrays)");
    strings_at = spos;
    lm->format = quant ? (array ? "klm-quant-array-trie" : "klm-quant-trie") : (array ? "klm-array-trie" : "klm-trie");
    auto bits_to_float = [](uint32_t u) {
      float v;
      std::memcpy(&v, &u, 4);
      return v;
    };
    // `next` pointer of record r of a middle level (r == entries: the end sentinel)
    auto next_of = [&](const Level& L, uint64_t r) -> uint64_t {
      const uint64_t inl = read_bits(L.base, r * (uint64_t)L.total_bits + word_bits + L.quant_bits, L.next_bits);
      if (!L.offsets) return inl;
      // upper_bound(offsets, offsets + n, r) - 1: the last e with offsets[e] <= r
      uint64_t lo = 0, hi = L.n_offsets;
      while (lo < hi) {
        const uint64_t midp = (lo + hi) / 2;
        if (rd<uint64_t>(L.offsets + 8 * midp) <= r) lo = midp + 1;
        else hi = midp;
      }
      const uint64_t e = lo ? lo - 1 : 0;
      return (e << L.next_bits) | inl;
    };
    // depth-first walk: path[0] = last word of the n-gram, path[d] = the word d positions before it
    std::vector<int32_t> path(order), ngram(order);
    struct Frame { uint64_t begin, end; };
    auto emit = [&](int depth, float prob, float backoff) {  // depth = n - 1
      const int n = depth + 1;
      for (int i = 0; i < n; ++i) ngram[i] = path[n - 1 - i];
      entries.push_back(LmEntry{lm_key(ngram.data(), n), prob, backoff});
    };
    std::string walk_err;
    // recursive lambda over the levels
    std::function<void(int, uint64_t, uint64_t)> descend = [&](int depth, uint64_t begin, uint64_t end) {
      // children at `depth` (n-gram order depth + 1 >= 2): records [begin, end) of level depth - 1 in mid / lng
      const bool last = depth == order - 1;
      const Level& L = last ? lng : mid[depth - 1];
      if (end > L.entries || begin > end) {
        walk_err = "lm: trie pointer out of range at order " + std::to_string(depth + 1);
        return;
      }
      for (uint64_t r = begin; r < end && walk_err.empty(); ++r) {
        const uint64_t off = r * (uint64_t)L.total_bits;
        path[depth] = (int32_t)read_bits(L.base, off, word_bits);
        const int n = depth + 1;
        float prob, backoff = 0.f;
        if (quant) {
          // SeparatelyQuantize: [backoff bin][prob bin] in a middle record, [prob bin] in a longest one
          if (last) {
            prob = prob_bins[n][read_bits(L.base, off + word_bits, prob_bits)];
          } else {
            backoff = backoff_bins[n][read_bits(L.base, off + word_bits, backoff_bits)] + 0.f;
            prob = prob_bins[n][read_bits(L.base, off + word_bits + backoff_bits, prob_bits)];
          }
          prob = neg_abs(prob);
        } else {
          prob = neg_abs(bits_to_float((uint32_t)read_bits(L.base, off + word_bits, 31)));
          if (!last) backoff = bits_to_float((uint32_t)read_bits(L.base, off + word_bits + 31, 32)) + 0.f;
        }
        if (last) {
          emit(depth, prob, 0.f);
        } else {
          emit(depth, prob, backoff);
          const uint64_t nb = next_of(L, r), ne = next_of(L, r + 1);
          if (ne < nb) {
            walk_err = "lm: decreasing trie pointers at order " + std::to_string(n);
            return;
          }
          if (ne > nb) descend(depth + 1, nb, ne);
        }
      }
    };
    for (uint64_t w = 0; w < counts[0] && walk_err.empty(); ++w) {
      const uint8_t* e = uni + w * 16;
      path[0] = (int32_t)w;
      emit(0, neg_abs(rd<float>(e)), rd<float>(e + 4) + 0.f);
      const uint64_t nb = rd<uint64_t>(e + 8), ne = rd<uint64_t>(e + 24);
      if (ne > nb) descend(1, nb, ne);
    }
    if (!walk_err.empty()) return fail(PPASR_EINVAL, walk_err);
    size_t total = 0;
    for (int n = 1; n <= order; ++n) total += counts[n - 1];
    if (entries.size() != total)
      return fail(PPASR_EINVAL, "lm: the trie walk found " + std::to_string(entries.size()) + " n-grams, header says " + std::to_string(total));
  }

  // ---- vocabulary strings: "<unk>\0<s>\0..." in index order, to the end of the file ----
  std::unordered_map<std::string, int32_t> words;
  {
    size_t p = strings_at;
    for (uint64_t i = 0; i < counts[0]; ++i) {
      const void* z = p < m.n ? std::memchr(f + p, 0, m.n - p) : nullptr;
      if (!z) return fail(PPASR_EINVAL, "lm: vocabulary strings end before word " + std::to_string(i));
      std::string w(reinterpret_cast<const char*>(f + p), static_cast<const uint8_t*>(z) - (f + p));
      if (i == 0 && w != "<unk>") return fail(PPASR_EINVAL, "lm: the vocabulary strings do not start with <unk> (layout mismatch)");
      if (!words.emplace(w, (int32_t)i).second) return fail(PPASR_EINVAL, "lm: duplicate word in the vocabulary strings: " + w);
      p = static_cast<const uint8_t*>(z) - f + 1;
    }
    if (p != m.n) return fail(PPASR_EINVAL, "lm: " + std::to_string(m.n - p) + " unexplained bytes after the vocabulary strings (layout mismatch)");
  }
  std::string err = lm_bind_vocabulary(*lm, words, vocab_utf8, V);
  if (!err.empty()) return fail(PPASR_EINVAL, err);
  err = lm_build_table(*lm, entries);
  if (!err.empty()) return fail(PPASR_EINVAL, err);
  if (!host_only) {
    ppasr_status s = lm_upload(*lm);
    if (s != PPASR_OK) return s;
  }
  *out = lm.release();
  return PPASR_OK;
}
