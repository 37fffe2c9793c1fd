// spartan_amd host driver: the zlib stream of R1CSShape::get_digest (src/r1cs.rs:154-158).
//
// The reference binds a NIZK proof to its instance by absorbing `digest = ZlibEncoder(Compression::default())(bincode(shape))` into
// the transcript (src/lib.rs:514): the digest is the COMPRESSED STREAM ITSELF, so byte parity needs the exact deflater.
// flate2 = "1.0.14" with the rust_backend (Cargo.toml:31,75) is miniz_oxide, a line-by-line port of miniz's tdefl; neither is under
// /root/reference. This file restates tdefl's level-6 path (what Compression::default() selects) from the published algorithm:
//   * parameters (tdefl_create_comp_flags_from_zip_params(6, 15, 0)): 128 hash-chain probes (43 once a match of >= 32 is held),
//     lazy parsing (greedy only up to level 3), zlib header, Adler-32 trailer;
//   * dictionary 32 KiB, 3-byte hash (b0 << 10 ^ b1 << 5 ^ b2) & 32767 with 16-bit position chains, matches 3..258, a length-3 match
//     further than 8 KiB away is dropped, a held match of >= 128 is taken at once;
//   * a block ends when the 64 KiB LZ code buffer is nearly full or, past 31 KiB of input, when the code buffer stops paying
//     (code bytes * 115 / 128 >= input bytes); each block is dynamic-Huffman (static below 48 input bytes), or stored when the
//     coded form is not smaller; code lengths by the in-place minimum-redundancy algorithm over the frequency-sorted symbols
//     (stable: ties by symbol index), limited to 15 / 7 bits by the Kraft fix-up, code-length alphabet run-length packed as tdefl does.
// PINNED AGAINST THE REAL C MINIZ (tests/test_host_transcript.py::test_deflater_equals_real_miniz): libtorch_cpu.so in this image bundles
// miniz 3.0.2 and exports mz_compress2; at every lazy-parsing level (4..10, i.e. 16/32/128/256/512/768/1500 probes — level 6 is what
// flate2's Compression::default() selects) the WHOLE zlib stream of this file, header and Adler-32 included, equals miniz's on the bincode
// of synthetic R1CS shapes 2^4..2^16 (multi-block, dynamic Huffman), random data (stored blocks), text, and block-boundary inputs. The
// stream also inflates (Python zlib) to exactly the bincode the oracle serialises. What stays unpinned is only that miniz_oxide (a
// line-by-line Rust port of tdefl) equals miniz; scripts/compare_with_libspartan.sh prints both digests. The header bytes are the one
// documented variable: miniz >= 2.2 and miniz_oxide >= 0.4 derive FLEVEL from the probe count (level 6: 0x78 0x9C), older ones write
// 0x78 0x01 (the `old_header` argument — an explicit API parameter, spz_instance_set_digest_header, never an environment switch).
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

namespace spz {
namespace {
constexpr unsigned DICT = 32768, DMASK = DICT - 1, MINM = 3, MAXM = 258, LZBUF = 64 * 1024, HBITS = 15, HSHIFT = 5, HSIZE = 1u << HBITS;

struct SymFreq { uint16_t key, sym; };

struct Tdefl {
  std::vector<uint8_t> out;
  uint64_t bitbuf = 0;
  unsigned bits_in = 0;
  uint8_t dict[DICT + MAXM - 1];
  uint16_t next[DICT], hash[HSIZE];
  // LZ codes of the current block: tokens + the byte count tdefl's code buffer would hold (flag bytes included)
  struct Tok { uint16_t len_m3, dist_m1; bool match; uint8_t lit; };
  std::vector<Tok> toks;
  size_t code_bytes = 1;       // pLZ_code_buf - lz_code_buf (starts past the first flag byte)
  unsigned flags_left = 8;
  uint16_t count[3][288];
  uint16_t codes[3][288];
  uint8_t sizes[3][288];
  unsigned lookahead_pos = 0, lookahead_size = 0, dict_size = 0, total_lz_bytes = 0, lz_dict_pos = 0, block_index = 0;
  unsigned saved_match_dist = 0, saved_match_len = 0, saved_lit = 0;
  unsigned max_probes[2];
  uint32_t adler = 1;
  bool old_header = false;
  unsigned probes = 128;  // tdefl flags & 0xFFF: s_tdefl_num_probes[level] = {0, 1, 6, 32, 16, 32, 128, 256, 512, 768, 1500}; 128 = level 6

  void put(unsigned b, unsigned l) {
    bitbuf |= (uint64_t)b << bits_in;
    bits_in += l;
    while (bits_in >= 8) { out.push_back((uint8_t)bitbuf); bitbuf >>= 8; bits_in -= 8; }
  }
  // ---- symbol tables of RFC 1951 in tdefl's indexing: length - 3 -> (symbol, extra bits); distance - 1 -> (symbol, extra bits)
  static void len_code(unsigned len_m3, unsigned* sym, unsigned* nextra) {
    static const uint16_t base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    unsigned len = len_m3 + 3, k = 28;
    while (base[k] > len) k--;
    *sym = 257 + k; *nextra = extra[k];
  }
  static void dist_code(unsigned dist_m1, unsigned* sym, unsigned* nextra) {
    static const uint16_t base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    unsigned dist = dist_m1 + 1, k = 29;
    while (base[k] > dist) k--;
    *sym = k; *nextra = extra[k];
  }

  void record_literal(uint8_t lit) {
    total_lz_bytes++;
    toks.push_back(Tok{0, 0, false, lit});
    code_bytes += 1;
    if (--flags_left == 0) { flags_left = 8; code_bytes++; }
    count[0][lit]++;
  }
  void record_match(unsigned len, unsigned dist) {
    total_lz_bytes += len;
    toks.push_back(Tok{(uint16_t)(len - MINM), (uint16_t)(dist - 1), true, 0});
    code_bytes += 3;
    if (--flags_left == 0) { flags_left = 8; code_bytes++; }
    unsigned s, e;
    dist_code(dist - 1, &s, &e);
    count[1][s]++;
    len_code(len - MINM, &s, &e);
    count[0][s]++;
  }

  // ---- Huffman tables (tdefl_optimize_huffman_table)
  static void minimum_redundancy(SymFreq* A, int n) {  // in-place: keys become code lengths (frequency-sorted input)
    int root, leaf, next, avbl, used, dpth;
    if (n == 0) return;
    if (n == 1) { A[0].key = 1; return; }
    A[0].key = (uint16_t)(A[0].key + A[1].key); root = 0; leaf = 2;
    for (next = 1; next < n - 1; next++) {
      if (leaf >= n || A[root].key < A[leaf].key) { A[next].key = A[root].key; A[root++].key = (uint16_t)next; } else A[next].key = A[leaf++].key;
      if (leaf >= n || (root < next && A[root].key < A[leaf].key)) { A[next].key = (uint16_t)(A[next].key + A[root].key); A[root++].key = (uint16_t)next; }
      else A[next].key = (uint16_t)(A[next].key + A[leaf++].key);
    }
    A[n - 2].key = 0;
    for (next = n - 3; next >= 0; next--) A[next].key = (uint16_t)(A[A[next].key].key + 1);
    avbl = 1; used = dpth = 0; root = n - 2; next = n - 1;
    while (avbl > 0) {
      while (root >= 0 && (int)A[root].key == dpth) { used++; root--; }
      while (avbl > used) { A[next--].key = (uint16_t)dpth; avbl--; }
      avbl = 2 * used; dpth++; used = 0;
    }
  }
  static void enforce_max_code_size(int* num_codes, int code_list_len, int max_code_size) {
    if (code_list_len <= 1) return;
    for (int i = max_code_size + 1; i <= 32; i++) num_codes[max_code_size] += num_codes[i];
    uint32_t total = 0;
    for (int i = max_code_size; i > 0; i--) total += ((uint32_t)num_codes[i]) << (max_code_size - i);
    while (total != (1u << max_code_size)) {
      num_codes[max_code_size]--;
      for (int i = max_code_size - 1; i > 0; i--)
        if (num_codes[i]) { num_codes[i]--; num_codes[i + 1] += 2; break; }
      total--;
    }
  }
  void optimize_table(int t, int table_len, int limit, bool is_static) {
    int num_codes[1 + 32];
    unsigned next_code[32 + 1];
    memset(num_codes, 0, sizeof num_codes);
    if (is_static) {
      for (int i = 0; i < table_len; i++) num_codes[sizes[t][i]]++;
    } else {
      SymFreq syms[288];
      int n = 0;
      for (int i = 0; i < table_len; i++)
        if (count[t][i]) { syms[n].key = count[t][i]; syms[n++].sym = (uint16_t)i; }
      std::stable_sort(syms, syms + n, [](const SymFreq& a, const SymFreq& b) { return a.key < b.key; });  // tdefl's two-pass LSD radix sort
      minimum_redundancy(syms, n);
      for (int i = 0; i < n; i++) num_codes[syms[i].key]++;
      enforce_max_code_size(num_codes, n, limit);
      memset(sizes[t], 0, sizeof sizes[t]);
      memset(codes[t], 0, sizeof codes[t]);
      for (int i = 1, j = n; i <= limit; i++)
        for (int l = num_codes[i]; l > 0; l--) sizes[t][syms[--j].sym] = (uint8_t)i;
    }
    next_code[1] = 0;
    for (int j = 0, i = 2; i <= limit; i++) next_code[i] = j = ((j + num_codes[i - 1]) << 1);
    for (int i = 0; i < table_len; i++) {
      unsigned rev = 0, code, sz = sizes[t][i];
      if (!sz) continue;
      code = next_code[sz]++;
      for (unsigned l = sz; l > 0; l--, code >>= 1) rev = (rev << 1) | (code & 1);
      codes[t][i] = (uint16_t)rev;
    }
  }
  void start_static_block() {
    uint8_t* p = sizes[0];
    int i = 0;
    for (; i <= 143; ++i) p[i] = 8;
    for (; i <= 255; ++i) p[i] = 9;
    for (; i <= 279; ++i) p[i] = 7;
    for (; i <= 287; ++i) p[i] = 8;
    memset(sizes[1], 5, 32);
    optimize_table(0, 288, 15, true);
    optimize_table(1, 32, 15, true);
    put(1, 2);
  }
  void start_dynamic_block() {
    static const uint8_t swizzle[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint8_t to_pack[288 + 32], packed[288 + 32], prev = 0xFF;
    unsigned npacked = 0, rle_z = 0, rle_rep = 0;
    count[0][256] = 1;
    optimize_table(0, 288, 15, false);
    optimize_table(1, 32, 15, false);
    int nlit, ndist, nbl;
    for (nlit = 286; nlit > 257; nlit--) if (sizes[0][nlit - 1]) break;
    for (ndist = 30; ndist > 1; ndist--) if (sizes[1][ndist - 1]) break;
    memcpy(to_pack, sizes[0], nlit);
    memcpy(to_pack + nlit, sizes[1], ndist);
    unsigned total = nlit + ndist;
    memset(count[2], 0, sizeof(uint16_t) * 19);
    auto rle_prev = [&]() {
      if (rle_rep) {
        if (rle_rep < 3) { count[2][prev] = (uint16_t)(count[2][prev] + rle_rep); while (rle_rep--) packed[npacked++] = prev; }
        else { count[2][16]++; packed[npacked++] = 16; packed[npacked++] = (uint8_t)(rle_rep - 3); }
        rle_rep = 0;
      }
    };
    auto rle_zero = [&]() {
      if (rle_z) {
        if (rle_z < 3) { count[2][0] = (uint16_t)(count[2][0] + rle_z); while (rle_z--) packed[npacked++] = 0; }
        else if (rle_z <= 10) { count[2][17]++; packed[npacked++] = 17; packed[npacked++] = (uint8_t)(rle_z - 3); }
        else { count[2][18]++; packed[npacked++] = 18; packed[npacked++] = (uint8_t)(rle_z - 11); }
        rle_z = 0;
      }
    };
    for (unsigned i = 0; i < total; i++) {
      uint8_t cs = to_pack[i];
      if (!cs) {
        rle_prev();
        if (++rle_z == 138) rle_zero();
      } else {
        rle_zero();
        if (cs != prev) { rle_prev(); count[2][cs]++; packed[npacked++] = cs; }
        else if (++rle_rep == 6) rle_prev();
      }
      prev = cs;
    }
    if (rle_rep) rle_prev(); else rle_zero();
    optimize_table(2, 19, 7, false);
    put(2, 2);
    put(nlit - 257, 5);
    put(ndist - 1, 5);
    for (nbl = 18; nbl >= 0; nbl--) if (sizes[2][swizzle[nbl]]) break;
    nbl = std::max(4, nbl + 1);
    put(nbl - 4, 4);
    for (int i = 0; i < nbl; i++) put(sizes[2][swizzle[i]], 3);
    for (unsigned k = 0; k < npacked;) {
      unsigned code = packed[k++];
      put(codes[2][code], sizes[2][code]);
      if (code >= 16) put(packed[k++], code == 16 ? 2 : (code == 17 ? 3 : 7));
    }
  }
  void compress_lz_codes() {
    for (const Tok& t : toks) {
      if (t.match) {
        unsigned s, e;
        len_code(t.len_m3, &s, &e);
        static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        put(codes[0][s], sizes[0][s]);
        put((t.len_m3 + 3) - lbase[s - 257], e);   // == match_len & mask(extra) in tdefl's table form
        dist_code(t.dist_m1, &s, &e);
        static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
        put(codes[1][s], sizes[1][s]);
        put((t.dist_m1 + 1) - dbase[s], e);
      } else {
        put(codes[0][t.lit], sizes[0][t.lit]);
      }
    }
    put(codes[0][256], sizes[0][256]);
  }
  void flush_block(bool finish) {
    // (the flag byte that was opened but holds no code is given back: code_bytes -= (flags_left == 8) — nothing reads it afterwards)
    if (block_index == 0) {
      // miniz >= 2.2 / miniz_oxide >= 0.4: FLEVEL recovered from the probe count (index in s_tdefl_num_probes), FCHECK makes the 16 bits a multiple of 31
      static const unsigned num_probes[11] = {0, 1, 6, 32, 16, 32, 128, 256, 512, 768, 1500};
      unsigned i = 0, flevel = 3;
      for (; i < 11; i++) if (num_probes[i] == probes) break;
      if (i < 2) flevel = 0; else if (i < 6) flevel = 1; else if (i == 6) flevel = 2;
      unsigned header = (0x78u << 8) | (flevel << 6);
      header += 31 - (header % 31);
      put(0x78, 8); put(old_header ? 0x01 : (header & 0xFF), 8);
    }
    put(finish ? 1 : 0, 1);
    const size_t saved_out = out.size();
    const uint64_t saved_buf = bitbuf;
    const unsigned saved_bits = bits_in;
    if (total_lz_bytes < 48) start_static_block(); else start_dynamic_block();
    compress_lz_codes();
    // if the block got expanded, forget it and send a stored block instead (the data is still in the dictionary)
    if (total_lz_bytes && (out.size() - saved_out + 1u) >= total_lz_bytes && (lookahead_pos - lz_dict_pos) <= dict_size) {
      out.resize(saved_out); bitbuf = saved_buf; bits_in = saved_bits;
      put(0, 2);
      if (bits_in) put(0, 8 - bits_in);
      put(total_lz_bytes & 0xFFFF, 16);
      put((total_lz_bytes ^ 0xFFFF) & 0xFFFF, 16);
      for (unsigned i = 0; i < total_lz_bytes; ++i) put(dict[(lz_dict_pos + i) & DMASK], 8);
    }
    if (finish) {
      if (bits_in) put(0, 8 - bits_in);
      uint32_t a = adler;
      for (int i = 0; i < 4; i++) { put((a >> 24) & 0xFF, 8); a <<= 8; }
    }
    memset(count[0], 0, sizeof count[0]);
    memset(count[1], 0, sizeof count[1]);
    toks.clear();
    code_bytes = 1; flags_left = 8;
    lz_dict_pos += total_lz_bytes;
    total_lz_bytes = 0;
    block_index++;
  }
  void find_match(unsigned lpos, unsigned max_dist, unsigned max_match_len, unsigned* pdist, unsigned* plen) {
    unsigned dist = 0, pos = lpos & DMASK, match_len = *plen, probe_pos = pos, next_probe_pos, probe_len;
    unsigned probes_left = max_probes[match_len >= 32];
    if (max_match_len <= match_len) return;
    uint8_t c0 = dict[pos + match_len], c1 = dict[pos + match_len - 1];
    for (;;) {
      for (;;) {
        if (--probes_left == 0) return;
        bool hit = false;
        for (int k = 0; k < 3 && !hit; k++) {
          next_probe_pos = next[probe_pos];
          if (!next_probe_pos || (dist = (uint16_t)(lpos - next_probe_pos)) > max_dist) return;
          probe_pos = next_probe_pos & DMASK;
          if (dict[probe_pos + match_len] == c0 && dict[probe_pos + match_len - 1] == c1) hit = true;
        }
        if (hit) break;
      }
      if (!dist) break;
      const uint8_t *p = dict + pos, *q = dict + probe_pos;
      for (probe_len = 0; probe_len < max_match_len; probe_len++) if (*p++ != *q++) break;
      if (probe_len > match_len) {
        *pdist = dist;
        if ((*plen = match_len = probe_len) == max_match_len) return;
        c0 = dict[pos + match_len]; c1 = dict[pos + match_len - 1];
      }
    }
  }
  void run(const uint8_t* src, size_t n) {
    memset(dict, 0, sizeof dict); memset(next, 0, sizeof next); memset(hash, 0, sizeof hash); memset(count, 0, sizeof count);
    memset(codes, 0, sizeof codes); memset(sizes, 0, sizeof sizes);
    const unsigned flags = probes;  // lazy parsing (levels >= 4; the greedy parser of levels 1..3 is not restated)
    max_probes[0] = 1 + ((flags & 0xFFF) + 2) / 3;
    max_probes[1] = 1 + (((flags & 0xFFF) >> 2) + 2) / 3;
    // Adler-32 of the whole input (tdefl updates it per call; the value at the end is the same)
    {
      uint32_t s1 = 1, s2 = 0;
      for (size_t i = 0; i < n;) {
        size_t blk = std::min<size_t>(5552, n - i);
        for (size_t k = 0; k < blk; k++) { s1 += src[i + k]; s2 += s1; }
        s1 %= 65521; s2 %= 65521; i += blk;
      }
      adler = (s2 << 16) | s1;
    }
    size_t left = n;
    while (left || lookahead_size) {
      // fill the lookahead, inserting every 3-byte string into the hash chains as its last byte arrives
      while (left && lookahead_size < MAXM) {
        uint8_t c = *src++;
        left--;
        unsigned dst = (lookahead_pos + lookahead_size) & DMASK;
        dict[dst] = c;
        if (dst < MAXM - 1) dict[DICT + dst] = c;
        if (++lookahead_size + dict_size >= MINM) {
          unsigned ins = lookahead_pos + (lookahead_size - 1) - 2;
          unsigned h = ((dict[ins & DMASK] << (HSHIFT * 2)) ^ (dict[(ins + 1) & DMASK] << HSHIFT) ^ c) & (HSIZE - 1);
          next[ins & DMASK] = hash[h];
          hash[h] = (uint16_t)ins;
        }
      }
      dict_size = std::min(DICT - lookahead_size, dict_size);
      if (left == 0 && lookahead_size == 0) break;
      // (no-flush calls stop here while the lookahead is short; with the whole input in hand the lookahead is full until the tail)
      unsigned len_to_move = 1, cur_match_dist = 0, cur_match_len = saved_match_len ? saved_match_len : (MINM - 1), cur_pos = lookahead_pos & DMASK;
      find_match(lookahead_pos, dict_size, lookahead_size, &cur_match_dist, &cur_match_len);
      if ((cur_match_len == MINM && cur_match_dist >= 8u * 1024u) || cur_pos == cur_match_dist) cur_match_dist = cur_match_len = 0;
      if (saved_match_len) {
        if (cur_match_len > saved_match_len) {
          record_literal((uint8_t)saved_lit);
          if (cur_match_len >= 128) { record_match(cur_match_len, cur_match_dist); saved_match_len = 0; len_to_move = cur_match_len; }
          else { saved_lit = dict[cur_pos]; saved_match_dist = cur_match_dist; saved_match_len = cur_match_len; }
        } else {
          record_match(saved_match_len, saved_match_dist);
          len_to_move = saved_match_len - 1;
          saved_match_len = 0;
        }
      } else if (!cur_match_dist) {
        record_literal(dict[cur_pos]);
      } else if (cur_match_len >= 128) {
        record_match(cur_match_len, cur_match_dist);
        len_to_move = cur_match_len;
      } else {
        saved_lit = dict[cur_pos]; saved_match_dist = cur_match_dist; saved_match_len = cur_match_len;
      }
      lookahead_pos += len_to_move;
      lookahead_size -= len_to_move;
      dict_size = std::min(dict_size + len_to_move, DICT);
      if (code_bytes > LZBUF - 8 || (total_lz_bytes > 31 * 1024 && ((code_bytes * 115) >> 7) >= total_lz_bytes)) flush_block(false);
    }
    // finish: a match still held is emitted by the parser above before the lookahead runs dry (its len_to_move consumes it)
    flush_block(true);
  }
};
}  // namespace

std::vector<uint8_t> zlib_miniz_probes(const uint8_t* data, size_t n, unsigned probes, bool old_header) {
  std::unique_ptr<Tdefl> d(new Tdefl());
  d->old_header = old_header;
  d->probes = probes & 0xFFF;
  d->run(data, n);
  return std::move(d->out);
}
std::vector<uint8_t> zlib_level6_miniz(const uint8_t* data, size_t n, bool old_header) { return zlib_miniz_probes(data, n, 128, old_header); }

}  // namespace spz
