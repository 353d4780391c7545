// hashes.cpp — digests behind the crypto syscalls of the host interpreter.
//
// zkir-runtime/src/crypto.rs computes them with third-party crates that are not vendored in the
// reference: sha2 0.10 (Sha256, crypto.rs:247-249), sha3 0.10 (Keccak256 — original Keccak padding,
// crypto.rs:346-348) and blake3 1.5 (blake3::hash, crypto.rs:387).  These are the published
// algorithms (FIPS 180-4; Keccak-f[1600] r=1088 c=512 with pad byte 0x01; BLAKE3 spec, default
// mode), written block-streaming.  Pinned by the digests the reference's tests hold
// (crypto.rs:402-546, crypto_edge_cases.rs:36-127) via tests/test_oracle_kats.py (the [product_host] parametrisation drives these functions through the interpreter's syscalls).
#include "host.h"

namespace zkir {
namespace {

inline uint32_t ror(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline uint32_t le32w(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

// ---------------- SHA-256 ----------------
const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3,
    0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13,
    0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

void sha256_block(uint32_t h[8], const uint8_t blk[64]) {
  uint32_t w[16];
  for (int i = 0; i < 16; i++) w[i] = be32(blk + 4 * i);
  uint32_t s[8];
  memcpy(s, h, 32);
  for (int t = 0; t < 64; t++) {
    if (t >= 16) {   // rolling 16-word schedule window
      const uint32_t w15 = w[(t + 1) & 15], w2 = w[(t + 14) & 15];
      w[t & 15] += (ror(w15, 7) ^ ror(w15, 18) ^ (w15 >> 3)) + w[(t + 9) & 15] + (ror(w2, 17) ^ ror(w2, 19) ^ (w2 >> 10));
    }
    const uint32_t e = s[4], a = s[0];
    const uint32_t t1 = s[7] + (ror(e, 6) ^ ror(e, 11) ^ ror(e, 25)) + ((e & s[5]) ^ (~e & s[6])) + K256[t] + w[t & 15];
    const uint32_t t2 = (ror(a, 2) ^ ror(a, 13) ^ ror(a, 22)) + ((a & s[1]) ^ (a & s[2]) ^ (s[1] & s[2]));
    s[7] = s[6]; s[6] = s[5]; s[5] = s[4]; s[4] = s[3] + t1; s[3] = s[2]; s[2] = s[1]; s[1] = s[0]; s[0] = t1 + t2;
  }
  for (int i = 0; i < 8; i++) h[i] += s[i];
}

// ---------------- Keccak-f[1600] ----------------
const uint64_t KRC[24] = {0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull,
                          0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull,
                          0x0000000080008009ull, 0x000000008000000aull, 0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull,
                          0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
                          0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
const int KROT[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
const int KPIL[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
inline uint64_t rol64(uint64_t x, int n) { return (x << n) | (x >> (64 - n)); }

void keccakf(uint64_t st[25]) {   // rho-pi as the classic in-place lane walk
  for (int round = 0; round < 24; round++) {
    uint64_t bc[5];
    for (int i = 0; i < 5; i++) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
    for (int i = 0; i < 5; i++) {
      const uint64_t t = bc[(i + 4) % 5] ^ rol64(bc[(i + 1) % 5], 1);
      for (int j = 0; j < 25; j += 5) st[j + i] ^= t;
    }
    uint64_t t = st[1];
    for (int i = 0; i < 24; i++) {
      const int j = KPIL[i];
      const uint64_t b = st[j];
      st[j] = rol64(t, KROT[i]);
      t = b;
    }
    for (int j = 0; j < 25; j += 5) {
      for (int i = 0; i < 5; i++) bc[i] = st[j + i];
      for (int i = 0; i < 5; i++) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
    }
    st[0] ^= KRC[round];
  }
}

// ---------------- BLAKE3 ----------------
const uint32_t IV[8] = {0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19};
const uint8_t SIGMA[7][16] = {   // message schedule per round (the fixed permutation applied r times)
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8},
    {3, 4, 10, 12, 13, 2, 7, 14, 6, 5, 9, 0, 11, 15, 8, 1}, {10, 7, 12, 9, 14, 3, 13, 15, 4, 0, 11, 2, 5, 8, 1, 6},
    {12, 13, 9, 11, 15, 10, 14, 8, 7, 2, 5, 3, 0, 1, 6, 4}, {9, 14, 11, 5, 8, 12, 15, 1, 13, 3, 0, 10, 2, 6, 4, 7},
    {11, 15, 5, 0, 1, 9, 8, 6, 14, 10, 2, 12, 3, 4, 7, 13}};
enum : uint32_t { CHUNK_START = 1, CHUNK_END = 2, PARENT = 4, ROOT = 8 };

#define B3G(a, b, c, d, x, y)                         \
  v[a] += v[b] + (x); v[d] = ror(v[d] ^ v[a], 16);    \
  v[c] += v[d];       v[b] = ror(v[b] ^ v[c], 12);    \
  v[a] += v[b] + (y); v[d] = ror(v[d] ^ v[a], 8);     \
  v[c] += v[d];       v[b] = ror(v[b] ^ v[c], 7);

void b3_compress(const uint32_t cv[8], const uint32_t m[16], uint64_t counter, uint32_t blen, uint32_t flags, uint32_t out_cv[8]) {
  uint32_t v[16] = {cv[0], cv[1], cv[2], cv[3], cv[4], cv[5], cv[6], cv[7], IV[0], IV[1], IV[2], IV[3], (uint32_t)counter, (uint32_t)(counter >> 32), blen, flags};
  for (int r = 0; r < 7; r++) {
    const uint8_t* s = SIGMA[r];
    B3G(0, 4, 8, 12, m[s[0]], m[s[1]]) B3G(1, 5, 9, 13, m[s[2]], m[s[3]]) B3G(2, 6, 10, 14, m[s[4]], m[s[5]]) B3G(3, 7, 11, 15, m[s[6]], m[s[7]])
    B3G(0, 5, 10, 15, m[s[8]], m[s[9]]) B3G(1, 6, 11, 12, m[s[10]], m[s[11]]) B3G(2, 7, 8, 13, m[s[12]], m[s[13]]) B3G(3, 4, 9, 14, m[s[14]], m[s[15]])
  }
  for (int i = 0; i < 8; i++) out_cv[i] = v[i] ^ v[i + 8];
}
#undef B3G

}  // namespace

void sha256(const uint8_t* data, size_t len, uint32_t out[8]) {
  uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  size_t off = 0;
  for (; off + 64 <= len; off += 64) sha256_block(h, data + off);
  uint8_t tail[128] = {0};
  const size_t rem = len - off;
  memcpy(tail, data + off, rem);
  tail[rem] = 0x80;
  const size_t tl = rem < 56 ? 64 : 128;
  const uint64_t bits = (uint64_t)len * 8;
  for (int i = 0; i < 8; i++) tail[tl - 1 - i] = (uint8_t)(bits >> (8 * i));
  sha256_block(h, tail);
  if (tl == 128) sha256_block(h, tail + 64);
  memcpy(out, h, 32);
}

void keccak256(const uint8_t* data, size_t len, uint8_t out[32]) {
  constexpr size_t RATE = 136;
  uint64_t st[25] = {0};
  auto absorb = [&](const uint8_t* blk) {
    for (size_t i = 0; i < RATE / 8; i++) { uint64_t lane; memcpy(&lane, blk + 8 * i, 8); st[i] ^= lane; }   // little-endian host
    keccakf(st);
  };
  size_t off = 0;
  for (; off + RATE <= len; off += RATE) absorb(data + off);
  uint8_t last[RATE] = {0};
  memcpy(last, data + off, len - off);
  last[len - off] ^= 0x01;
  last[RATE - 1] ^= 0x80;
  absorb(last);
  memcpy(out, st, 32);
}

void blake3(const uint8_t* data, size_t len, uint8_t out[32]) {
  // Incremental chunk-stack formulation: finish each 1024-byte chunk, push its chaining value, and merge
  // completed subtrees while the chunk count has trailing zero bits; the final merge carries ROOT.
  uint32_t stack[54][8];
  int sp = 0;
  const uint64_t n_chunks = len == 0 ? 1 : (len + 1023) / 1024;
  uint32_t out_words[8];
  for (uint64_t c = 0; c < n_chunks; c++) {
    const uint8_t* p = data + c * 1024;
    const size_t clen = (c == n_chunks - 1) ? len - (size_t)c * 1024 : 1024;
    const size_t nb = clen == 0 ? 1 : (clen + 63) / 64;
    const bool only_chunk = n_chunks == 1;
    uint32_t cv[8];
    memcpy(cv, IV, 32);
    for (size_t b = 0; b < nb; b++) {
      uint8_t blk[64] = {0};
      const size_t bl = clen == 0 ? 0 : ((b == nb - 1) ? clen - b * 64 : 64);
      memcpy(blk, p + b * 64, bl);
      uint32_t m[16];
      for (int i = 0; i < 16; i++) m[i] = le32w(blk + 4 * i);
      uint32_t fl = (b == 0 ? CHUNK_START : 0) | (b == nb - 1 ? CHUNK_END : 0);
      if (only_chunk && b == nb - 1) fl |= ROOT;
      uint32_t nx[8];
      b3_compress(cv, m, c, (uint32_t)bl, fl, nx);
      memcpy(cv, nx, 32);
    }
    if (only_chunk) { memcpy(out_words, cv, 32); break; }
    memcpy(stack[sp++], cv, 32);
    if (c + 1 < n_chunks) {
      uint64_t total = c + 1;                      // merge while the number of chunks so far is even
      while ((total & 1) == 0) {
        uint32_t m[16], nx[8];
        memcpy(m, stack[sp - 2], 32); memcpy(m + 8, stack[sp - 1], 32);
        b3_compress(IV, m, 0, 64, PARENT, nx);
        sp -= 2;
        memcpy(stack[sp++], nx, 32);
        total >>= 1;
      }
    } else {
      while (sp > 1) {                             // final right-to-left fold; the last merge is the root
        uint32_t m[16], nx[8];
        memcpy(m, stack[sp - 2], 32); memcpy(m + 8, stack[sp - 1], 32);
        b3_compress(IV, m, 0, 64, PARENT | (sp == 2 ? ROOT : 0), nx);
        sp -= 2;
        memcpy(stack[sp++], nx, 32);
      }
      memcpy(out_words, stack[0], 32);
    }
  }
  for (int i = 0; i < 32; i++) out[i] = (uint8_t)(out_words[i / 4] >> (8 * (i % 4)));
}

}  // namespace zkir
