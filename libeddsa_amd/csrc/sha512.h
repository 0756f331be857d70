// sha512.h - SHA-512 on the device, one message per lane.
//
// Replaces the reference's lib/sha512.c (compress: sha512.c:83-124; padding and the 128-bit length
// field: sha512.c:176-210) for the three hashes of the Ed25519 path:
//   key setup   SHA-512(sk)                 ed25519-sha512.c:31-47
//   nonce       SHA-512(h[32..64) || M)     ed25519-sha512.c:101-106
//   challenge   SHA-512(R || A || M)        ed25519-sha512.c:113-118 and :166-171
// i.e. always "a prefix of NPRE little-endian 32-bit words held in registers, then a message in
// global memory".  All 80 rounds are unrolled so the 16-word schedule ring stays in registers.
#pragma once
#include "fe25519.h"

namespace ed {

// (a constexpr function: with the 80 rounds unrolled every constant becomes an instruction literal;
// as a __constant__ array the 80 loads were hoisted and held ~160 registers)
ED_DEV constexpr uint64_t sha512_k(int r) {
  constexpr uint64_t SHA512_K[80] = {
    0x428a2f98d728ae22ULL, 0x7137449123ef65cdULL, 0xb5c0fbcfec4d3b2fULL, 0xe9b5dba58189dbbcULL,
    0x3956c25bf348b538ULL, 0x59f111f1b605d019ULL, 0x923f82a4af194f9bULL, 0xab1c5ed5da6d8118ULL,
    0xd807aa98a3030242ULL, 0x12835b0145706fbeULL, 0x243185be4ee4b28cULL, 0x550c7dc3d5ffb4e2ULL,
    0x72be5d74f27b896fULL, 0x80deb1fe3b1696b1ULL, 0x9bdc06a725c71235ULL, 0xc19bf174cf692694ULL,
    0xe49b69c19ef14ad2ULL, 0xefbe4786384f25e3ULL, 0x0fc19dc68b8cd5b5ULL, 0x240ca1cc77ac9c65ULL,
    0x2de92c6f592b0275ULL, 0x4a7484aa6ea6e483ULL, 0x5cb0a9dcbd41fbd4ULL, 0x76f988da831153b5ULL,
    0x983e5152ee66dfabULL, 0xa831c66d2db43210ULL, 0xb00327c898fb213fULL, 0xbf597fc7beef0ee4ULL,
    0xc6e00bf33da88fc2ULL, 0xd5a79147930aa725ULL, 0x06ca6351e003826fULL, 0x142929670a0e6e70ULL,
    0x27b70a8546d22ffcULL, 0x2e1b21385c26c926ULL, 0x4d2c6dfc5ac42aedULL, 0x53380d139d95b3dfULL,
    0x650a73548baf63deULL, 0x766a0abb3c77b2a8ULL, 0x81c2c92e47edaee6ULL, 0x92722c851482353bULL,
    0xa2bfe8a14cf10364ULL, 0xa81a664bbc423001ULL, 0xc24b8b70d0f89791ULL, 0xc76c51a30654be30ULL,
    0xd192e819d6ef5218ULL, 0xd69906245565a910ULL, 0xf40e35855771202aULL, 0x106aa07032bbd1b8ULL,
    0x19a4c116b8d2d0c8ULL, 0x1e376c085141ab53ULL, 0x2748774cdf8eeb99ULL, 0x34b0bcb5e19b48a8ULL,
    0x391c0cb3c5c95a63ULL, 0x4ed8aa4ae3418acbULL, 0x5b9cca4f7763e373ULL, 0x682e6ff3d6b2b8a3ULL,
    0x748f82ee5defb2fcULL, 0x78a5636f43172f60ULL, 0x84c87814a1f0ab72ULL, 0x8cc702081a6439ecULL,
    0x90befffa23631e28ULL, 0xa4506cebde82bde9ULL, 0xbef9a3f7b2c67915ULL, 0xc67178f2e372532bULL,
    0xca273eceea26619cULL, 0xd186b8c721c0c207ULL, 0xeada7dd6cde0eb1eULL, 0xf57d4f7fee6ed178ULL,
    0x06f067aa72176fbaULL, 0x0a637dc5a2c898a6ULL, 0x113f9804bef90daeULL, 0x1b710b35131c471bULL,
    0x28db77f523047d84ULL, 0x32caab7b40c72493ULL, 0x3c9ebe0a15c9bebcULL, 0x431d67c49c100d4cULL,
    0x4cc5d4becb3e42b6ULL, 0x597f299cfc657e2aULL, 0x5fcb6fab3ad6faecULL, 0x6c44198c4a475817ULL};
  return SHA512_K[r];
}

// 64-bit rotation, n a constant in 1..63 other than 32.  On the device: two v_alignbit_b32 (hipcc expands the portable
// form into 64-bit shifts and an OR, 3.2 instructions per rotation on average; a compression has 736 rotations)
ED_DEV uint64_t rotr64(uint64_t x, int n) {
#ifdef ED_HOST_CHECK
  return (x >> n) | (x << (64 - n));
#else
  const uint32_t lo = (uint32_t)(n < 32 ? x : x >> 32), hi = (uint32_t)(n < 32 ? x >> 32 : x);
  const uint32_t rl = __builtin_amdgcn_alignbit(hi, lo, (uint32_t)(n & 31)), rh = __builtin_amdgcn_alignbit(lo, hi, (uint32_t)(n & 31));
  return ((uint64_t)rh << 32) | rl;
#endif
}

// Maj(a, b, c) = (a & b) ^ (a & c) ^ (b & c): c where a and b differ, b where they agree - a bit select under a ^ b.
// hipcc turns every C spelling of it back into two ANDs and two XORs per half; v_bfi_b32 is one.
ED_DEV uint64_t sha512_maj(uint64_t a, uint64_t b, uint64_t c) {
#ifdef ED_HOST_CHECK
  return (a & b) ^ (a & c) ^ (b & c);
#else
  const uint64_t x = a ^ b;
  uint32_t lo, hi;
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(lo) : "v"((uint32_t)x), "v"((uint32_t)c), "v"((uint32_t)b));
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(hi) : "v"((uint32_t)(x >> 32)), "v"((uint32_t)(c >> 32)), "v"((uint32_t)(b >> 32)));
  return ((uint64_t)hi << 32) | lo;
#endif
}

// sha512.c:83-124 compress, one round.  The NAMES a..h rotate over s[] instead of the values (round R's a is
// s[(80 - R) & 7]), and the rounds are instantiated one by one rather than unrolled from a loop: with the intrinsic
// rotations the body of an 80-round loop is past the size up to which hipcc honours `#pragma unroll`, and the rolled
// loop it then emits indexes w[] and the constants through s_set_gpr_idx.
template <int R>
ED_DEV void sha512_round(uint64_t (&s)[8], uint64_t (&w)[16]) {
  constexpr int o = (80 - R) & 7;
  const uint64_t a = s[o], b = s[(o + 1) & 7], c = s[(o + 2) & 7], e = s[(o + 4) & 7], f = s[(o + 5) & 7], g = s[(o + 6) & 7];
  if (R >= 16) {
    const uint64_t w15 = w[(R + 1) & 15], w2 = w[(R + 14) & 15];
    w[R & 15] += (rotr64(w15, 1) ^ rotr64(w15, 8) ^ (w15 >> 7)) + w[(R + 9) & 15] +
                 (rotr64(w2, 19) ^ rotr64(w2, 61) ^ (w2 >> 6));
  }
  const uint64_t t1 = s[(o + 7) & 7] + (rotr64(e, 14) ^ rotr64(e, 18) ^ rotr64(e, 41)) + ((e & f) ^ (~e & g)) +
                      sha512_k(R) + w[R & 15];
  const uint64_t t2 = (rotr64(a, 28) ^ rotr64(a, 34) ^ rotr64(a, 39)) + sha512_maj(a, b, c);
  s[(o + 3) & 7] += t1;                          // e of the next round
  s[(o + 7) & 7] = t1 + t2;                      // a of the next round
}
template <int R>
ED_DEV void sha512_rounds_from(uint64_t (&s)[8], uint64_t (&w)[16]) {
  if constexpr (R < 80) {
    sha512_round<R>(s, w);
    sha512_rounds_from<R + 1>(s, w);
  }
}

// sha512.c:83-124 compress; w[] is consumed (used as the schedule ring)
ED_DEV void sha512_compress(uint64_t st[8], uint64_t w[16]) {
  uint64_t s[8], (&wr)[16] = *reinterpret_cast<uint64_t (*)[16]>(w);
#pragma unroll
  for (int i = 0; i < 8; i++) s[i] = st[i];
  sha512_rounds_from<0>(s, wr);
#pragma unroll
  for (int i = 0; i < 8; i++) st[i] += s[i];
}

ED_DEV uint32_t bswap32(uint32_t x) { return __builtin_bswap32(x); }

// little-endian 32-bit word number idx of the padded message tail: bytes of msg, then 0x80, then 0
ED_DEV uint32_t sha_msg_word(const uint8_t* msg, size_t len, bool aligned, size_t idx) {
  const size_t off = 4 * idx;
  if (off + 4 <= len) {
    if (aligned) return *reinterpret_cast<const uint32_t*>(msg + off);
    return (uint32_t)msg[off] | ((uint32_t)msg[off + 1] << 8) | ((uint32_t)msg[off + 2] << 16) |
           ((uint32_t)msg[off + 3] << 24);
  }
  uint32_t v = 0;
#pragma unroll
  for (int t = 0; t < 4; t++) {
    const size_t pos = off + t;
    const uint32_t byte = pos < len ? (uint32_t)msg[pos] : (pos == len ? 0x80u : 0u);
    v |= byte << (8 * t);
  }
  return v;
}

// The sixteen big-endian 64-bit words of a block that lies wholly inside the message, from a pointer of ANY alignment: ragged
// messages start wherever the previous one ended.  The general path above assembles an unaligned word from four byte loads
// behind a bounds check per word - 128 byte loads and 32 branches per block; here 33 aligned loads and one v_perm_b32 per word,
// which shifts and swaps in one go (its selector is the pointer's low two bits).  The aligned words touched hold at least one
// byte of the block each, so nothing outside the message's own words is read.
ED_DEV void sha_full_block_words(uint64_t w[16], const uint8_t* p) {
#ifdef ED_HOST_CHECK
#pragma unroll
  for (int k = 0; k < 16; k++) {
    uint64_t v = 0;
    for (int t = 0; t < 8; t++) v = (v << 8) | p[8 * k + t];
    w[k] = v;
  }
#else
  const uint32_t a = (uint32_t)(reinterpret_cast<uintptr_t>(p) & 3);
  const uint32_t* q = reinterpret_cast<const uint32_t*>(p - a);
  uint32_t x[33];
#pragma unroll
  for (int j = 0; j < 32; j++) x[j] = q[j];
  x[32] = q[a ? 32 : 31];                         // (aligned: the 33rd word is not the block's and is not touched)
  // v_perm_b32 D, S0, S1, sel: byte i of D = byte sel[i] of the eight bytes S0:S1 (0-3: S1).  The message word at byte a of
  // x[j+1]:x[j], most significant byte first
  const uint32_t sel = (a + 3) | ((a + 2) << 8) | ((a + 1) << 16) | (a << 24);
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const uint32_t hi = __builtin_amdgcn_perm(x[2 * k + 1], x[2 * k], sel), lo = __builtin_amdgcn_perm(x[2 * k + 2], x[2 * k + 1], sel);
    w[k] = ((uint64_t)hi << 32) | lo;
  }
#endif
}

// out[16] (little-endian words of the 64-byte digest) = SHA-512(pre[0..NPRE) || msg[0..len)).
// NPRE is 0, 8 or 16 (so NPRE words never straddle the first block).
template <int NPRE>
ED_DEV void sha512_prefix_msg(uint32_t out[16], const uint32_t* pre, const uint8_t* msg, size_t len) {
  uint64_t st[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL,
                    0xa54ff53a5f1d36f1ULL, 0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL,
                    0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
  const size_t total = 4 * (size_t)NPRE + len;
  const size_t nblk = (total + 17 + 127) >> 7;
  const bool aligned = (reinterpret_cast<uintptr_t>(msg) & 3) == 0;
  uint64_t w[16];
  for (size_t b = 0; b < nblk; b++) {
    // a block wholly inside the message (every block of a long message but its first, when there is a prefix, and its last
    // one or two): no per-word bounds, any alignment
    if ((NPRE == 0 || b > 0) && 128 * (b + 1) - 4 * (size_t)NPRE <= len) {
      sha_full_block_words(w, msg + (128 * b - 4 * (size_t)NPRE));
    } else {
#pragma unroll
      for (int k = 0; k < 16; k++) {
        uint32_t lo32, hi32;   // stream words 2k (first in byte order) and 2k+1 of this block
        if (b == 0 && 2 * k + 1 < NPRE) {
          hi32 = pre[2 * k];
          lo32 = pre[2 * k + 1];
        } else {
          const size_t g = 32 * b + 2 * k - NPRE;
          hi32 = sha_msg_word(msg, len, aligned, g);
          lo32 = sha_msg_word(msg, len, aligned, g + 1);
        }
        w[k] = ((uint64_t)bswap32(hi32) << 32) | bswap32(lo32);
      }
    }
    if (b == nblk - 1) w[15] = (uint64_t)total << 3;   // sha512.c:196-203 (high 64 bits are 0)
    sha512_compress(st, w);
  }
#pragma unroll
  for (int k = 0; k < 8; k++) {
    out[2 * k] = bswap32((uint32_t)(st[k] >> 32));
    out[2 * k + 1] = bswap32((uint32_t)st[k]);
  }
}

}  // namespace ed
