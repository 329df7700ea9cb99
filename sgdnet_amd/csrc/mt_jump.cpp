// Jump-ahead for R's Mersenne-Twister (MT19937): the state J draws further down the SAME stream
// without generating the draws in between.
//
// Why: the reference takes one R::runif per inner iteration from R's single generator
// (src/saga-sparse.h:261, Rcpp::RNGScope in src/RcppExports.cpp:14,27), so "the sample order of
// set.seed(s)" is one sequential stream.  A single generator makes 10M draws in 5.3 ms on the
// device (r_rng_device.hip) -- six times the 0.84 ms epoch it feeds.  MT19937 is linear over
// GF(2): with phi its characteristic polynomial (degree 19937) and g(x) = x^J mod phi(x),
//     x[t + J + j] = XOR over { i : g_i = 1 } of x[t + i + j]          (every word j of the state window)
// so G generators started J = n apart (or L = n / G apart inside an epoch) reproduce the one stream
// in parallel.  (Haramoto, Matsumoto, Nishimura, Panneton, L'Ecuyer: "Efficient jump ahead for
// F2-linear random number generators", INFORMS J. Comput. 20(3), 2008; the polynomial arithmetic
// below is written from that description.)
//
// phi is not hard-coded: it is recovered once per process from the generator's own output with
// Berlekamp-Massey (the minimal polynomial of any non-zero bit sequence of MT19937 is phi, which
// is irreducible), then x^J mod phi by square-and-multiply.  Host cost: ~0.05 s for phi, ~0.05 s
// per distinct J (cached).
#include <stdint.h>
#include <string.h>

#include <map>
#include <mutex>
#include <vector>

#include "common.hpp"

namespace sgdnet {

namespace {

constexpr int kN = 624, kM = 397;
constexpr int kDeg = 19937;
constexpr int kW = (2 * kDeg + 64 + 63) / 64;   // words of the polynomial scratch (degree < 2 * 19937 + 64)
using Bits = std::vector<uint64_t>;

inline int get_bit(const Bits& b, int64_t i) { return (int)((b[(size_t)(i >> 6)] >> (i & 63)) & 1ull); }
inline void flip_bit(Bits& b, int64_t i) { b[(size_t)(i >> 6)] ^= 1ull << (i & 63); }

// the raw (untempered) word sequence of a generator state: window = mt[0..623], then the recurrence
void mt_sequence(const uint32_t* mt, size_t count, std::vector<uint32_t>& x) {
  x.resize(count);
  for (size_t i = 0; i < (size_t)kN && i < count; ++i) x[i] = mt[i];
  for (size_t k = kN; k < count; ++k) {
    const uint32_t y = (x[k - kN] & 0x80000000u) | (x[k - kN + 1] & 0x7fffffffu);
    x[k] = x[k - kN + kM] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
  }
}

// dst ^= src << shift (bit shift), src has `words` words
void xor_shifted(Bits& dst, const Bits& src, size_t words, int64_t shift) {
  const size_t ws = (size_t)(shift >> 6);
  const int bs = (int)(shift & 63);
  if (bs == 0) {
    for (size_t i = 0; i < words && i + ws < dst.size(); ++i) dst[i + ws] ^= src[i];
  } else {
    for (size_t i = 0; i < words; ++i) {
      if (i + ws < dst.size()) dst[i + ws] ^= src[i] << bs;
      if (i + ws + 1 < dst.size()) dst[i + ws + 1] ^= src[i] >> (64 - bs);
    }
  }
}

// Berlekamp-Massey over GF(2) on s[0..N): connection polynomial C (C_0 = 1) with
// s_n = XOR_{i=1..L} C_i s_{n-i}.  Word-parallel discrepancy: C against the bit-reversed window.
int berlekamp_massey(const Bits& s, int64_t N, Bits& C) {
  const size_t words = (size_t)((N + 63) / 64) + 2;
  C.assign(words, 0);
  Bits B(words, 0), T;
  C[0] = B[0] = 1;
  int64_t L = 0, m = 1;
  auto window_rev = [&](int64_t hi) -> uint64_t {   // bits s[hi], s[hi-1], ..., s[hi-63] as bits 0..63
    // gather s[hi-63 .. hi] then reverse
    const int64_t lo = hi - 63;
    uint64_t v;
    if (lo >= 0) {
      const size_t w = (size_t)(lo >> 6);
      const int b = (int)(lo & 63);
      v = s[w] >> b;
      if (b) v |= s[w + 1] << (64 - b);
    } else {
      if (hi < 0) return 0;
      v = s[0] << (-lo);                          // bits below index 0 are zero
    }
    return __builtin_bitreverse64(v);
  };
  for (int64_t n = 0; n < N; ++n) {
    // d = XOR_{i=0..L} C_i s_{n-i}
    uint64_t acc = 0;
    const size_t cw = (size_t)(L >> 6) + 1;
    for (size_t w = 0; w < cw; ++w) acc ^= C[w] & window_rev(n - (int64_t)(w << 6));
    if (__builtin_parityll(acc)) {
      if (2 * L <= n) {
        T = C;
        xor_shifted(C, B, (size_t)((n + 64) >> 6) + 1 < words ? (size_t)((n + 64) >> 6) + 1 : words, m);
        L = n + 1 - L;
        B.swap(T);
        m = 1;
      } else {
        xor_shifted(C, B, (size_t)((n + 64) >> 6) + 1 < words ? (size_t)((n + 64) >> 6) + 1 : words, m);
        ++m;
      }
    } else {
      ++m;
    }
  }
  return (int)L;
}

struct CharPoly {
  Bits phi;                       // phi_0 .. phi_19937 (phi_19937 = 1)
  std::vector<Bits> shifted;      // phi << s for s = 0..63, for word-aligned reduction
  bool ok = false;
};

const CharPoly& char_poly() {
  static CharPoly cp;
  static std::once_flag once;
  std::call_once(once, [] {
    sgdnet_rng r;
    sgdnet_rng_seed(&r, 5489u);
    std::vector<uint32_t> x;
    const int64_t N = 2 * (int64_t)kDeg + 64;
    mt_sequence(r.mt, (size_t)N + kN, x);
    Bits s((size_t)(N + 63) / 64 + 2, 0);
    for (int64_t t = 0; t < N; ++t)
      if (x[(size_t)t + kN] & 1u) flip_bit(s, t);  // lowest bit of the regenerated words
    Bits C;
    const int L = berlekamp_massey(s, N, C);
    if (L != kDeg) return;                         // cp.ok stays false: callers fall back
    cp.phi.assign((size_t)kW, 0);
    for (int k = 0; k <= kDeg; ++k)                // phi(x) = x^L C(1/x)
      if (get_bit(C, kDeg - k)) flip_bit(cp.phi, k);
    cp.shifted.resize(64);
    const size_t pw = (size_t)(kDeg >> 6) + 2;
    for (int sft = 0; sft < 64; ++sft) {
      cp.shifted[(size_t)sft].assign(pw + 1, 0);
      xor_shifted(cp.shifted[(size_t)sft], cp.phi, pw, sft);
    }
    cp.ok = true;
  });
  return cp;
}

// t mod phi, in place; t has bits up to `top`
void reduce(Bits& t, int64_t top, const CharPoly& cp) {
  const size_t pw = cp.shifted[0].size();
  for (int64_t i = top; i >= kDeg; --i) {
    if (!get_bit(t, i)) continue;
    const int64_t sh = i - kDeg;
    const size_t ws = (size_t)(sh >> 6);
    const Bits& ps = cp.shifted[(size_t)(sh & 63)];
    for (size_t w = 0; w < pw && w + ws < t.size(); ++w) t[w + ws] ^= ps[w];
  }
}

std::mutex g_cache_mu;
std::map<uint64_t, std::vector<uint32_t>> g_cache;

}  // namespace

// g(x) = x^J mod phi(x) as 624 32-bit words (bit i of the vector = g_i, i < 19937).
// Returns false if phi could not be established (never observed; the caller then keeps one generator).
bool mt_jump_poly(uint64_t J, uint32_t* out624) {
  {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    auto it = g_cache.find(J);
    if (it != g_cache.end()) {
      memcpy(out624, it->second.data(), sizeof(uint32_t) * kN);
      return true;
    }
  }
  const CharPoly& cp = char_poly();
  if (!cp.ok) return false;
  static uint16_t spread[256];
  static std::once_flag once;
  std::call_once(once, [] {
    for (int b = 0; b < 256; ++b) {
      uint16_t v = 0;
      for (int k = 0; k < 8; ++k)
        if (b & (1 << k)) v |= (uint16_t)(1u << (2 * k));
      spread[b] = v;
    }
  });
  Bits r((size_t)kW, 0), t((size_t)kW, 0);
  r[0] = 1;                                       // x^0
  int top_bit = 63;
  while (top_bit > 0 && !((J >> top_bit) & 1ull)) --top_bit;
  const size_t half = (size_t)(kDeg >> 6) + 1;    // words that hold a reduced polynomial
  for (int b = top_bit; b >= 0; --b) {
    // square: bit i -> bit 2i
    std::fill(t.begin(), t.end(), 0);
    for (size_t w = 0; w < half; ++w) {
      const uint64_t v = r[w];
      uint64_t lo = 0, hi = 0;
      for (int k = 0; k < 4; ++k) {
        lo |= (uint64_t)spread[(v >> (8 * k)) & 0xff] << (16 * k);
        hi |= (uint64_t)spread[(v >> (32 + 8 * k)) & 0xff] << (16 * k);
      }
      if (2 * w < t.size()) t[2 * w] = lo;
      if (2 * w + 1 < t.size()) t[2 * w + 1] = hi;
    }
    reduce(t, 2 * (int64_t)kDeg, cp);
    if ((J >> b) & 1ull) {                        // times x
      uint64_t carry = 0;
      for (size_t w = 0; w <= half; ++w) {
        const uint64_t nc = t[w] >> 63;
        t[w] = (t[w] << 1) | carry;
        carry = nc;
      }
      reduce(t, kDeg, cp);
    }
    r.swap(t);
  }
  std::vector<uint32_t> packed((size_t)kN, 0);
  for (int i = 0; i < kDeg; ++i)
    if (get_bit(r, i)) packed[(size_t)(i >> 5)] |= 1u << (i & 31);
  memcpy(out624, packed.data(), sizeof(uint32_t) * kN);
  std::lock_guard<std::mutex> lk(g_cache_mu);
  g_cache[J] = std::move(packed);
  return true;
}

// Host form of the jump (tests, and callers with a handful of states): the device form is
// mt_jump_kernel in r_rng_device.hip.  mti is kept: the window moves J words down the sequence.
void mt_jump_host(const sgdnet_rng* in, const uint32_t* poly624, sgdnet_rng* out) {
  std::vector<uint32_t> x;
  mt_sequence(in->mt, (size_t)kDeg + kN, x);
  uint32_t acc[kN];
  memset(acc, 0, sizeof(acc));
  for (int i = 0; i < kDeg; ++i) {
    if (!((poly624[i >> 5] >> (i & 31)) & 1u)) continue;
    const uint32_t* xi = x.data() + i;
    for (int j = 0; j < kN; ++j) acc[j] ^= xi[j];
  }
  out->mti = in->mti;
  memcpy(out->mt, acc, sizeof(acc));
}

}  // namespace sgdnet

extern "C" {

int sgdnet_rng_jump_poly(uint64_t draws, uint32_t* poly624) {
  if (!poly624) return SGDNET_EINVAL;
  return sgdnet::mt_jump_poly(draws, poly624) ? SGDNET_OK : SGDNET_EUNSUPPORTED;
}

void sgdnet_rng_jump(const sgdnet_rng* in, const uint32_t* poly624, sgdnet_rng* out) {
  sgdnet::mt_jump_host(in, poly624, out);
}

}  // extern "C"
