// Host model of the block algorithm in tools/study/exact_scan.h: 64 "lanes" x E elements, the same data flow as the
// device code (per-lane prefix, cross-lane exclusive scan, tie resolution through the parity of the truncated prefix at the
// previous tie, first-violation restart), checked against the literal sequential fp32 loop.  Built and run by
// tests/test_exact_scan_model.py (g++ -O2 -ffp-contract=off).  Exit code 0 = every case bit-identical.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "exact_scan.h"

using namespace hsm::xscan;

static inline unsigned f2u(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float u2f(unsigned u) { float f; memcpy(&f, &u, 4); return f; }

static long g_iterations = 0, g_restarts = 0, g_singles = 0, g_ties = 0;

// one block: elements x[0 .. 64*E), consumed from index `start`; returns the running sum after all of them
template <int E>
float block_sum(float s, const float* x, int start) {
  constexpr int N = 64 * E;
  volatile float vs;  // (keep every real addition a real fp32 addition)
  while (start < N) {
    const unsigned sb = f2u(s);
    const int es = (int)((sb >> 23) & 0xffu);
    if (es == 0 || es == 255) {
      if ((sb << 1) == 0u) {  // +-0: the sum stays 0 over zeros, then takes the first non-zero element exactly
        int j = start;
        while (j < N && (f2u(x[j]) << 1) == 0u) ++j;
        if (j == N) { vs = s + 0.0f; for (int k = start; k < N; ++k) { vs = vs + x[k]; } return vs; }  // (only signs of zero)
        vs = s; for (int k = start; k <= j; ++k) vs = vs + x[k];
        s = vs; start = j + 1;
      } else {
        vs = s + x[start]; s = vs; ++start;
      }
      ++g_singles;
      continue;
    }
    ++g_iterations;
    const unsigned sign = sb & 0x80000000u;
    const int S0 = (int)((sb & 0x7fffffu) | 0x800000u);
    // lanes
    unsigned Tl[64][E], fl[64][E];   // inclusive lane-local prefix of t (mod 2^32), flags
    unsigned LT[64];
    for (int l = 0; l < 64; ++l) {
      unsigned acc = 0;
      for (int j = 0; j < E; ++j) {
        const int g = l * E + j;
        Elem e = convert(f2u(x[g]), sign, es);
        if (g < start) { e.t = 0; e.flags = 0; }
        acc += (unsigned)e.t;
        Tl[l][j] = acc;
        fl[l][j] = e.flags;
      }
      LT[l] = acc;
    }
    unsigned base[64];
    { unsigned a = 0; for (int l = 0; l < 64; ++l) { base[l] = a; a += LT[l]; } }
    // ties: tau = parity of the truncated prefix at the tie; b = tau ^ tau(previous tie), the first against parity(S0)
    uint64_t has_tie = 0, last_tau = 0;
    for (int l = 0; l < 64; ++l) {
      int last = -1;
      for (int j = 0; j < E; ++j) if (fl[l][j] & kTie) last = j;
      if (last >= 0) {
        has_tie |= 1ull << l;
        if ((base[l] + Tl[l][last]) & 1u) last_tau |= 1ull << l;
      }
    }
    int cl[64][E]; int LC[64];
    for (int l = 0; l < 64; ++l) {
      const uint64_t below = has_tie & ((1ull << l) - 1ull);
      unsigned prev_tau = (unsigned)S0 & 1u;
      if (below) { const int p = 63 - __builtin_clzll(below); prev_tau = (unsigned)((last_tau >> p) & 1ull); }
      int acc = 0;
      for (int j = 0; j < E; ++j) {
        if (fl[l][j] & kTie) {
          const unsigned tau = (base[l] + Tl[l][j]) & 1u;
          const unsigned b = tau ^ prev_tau;
          prev_tau = tau;
          if (b) acc += (fl[l][j] & kNeg) ? -1 : 1;
          ++g_ties;
        }
        cl[l][j] = acc;
      }
      LC[l] = acc;
    }
    int cbase[64];
    { int a = 0; for (int l = 0; l < 64; ++l) { cbase[l] = a; a += LC[l]; } }
    // prefixes and the first violation
    int first = -1; int Sprev_at_first = 0; int S_last = 0;
    for (int l = 0; l < 64 && first < 0; ++l) {
      for (int j = 0; j < E; ++j) {
        const int g = l * E + j;
        const int S = (int)((unsigned)S0 + base[l] + Tl[l][j] + (unsigned)(cbase[l] + cl[l][j]));
        const bool viol = g >= start && ((fl[l][j] & kBig) || S < kLo || S > kHi);
        if (viol) {
          first = g;
          // the prefix before element g
          const unsigned Tprev = j ? Tl[l][j - 1] : 0u;
          const int cprev = j ? cl[l][j - 1] : 0;
          Sprev_at_first = (int)((unsigned)S0 + base[l] + Tprev + (unsigned)(cbase[l] + cprev));
          break;
        }
        S_last = S;
      }
    }
    if (first < 0) return u2f(compose(sign, es, S_last));
    const float sprev = first > start ? u2f(compose(sign, es, Sprev_at_first)) : s;
    vs = sprev + x[first];
    s = vs;
    start = first + 1;
    ++g_restarts;
  }
  return s;
}

template <int E>
float scan_sum(const std::vector<float>& x, int prologue) {
  constexpr int N = 64 * E;
  volatile float s = 0.0f;
  size_t i = 0;
  for (; i < x.size() && (int)i < prologue; ++i) s = s + x[i];   // the device's serial prologue of a step
  std::vector<float> blk(N);
  float r = s;
  while (i < x.size()) {
    const size_t n = std::min((size_t)N, x.size() - i);
    for (size_t k = 0; k < (size_t)N; ++k) blk[k] = k < n ? x[i + k] : 0.0f;
    r = block_sum<E>(r, blk.data(), 0);
    i += n;
  }
  return r;
}

static float literal_sum(const std::vector<float>& x) {
  volatile float s = 0.0f;
  for (float v : x) s = s + v;
  return s;
}

// `file <path> <n> <chains>`: chain-major fp32 products of one GN step; per chain the cost counters of the block algorithm
static int run_file(const char* path, int n, int chains) {
  FILE* f = fopen(path, "rb");
  if (!f) return 2;
  int bad = 0;
  for (int c = 0; c < chains; ++c) {
    std::vector<float> x(n);
    if (fread(x.data(), 4, n, f) != (size_t)n) return 2;
    g_iterations = g_restarts = g_singles = g_ties = 0;
    const float got = scan_sum<7>(x, 64), want = literal_sum(x);
    bad += f2u(got) != f2u(want);
    printf("chain %d: n %d blocks %d scan iterations %ld (restarts %ld) single adds %ld ties %ld %s\n", c, n, (n - 64 + 447) / 448, g_iterations,
           g_restarts, g_singles, g_ties, f2u(got) == f2u(want) ? "ok" : "MISMATCH");
  }
  fclose(f);
  return bad ? 1 : 0;
}

int main(int argc, char** argv) {
  if (argc > 4 && !strcmp(argv[1], "file")) return run_file(argv[2], atoi(argv[3]), atoi(argv[4]));
  const int cases = argc > 1 ? atoi(argv[1]) : 3000;
  std::mt19937_64 rng(12345);
  auto urand = [&]() { return (double)(rng() >> 11) * (1.0 / 9007199254740992.0); };
  long bad = 0, total = 0;
  for (int c = 0; c < cases; ++c) {
    const int kind = c % 12;
    const int n = 1 + (int)(rng() % 3000);
    std::vector<float> x(n);
    for (int i = 0; i < n; ++i) {
      float v;
      switch (kind) {
        case 0: v = (float)urand(); break;                                         // positive, one scale
        case 1: v = (float)(urand() - 0.5); break;                                  // signed random walk
        case 2: v = (float)std::ldexp(urand() - 0.5, (int)(rng() % 40) - 20); break;  // wild magnitudes
        case 3: v = (float)((int)(rng() % 17) - 8) * 0.125f; break;                 // few mantissa bits: ties everywhere, exact cancellations
        case 4: v = (float)std::ldexp((double)(1 + rng() % 7), (int)(rng() % 30) - 15) * ((rng() & 1) ? 1.f : -1.f); break;  // powers of two-ish
        case 5: v = (rng() % 3) ? 0.0f : (float)(urand() - 0.3); break;             // many zeros
        case 6: v = (i % 50 == 0) ? (float)std::ldexp(urand(), 20) * ((rng() & 1) ? 1.f : -1.f) : (float)(urand() * 1e-3); break;  // spikes
        case 7: v = (float)std::ldexp(urand() - 0.5, -140 + (int)(rng() % 20)); break;  // denormal range
        case 8: { const float a = (float)urand(); v = (i & 1) ? -a * 0.999f : a; } break;  // near-cancelling pairs
        case 9: v = (float)(urand() * urand() * urand()) * ((rng() % 5) ? 1.f : -1.f); break;   // products, mostly positive
        case 10: v = (i % 97 == 13) ? -(float)(i) * 0.5f : 0.5f; break;             // ramps with resets through zero
        default: v = (float)std::ldexp(1.0 + (double)(rng() % 4) * 0.25, (int)(rng() % 6) - 3) * ((rng() % 3) ? 1.f : -1.f); break;  // 2-bit mantissas
      }
      if (kind == 5 && (rng() % 11) == 0) v = -0.0f;
      x[i] = v;
    }
    if (kind == 6 && (c % 24) == 6) x[n / 2] = INFINITY;   // an infinity mid-way (the reference would carry it too)
    const float want = literal_sum(x);
    for (int prologue : {0, 64}) {
      const float g7 = scan_sum<7>(x, prologue), g4 = scan_sum<4>(x, prologue), g1 = scan_sum<1>(x, prologue);
      ++total;
      const bool ok = (f2u(g7) == f2u(want) || (std::isnan(want) && std::isnan(g7))) &&
                      (f2u(g4) == f2u(want) || (std::isnan(want) && std::isnan(g4))) &&
                      (f2u(g1) == f2u(want) || (std::isnan(want) && std::isnan(g1)));
      if (!ok) {
        if (bad < 10) fprintf(stderr, "MISMATCH case %d kind %d n %d prologue %d: want %a got7 %a got4 %a got1 %a\n", c, kind, n, prologue, want, g7, g4, g1);
        ++bad;
      }
    }
  }
  printf("{\"cases\": %ld, \"mismatches\": %ld, \"scan_iterations\": %ld, \"restarts\": %ld, \"single_adds\": %ld, \"ties\": %ld}\n",
         total, bad, g_iterations, g_restarts, g_singles, g_ties);
  return bad ? 1 : 0;
}
