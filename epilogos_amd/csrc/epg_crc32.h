// CRC-32 (the gzip / zlib polynomial, reflected) by carry-less multiplication: Gopal et al., "Fast CRC Computation for Generic
// Polynomials Using PCLMULQDQ Instruction" (Intel, 2009).  The reader checks the CRC of every inflated member; zlib's table
// driven crc32 runs at ~1 GB/s and was a quarter of the CPU time of a cold whole-genome run.  Four 128-bit lanes are folded
// forward by 512 bits per step (x^(512+64) mod P and x^512 mod P), then into one lane by 128 bits per step, then reduced
// 128 -> 64 -> 32 bits (Barrett).  Same result as zlib's crc32() for every input (tests/test_native_io.py); on CPUs without
// PCLMULQDQ, and for the last < 16 bytes, zlib's function is used.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <zlib.h>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace epgcrc {

#if defined(__x86_64__)
// n >= 64, n % 16 == 0; state in / out is the bit-inverted CRC register like inside zlib
__attribute__((target("pclmul,sse4.1"))) inline uint32_t fold_pclmul(uint32_t state, const unsigned char* p, size_t n) {
    // constants of the reflected domain for P = 0x104C11DB7: x^(4*128+64), x^(4*128), x^(128+64), x^128, x^96 (mod P), P' and mu
    const __m128i k512 = _mm_set_epi64x(0x01c6e41596, 0x0154442bd4);
    const __m128i k128 = _mm_set_epi64x(0x00ccaa009e, 0x01751997d0);
    const __m128i k96 = _mm_set_epi64x(0, 0x0163cd6124);
    const __m128i pmu = _mm_set_epi64x(0x01f7011641, 0x01db710641);
    __m128i lane[4];
    for (int k = 0; k < 4; ++k) lane[k] = _mm_loadu_si128(reinterpret_cast<const __m128i*>(p + 16 * k));
    lane[0] = _mm_xor_si128(lane[0], _mm_cvtsi32_si128((int)state));
    p += 64;
    n -= 64;
    for (; n >= 64; p += 64, n -= 64) {
        for (int k = 0; k < 4; ++k) {
            const __m128i lo = _mm_clmulepi64_si128(lane[k], k512, 0x00), hi = _mm_clmulepi64_si128(lane[k], k512, 0x11);
            lane[k] = _mm_xor_si128(_mm_xor_si128(lo, hi), _mm_loadu_si128(reinterpret_cast<const __m128i*>(p + 16 * k)));
        }
    }
    __m128i acc = lane[0];
    for (int k = 1; k < 4; ++k) {
        const __m128i lo = _mm_clmulepi64_si128(acc, k128, 0x00), hi = _mm_clmulepi64_si128(acc, k128, 0x11);
        acc = _mm_xor_si128(_mm_xor_si128(lo, hi), lane[k]);
    }
    for (; n >= 16; p += 16, n -= 16) {
        const __m128i lo = _mm_clmulepi64_si128(acc, k128, 0x00), hi = _mm_clmulepi64_si128(acc, k128, 0x11);
        acc = _mm_xor_si128(_mm_xor_si128(lo, hi), _mm_loadu_si128(reinterpret_cast<const __m128i*>(p)));
    }
    // 128 -> 64 bits
    const __m128i low32 = _mm_setr_epi32(-1, 0, -1, 0);
    __m128i t = _mm_clmulepi64_si128(acc, k128, 0x10);
    acc = _mm_xor_si128(_mm_srli_si128(acc, 8), t);
    t = _mm_srli_si128(acc, 4);
    acc = _mm_clmulepi64_si128(_mm_and_si128(acc, low32), k96, 0x00);
    acc = _mm_xor_si128(acc, t);
    // Barrett reduction to 32 bits
    t = _mm_clmulepi64_si128(_mm_and_si128(acc, low32), pmu, 0x10);
    t = _mm_clmulepi64_si128(_mm_and_si128(t, low32), pmu, 0x00);
    acc = _mm_xor_si128(acc, t);
    return (uint32_t)_mm_extract_epi32(acc, 1);
}

inline bool have_pclmul() {
    static const bool ok = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    return ok;
}
#endif

// zlib's crc32(crc, p, n) semantics (crc = 0 to start)
inline uint32_t crc32_fast(uint32_t crc, const unsigned char* p, size_t n) {
#if defined(__x86_64__)
    if (n >= 64 && have_pclmul()) {
        const size_t body = n & ~(size_t)15;
        crc = ~fold_pclmul(~crc, p, body);
        p += body;
        n -= body;
    }
#endif
    while (n) {
        const size_t k = n < (1u << 30) ? n : (1u << 30);
        crc = (uint32_t)crc32(crc, p, (uInt)k);
        p += k;
        n -= k;
    }
    return crc;
}

}  // namespace epgcrc
