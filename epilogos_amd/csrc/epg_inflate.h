// A fast INFLATE (RFC 1951) for whole gzip files that are already in memory (mmap) and are decompressed into ONE contiguous
// buffer -- what the reader does with every input matrix (helpers.py:152-155 reads them through pandas/gzip).  Not a port of
// zlib: no streaming state machine, no window copy (the output buffer IS the window), a 64-bit bit buffer refilled with one
// unaligned load, 11-bit / 8-bit root tables with second-level tables, matches copied eight bytes at a time.  zlib's inflate
// gives 0.3 GB/s of text on the state matrices and the largest chromosome is the critical path of a cold run.
//
// Safety: every member's CRC-32 and ISIZE are checked by the caller (in parallel, see slurp()); any failure here or there makes
// the caller fall back to zlib's inflate, so a bug in this file costs time, not correctness.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

namespace epginflate {

constexpr int LIT_ROOT = 11, DIST_ROOT = 8;
constexpr uint32_t K_LITERAL = 0x8000u, K_EOB = 0x4000u, K_PTR = 0x2000u, K_INVALID = 0x1000u;   // in bits 8..15 of an entry
// entry = value << 16 | kind/extra << 8 | bits to consume (whole code length, also in second-level tables)

struct Tables {
    uint32_t lit[1 << LIT_ROOT];
    uint32_t dist[1 << DIST_ROOT];
    std::vector<uint32_t> lit_sub, dist_sub;
};

static const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

inline uint32_t symbol_entry(bool litlen, int sym, int len) {
    if (litlen) {
        if (sym < 256) return ((uint32_t)sym << 16) | K_LITERAL | (uint32_t)len;
        if (sym == 256) return K_EOB | (uint32_t)len;
        if (sym > 285) return K_INVALID | (uint32_t)len;
        return ((uint32_t)LEN_BASE[sym - 257] << 16) | ((uint32_t)LEN_EXTRA[sym - 257] << 8) | (uint32_t)len;
    }
    if (sym > 29) return K_INVALID | (uint32_t)len;
    return ((uint32_t)DIST_BASE[sym] << 16) | ((uint32_t)DIST_EXTRA[sym] << 8) | (uint32_t)len;
}

// Decode tables of one Huffman code.  Returns false for an over-subscribed code and for an incomplete one, with zlib's one
// exception: a code whose longest length is 1 (the one-code distance tree of RFC 1951 3.2.7; an empty distance tree of a
// block without matches) -- its unused entries are invalid and end the decode if they are ever hit.
inline bool build(const uint8_t* lens, int n, bool litlen, int root, uint32_t* table, std::vector<uint32_t>& sub) {
    int count[16] = {0};
    for (int i = 0; i < n; ++i) ++count[lens[i]];
    count[0] = 0;
    long left = 1;
    for (int b = 1; b < 16; ++b) {
        left = (left << 1) - count[b];
        if (left < 0) return false;
    }
    if (left > 0) {                                       // incomplete: a corrupt stream should fail here, not at the CRC
        int maxl = 0;
        for (int b = 1; b < 16; ++b) if (count[b]) maxl = b;
        if (maxl > 1) return false;
    }
    uint32_t next[16];
    uint32_t code = 0;
    for (int b = 1; b < 16; ++b) {
        code = (code + (uint32_t)count[b - 1]) << 1;
        next[b] = code;
    }
    const uint32_t rsize = 1u << root, rmask = rsize - 1;
    for (uint32_t i = 0; i < rsize; ++i) table[i] = K_INVALID | 1u;
    uint8_t maxlen[1 << LIT_ROOT];
    memset(maxlen, 0, rsize);
    uint16_t rev[288];
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (!l) continue;
        uint32_t c = next[l]++, r = 0;
        for (int b = 0; b < l; ++b) { r = (r << 1) | (c & 1); c >>= 1; }
        rev[s] = (uint16_t)r;
        if (l <= root) {
            const uint32_t e = symbol_entry(litlen, s, l);
            for (uint32_t i = r; i < rsize; i += 1u << l) table[i] = e;
        } else if (l > maxlen[r & rmask]) {
            maxlen[r & rmask] = (uint8_t)l;
        }
    }
    sub.clear();
    for (uint32_t p = 0; p < rsize; ++p) {
        if (!maxlen[p]) continue;
        const uint32_t sb = (uint32_t)maxlen[p] - (uint32_t)root;
        const uint32_t start = (uint32_t)sub.size();
        if (start + (1u << sb) > 0xffffu) return false;
        sub.resize(start + (1u << sb), K_INVALID | 1u);
        table[p] = (start << 16) | K_PTR | (sb << 8) | (uint32_t)root;
    }
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (l <= root) continue;
        const uint32_t r = rev[s], p = r & rmask;
        const uint32_t start = table[p] >> 16, sb = (table[p] >> 8) & 0xf;
        const uint32_t e = symbol_entry(litlen, s, l);
        for (uint32_t i = r >> root; i < (1u << sb); i += 1u << (l - root)) sub[start + i] = e;
    }
    return true;
}

struct Out {
    unsigned char* base;
    size_t pos, cap;          // the buffer has 320 more bytes behind cap: a symbol's writes may run that far past it
};

// Inflates ONE raw DEFLATE stream that starts at in[0] and lies within in[0, in_avail) (what follows it -- a gzip trailer,
// the next member -- is never interpreted; the last 16 bytes are decoded from a zero-padded copy, so nothing behind
// in + in_avail is read).  Appends to out, growing it through grow(min_cap) (false = cannot).  Returns the number of input
// bytes consumed, 0 on any error.
template <typename Grow>
inline size_t inflate_raw(const unsigned char* in, size_t in_avail, Out& out, Grow&& grow) {
    static const uint8_t CL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    const unsigned char* const in0 = in;
    const size_t out0 = out.pos;                          // a match never reaches back past this stream's first byte (an offset:
                                                          // grow() may move the buffer)
    const unsigned char* in_end = in + in_avail;        // true end of the readable input (of the tail copy once switched)
    const unsigned char* in_lim = in_avail >= 16 ? in_end - 16 : in - 1;   // refills are allowed while in <= in_lim
    unsigned char tail[48];
    bool in_tail = false;
    size_t tail_origin = 0;                               // offset in the real input of tail[0]
    uint64_t bitbuf = 0;
    int bitcnt = 0;
    Tables T, F;
    uint8_t lens[320];
    bool have_fixed = false;
    // before any refill: move to the zero-padded copy of the last bytes when the real input is nearly used up
    auto input_ok = [&]() -> bool {
        if (in <= in_lim) return true;
        if (in_tail) return false;                        // more than 8 bytes of padding consumed: the stream is truncated
        const size_t left = in < in_end ? (size_t)(in_end - in) : 0;
        if (in > in_end) return false;
        tail_origin = (size_t)(in - in0);
        memset(tail, 0, sizeof(tail));
        memcpy(tail, in, left);
        in = tail;
        in_end = tail + left;
        in_lim = in_end + 8;                              // the copy has >= 24 zero bytes behind the data
        in_tail = true;
        return true;
    };
#define EPG_REFILL()                                                          \
    do {                                                                      \
        uint64_t w_;                                                          \
        memcpy(&w_, in, 8);                                                   \
        bitbuf |= w_ << bitcnt;                                               \
        in += (63 - bitcnt) >> 3;                                             \
        bitcnt |= 56;                                                         \
    } while (0)
#define EPG_TAKE(n) (bitbuf >>= (n), bitcnt -= (n))
    for (;;) {
        if (!input_ok()) return 0;
        EPG_REFILL();
        const int final_block = (int)(bitbuf & 1), type = (int)((bitbuf >> 1) & 3);
        EPG_TAKE(3);
        if (type == 0) {
            // stored: back to a byte boundary, LEN / NLEN, raw bytes
            EPG_TAKE(bitcnt & 7);
            if (in_tail) {
                // The bit buffer may hold bytes that were read from the REAL input before the switch to the tail copy: giving
                // them back can lead to before tail[0] (found by tools/asan_io.sh: a stored block met in the last 16 bytes
                // read tail[-6..]).  A stored block is read with explicit bounds, so go back to the real input for it.
                const size_t consumed = tail_origin + (size_t)(in - tail) - (size_t)(bitcnt >> 3);
                in = in0 + consumed;
                in_end = in0 + in_avail;
                in_lim = in_avail >= 16 ? in_end - 16 : in0 - 1;
                in_tail = false;
            } else {
                in -= bitcnt >> 3;                        // give the whole bytes in the bit buffer back
            }
            bitbuf = 0; bitcnt = 0;
            if (in + 4 > in_end) return 0;
            const uint32_t len = (uint32_t)in[0] | ((uint32_t)in[1] << 8), nlen = (uint32_t)in[2] | ((uint32_t)in[3] << 8);
            if ((len ^ nlen) != 0xffffu) return 0;
            in += 4;
            if (in + len > in_end) return 0;
            if (out.pos + len > out.cap && !grow(out.pos + len)) return 0;
            memcpy(out.base + out.pos, in, len);
            out.pos += len;
            in += len;
            if (final_block) break;
            continue;
        }
        const Tables* tb;
        if (type == 1) {
            if (!have_fixed) {
                for (int i = 0; i < 144; ++i) lens[i] = 8;
                for (int i = 144; i < 256; ++i) lens[i] = 9;
                for (int i = 256; i < 280; ++i) lens[i] = 7;
                for (int i = 280; i < 288; ++i) lens[i] = 8;
                if (!build(lens, 288, true, LIT_ROOT, F.lit, F.lit_sub)) return 0;
                for (int i = 0; i < 32; ++i) lens[i] = 5;
                if (!build(lens, 32, false, DIST_ROOT, F.dist, F.dist_sub)) return 0;
                have_fixed = true;
            }
            tb = &F;
        } else if (type == 2) {
            const int hlit = (int)(bitbuf & 31) + 257, hdist = (int)((bitbuf >> 5) & 31) + 1, hclen = (int)((bitbuf >> 10) & 15) + 4;
            EPG_TAKE(14);
            if (hlit > 286 || hdist > 30) return 0;
            uint8_t cl[19] = {0};
            for (int i = 0; i < hclen; ++i) {
                if (bitcnt < 3) { if (!input_ok()) return 0; EPG_REFILL(); }
                cl[CL_ORDER[i]] = (uint8_t)(bitbuf & 7);
                EPG_TAKE(3);
            }
            uint32_t ct[128];                             // the code-length code: symbol << 8 | length, by the next 7 bits
            {
                int count[8] = {0};
                for (int i = 0; i < 19; ++i) ++count[cl[i]];
                count[0] = 0;
                long left = 1;
                for (int b = 1; b < 8; ++b) { left = (left << 1) - count[b]; if (left < 0) return 0; }
                if (left > 0) return 0;                   // an incomplete code-length code (zlib rejects it as well)
                uint32_t next[8], code = 0;
                for (int b = 1; b < 8; ++b) { code = (code + (uint32_t)count[b - 1]) << 1; next[b] = code; }
                for (int i = 0; i < 128; ++i) ct[i] = 0xffffffffu;
                for (int sy = 0; sy < 19; ++sy) {
                    const int l = cl[sy];
                    if (!l) continue;
                    uint32_t c = next[l]++, r = 0;
                    for (int b = 0; b < l; ++b) { r = (r << 1) | (c & 1); c >>= 1; }
                    for (uint32_t i = r; i < 128; i += 1u << l) ct[i] = ((uint32_t)sy << 8) | (uint32_t)l;
                }
            }
            int i = 0;
            const int total = hlit + hdist;
            while (i < total) {
                if (bitcnt < 14) { if (!input_ok()) return 0; EPG_REFILL(); }
                const uint32_t e = ct[bitbuf & 127];
                if (e == 0xffffffffu) return 0;
                EPG_TAKE((int)(e & 0xff));
                const int sym = (int)(e >> 8);
                if (sym < 16) { lens[i++] = (uint8_t)sym; continue; }
                int rep;
                uint8_t val = 0;
                if (sym == 16) {
                    if (i == 0) return 0;
                    val = lens[i - 1];
                    rep = 3 + (int)(bitbuf & 3);
                    EPG_TAKE(2);
                } else if (sym == 17) {
                    rep = 3 + (int)(bitbuf & 7);
                    EPG_TAKE(3);
                } else {
                    rep = 11 + (int)(bitbuf & 127);
                    EPG_TAKE(7);
                }
                if (i + rep > total) return 0;
                memset(lens + i, val, (size_t)rep);
                i += rep;
            }
            if (lens[256] == 0) return 0;                 // no end-of-block code
            if (!build(lens, hlit, true, LIT_ROOT, T.lit, T.lit_sub)) return 0;
            if (!build(lens + hlit, hdist, false, DIST_ROOT, T.dist, T.dist_sub)) return 0;
            tb = &T;
        } else {
            return 0;
        }
        const uint32_t* const lit = tb->lit;
        const uint32_t* const dst = tb->dist;
        const uint32_t* const lsub = tb->lit_sub.data();
        const uint32_t* const dsub = tb->dist_sub.data();
        constexpr uint32_t LMASK = (1u << LIT_ROOT) - 1, DMASK = (1u << DIST_ROOT) - 1;
#define EPG_LIT_ENTRY(e)                                                                                       \
    do {                                                                                                       \
        (e) = lit[bitbuf & LMASK];                                                                             \
        if ((e) & K_PTR) (e) = lsub[((e) >> 16) + ((bitbuf >> LIT_ROOT) & ((1u << (((e) >> 8) & 0xf)) - 1))]; \
    } while (0)
        // ---- the block's symbols
        bool eob = false;
        while (!eob) {
            if (out.pos > out.cap && !grow(out.pos + (1u << 20))) return 0;
            if (!input_ok()) return 0;
            unsigned char* o = out.base + out.pos;
            unsigned char* const o_stop = out.base + out.cap;        // a symbol's writes may run up to 320 bytes past it
            while (o <= o_stop && in <= in_lim) {
                EPG_REFILL();                                        // >= 56 bits: three codes (45) or one code + length extra (20)
                uint32_t e;
                EPG_LIT_ENTRY(e);
                EPG_TAKE((int)(e & 0xff));
                if (e & K_LITERAL) {
                    *o++ = (unsigned char)(e >> 16);
                    EPG_LIT_ENTRY(e);
                    EPG_TAKE((int)(e & 0xff));
                    if (e & K_LITERAL) {
                        *o++ = (unsigned char)(e >> 16);
                        EPG_LIT_ENTRY(e);
                        EPG_TAKE((int)(e & 0xff));
                        if (e & K_LITERAL) {
                            *o++ = (unsigned char)(e >> 16);
                            continue;
                        }
                    }
                }
                if (e & (K_EOB | K_INVALID)) {
                    if (e & K_INVALID) return 0;
                    eob = true;
                    break;
                }
                // length: base + extra bits (<= 5), then the distance code (<= 15 + 13 bits)
                const uint32_t lx = (e >> 8) & 0xf;
                const uint32_t len = (e >> 16) + (uint32_t)(bitbuf & ((1u << lx) - 1));
                EPG_TAKE((int)lx);
                if (bitcnt < 32) {
                    if (in > in_lim) {                               // (cannot refill here: finish this symbol through the slow door)
                        out.pos = (size_t)(o - out.base);
                        if (!input_ok()) return 0;
                        o = out.base + out.pos;
                    }
                    EPG_REFILL();
                }
                uint32_t d = dst[bitbuf & DMASK];
                if (d & K_PTR) d = dsub[(d >> 16) + ((bitbuf >> DIST_ROOT) & ((1u << ((d >> 8) & 0xf)) - 1))];
                if (d & (K_INVALID | K_LITERAL | K_EOB)) return 0;
                EPG_TAKE((int)(d & 0xff));
                const uint32_t dx = (d >> 8) & 0xf;
                const size_t dist = (size_t)(d >> 16) + (size_t)(bitbuf & ((1u << dx) - 1));
                EPG_TAKE((int)dx);
                if (dist > (size_t)(o - out.base) - out0) return 0;  // before the start of THIS member's output
                const unsigned char* sp = o - dist;
                unsigned char* const oe = o + len;
                if (dist >= 8) {
                    do {
                        uint64_t w;
                        memcpy(&w, sp, 8);
                        memcpy(o, &w, 8);
                        sp += 8;
                        o += 8;
                    } while (o < oe);
                } else if (dist == 1) {
                    memset(o, *sp, len);
                } else {
                    // 2 <= dist < 8 (a state and its tab repeated: "18\t18\t..."): the output is periodic from sp on, so after
                    // `dist` bytes the same source serves at twice the distance; once that reaches 8, word copies
                    size_t d = dist;
                    while (d < 8) {
                        for (size_t k = 0; k < d; ++k) o[k] = sp[k];     // may run past oe: there is slack
                        o += d;
                        d <<= 1;
                        if (o >= oe) break;
                    }
                    while (o < oe) {
                        uint64_t w;
                        memcpy(&w, o - d, 8);
                        memcpy(o, &w, 8);
                        o += 8;
                    }
                }
                o = oe;
            }
            out.pos = (size_t)(o - out.base);
        }
#undef EPG_LIT_ENTRY
        if (final_block) break;
    }
#undef EPG_REFILL
#undef EPG_TAKE
    // whole bytes still in the bit buffer were not consumed (in tail mode some of them may stem from before tail[0]: offsets,
    // not pointers)
    const size_t back = (size_t)(bitcnt >> 3);
    if (in_tail) {
        const size_t fwd = (size_t)(in - tail);
        if (fwd > (size_t)(in_end - tail) + back) return 0;   // the stream ran into the padding: truncated
        return tail_origin + fwd - back;
    }
    if (in - back > in_end) return 0;
    return (size_t)(in - in0) - back;
}

}  // namespace epginflate
