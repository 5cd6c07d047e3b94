// A small, fast DEFLATE compressor for the text the writer produces (RFC 1951 streams in RFC 1952 gzip members; any inflate
// reads them).  Not a port of zlib: one pass of greedy LZ77 over a 32 KiB window with a single-probe hash of SIX bytes, then one
// dynamic-Huffman block per 256 K tokens.
//
// Why not zlib's deflate: scores_*.txt.gz is "chr\tstart\tend" plus S numbers "%.5f" per line.  The digits of the numbers carry
// the information and no LZ77 match shortens them; what zlib's 3- and 4-byte matches do on such text is replace four digits at
// ~3.4 bits each by a length and a distance code of ~20 bits, and spoil the literal statistics on the way.  Level 6 spends 75 %
// of the writer's time walking hash chains for that.  Here a match must be at least 6 bytes long (a repeated number, a repeated
// row, the coordinates' common prefix), found with one probe; everything else goes out as Huffman-coded literals.  On the
// synthetic 833-biosample scores this is ~6 x faster than level 6 at a SMALLER output (tools/io_bench.py); on rows that repeat
// (real chromatin data: long runs of identical bins) the 6-byte hash finds the same long matches zlib does.
//
// Reference behaviour replaced: gzip.open(..., "wt") in scores.py:523 (Python's zlib, level 9).
#pragma once
#include <stdint.h>
#include <string.h>
#include "epg_crc32.h"   // CRC-32 of the gzip trailer

#include <algorithm>
#include <vector>

namespace epgdeflate {

struct BitWriter {
    unsigned char* p;
    uint64_t acc = 0;
    int n = 0;                                       // bits in acc, < 8 between calls
    explicit BitWriter(unsigned char* out) : p(out) {}
    inline void put(uint64_t bits, int nbits) {      // nbits <= 56, LSB first; writes up to 8 bytes past p
        acc |= bits << n;
        n += nbits;
        memcpy(p, &acc, 8);                          // little endian hosts only (x86-64, aarch64 LE)
        p += n >> 3;
        acc >>= n & ~7;
        n &= 7;
    }
    inline void align_byte() {
        if (n > 0) { *p++ = (unsigned char)acc; }
        acc = 0;
        n = 0;
    }
};

// Huffman code lengths (<= maxbits) for n symbols with the given frequencies; symbols with frequency 0 get length 0.
// Plain two-queue Huffman on the sorted frequencies; if the tree is deeper than maxbits the frequencies are flattened
// (halved, floor 1) and the tree rebuilt -- slightly off the optimum, always terminates, and blocks of 256 K tokens of text
// do not get there in practice.
inline void huffman_lengths(const uint32_t* freq, int n, int maxbits, uint8_t* lens) {
    struct Node { uint64_t w; int left, right; };
    std::vector<uint32_t> f(freq, freq + n);
    std::vector<int> order;
    std::vector<Node> nodes;
    std::vector<int> depth;
    for (;;) {
        order.clear();
        for (int i = 0; i < n; ++i)
            if (f[i]) order.push_back(i);
        memset(lens, 0, (size_t)n);
        if (order.empty()) return;
        if (order.size() == 1) { lens[order[0]] = 1; return; }
        std::sort(order.begin(), order.end(), [&](int a, int b) { return f[a] != f[b] ? f[a] < f[b] : a < b; });
        const int m = (int)order.size();
        nodes.assign((size_t)2 * m - 1, Node{0, -1, -1});
        for (int i = 0; i < m; ++i) nodes[i].w = f[order[i]];
        int leaf = 0, inner = m, made = m;         // two queues: leaves [leaf, m), inner nodes [inner, made)
        auto pop = [&]() {
            if (leaf < m && (inner >= made || nodes[leaf].w <= nodes[inner].w)) return leaf++;
            return inner++;
        };
        while (made < 2 * m - 1) {
            const int a = pop(), b = pop();
            nodes[made] = Node{nodes[a].w + nodes[b].w, a, b};
            ++made;
        }
        depth.assign((size_t)2 * m - 1, 0);
        int deepest = 0;
        for (int i = 2 * m - 2; i >= m; --i) {
            depth[nodes[i].left] = depth[i] + 1;
            depth[nodes[i].right] = depth[i] + 1;
        }
        for (int i = 0; i < m; ++i) deepest = std::max(deepest, depth[i]);
        if (deepest <= maxbits) {
            for (int i = 0; i < m; ++i) lens[order[i]] = (uint8_t)depth[i];
            return;
        }
        for (int i = 0; i < n; ++i)
            if (f[i]) f[i] = (f[i] + 1) / 2;
    }
}

// canonical codes, bit-reversed so that they can be written LSB first
inline void huffman_codes(const uint8_t* lens, int n, uint16_t* codes) {
    uint32_t count[16] = {0}, next[16] = {0};
    for (int i = 0; i < n; ++i) ++count[lens[i]];
    count[0] = 0;
    uint32_t code = 0;
    for (int b = 1; b < 16; ++b) {
        code = (code + count[b - 1]) << 1;
        next[b] = code;
    }
    for (int i = 0; i < n; ++i) {
        const int l = lens[i];
        if (!l) { codes[i] = 0; continue; }
        uint32_t c = next[l]++, r = 0;
        for (int b = 0; b < l; ++b) { r = (r << 1) | (c & 1); c >>= 1; }
        codes[i] = (uint16_t)r;
    }
}

struct Tables {
    uint8_t len_code[256];      // (length - 3) -> length code - 257
    uint8_t len_extra[29];
    uint16_t len_base[29];
    uint8_t dist_code_lo[512];  // (dist - 1) < 512 -> distance code; larger: dist_code_hi[(dist - 1) >> 7]
    uint8_t dist_code_hi[256];
    uint8_t dist_extra[30];
    uint16_t dist_base[30];
    Tables() {
        static const uint8_t le[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        static const uint8_t de[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        int base = 3;
        for (int c = 0; c < 28; ++c) {
            len_extra[c] = le[c];
            len_base[c] = (uint16_t)base;
            for (int k = 0; k < (1 << le[c]); ++k) len_code[base - 3 + k] = (uint8_t)c;
            base += 1 << le[c];
        }
        len_extra[28] = 0; len_base[28] = 258; len_code[255] = 28;      // length 258 has its own code
        int db = 1;
        for (int c = 0; c < 30; ++c) {
            dist_extra[c] = de[c];
            dist_base[c] = (uint16_t)db;
            for (int k = 0; k < (1 << de[c]); ++k) {
                const int d1 = db - 1 + k;                                // dist - 1
                if (d1 < 512) dist_code_lo[d1] = (uint8_t)c;
                if ((k & 127) == 0 && d1 >= 512) dist_code_hi[d1 >> 7] = (uint8_t)c;
            }
            db += 1 << de[c];
        }
    }
    inline int dcode(uint32_t d1) const { return d1 < 512 ? dist_code_lo[d1] : dist_code_hi[d1 >> 7]; }
};

inline const Tables& tables() {
    static const Tables t;
    return t;
}

constexpr int kHashBits = 15;
constexpr uint32_t kWindow = 32768;
constexpr size_t kBlockBytes = 1u << 19;          // input bytes per DEFLATE block

inline uint32_t hash6(uint64_t v) { return (uint32_t)(((v << 16) * 0x9E3779B185EBCA87ull) >> (64 - kHashBits)); }

struct Match {
    uint32_t pos;        // offset in the block
    uint16_t len3;       // length - 3
    uint16_t dist1;      // distance - 1
};

// One block: raw[0, raw_len) with the matches found in it (ascending, not overlapping); everything between matches is literals
// -- they are read from the input itself, there is no token buffer.
inline void emit_block(BitWriter& bw, const unsigned char* raw, size_t raw_len, const Match* mt, size_t nm, bool final_block) {
    const Tables& T = tables();
    uint32_t lf[286] = {0}, df[30] = {0};
    {
        uint32_t h[4][256];                          // four histograms: no store-to-load chains on runs of one byte
        memset(h, 0, sizeof(h));
        size_t p = 0;
        for (size_t k = 0; k <= nm; ++k) {
            const size_t e = k < nm ? mt[k].pos : raw_len;
            for (; p + 4 <= e; p += 4) { ++h[0][raw[p]]; ++h[1][raw[p + 1]]; ++h[2][raw[p + 2]]; ++h[3][raw[p + 3]]; }
            for (; p < e; ++p) ++h[0][raw[p]];
            if (k < nm) {
                ++lf[257 + T.len_code[mt[k].len3]];
                ++df[T.dcode(mt[k].dist1)];
                p = e + mt[k].len3 + 3;
            }
        }
        for (int i = 0; i < 256; ++i) lf[i] = h[0][i] + h[1][i] + h[2][i] + h[3][i];
    }
    lf[256] = 1;
    uint8_t ll[286], dl[30];
    huffman_lengths(lf, 286, 15, ll);
    huffman_lengths(df, 30, 15, dl);
    int nd = 0;
    for (int i = 0; i < 30; ++i) nd += dl[i] != 0;
    if (nd < 2) {                                  // fewer than two distance codes in use: a complete tree of two one-bit codes
        int used = -1;                             // (RFC 1951 3.2.7 allows a single code; two are read by every inflate)
        for (int i = 0; i < 30; ++i)
            if (dl[i]) used = i;
        memset(dl, 0, sizeof(dl));
        if (used <= 0) { dl[0] = 1; dl[1] = 1; }
        else { dl[0] = 1; dl[used] = 1; }
    }
    uint16_t lc[286], dc[30];
    huffman_codes(ll, 286, lc);
    huffman_codes(dl, 30, dc);
    // stored blocks when Huffman coding does not pay (incompressible input)
    uint64_t bits = 0;
    for (int i = 0; i < 286; ++i) bits += (uint64_t)lf[i] * ll[i];
    for (int c = 0; c < 29; ++c) bits += (uint64_t)lf[257 + c] * T.len_extra[c];
    for (int c = 0; c < 30; ++c) bits += (uint64_t)df[c] * (dl[c] + T.dist_extra[c]);
    if (bits / 8 + 200 > raw_len + 5 * (raw_len / 65535 + 1)) {
        size_t off = 0;
        do {
            const size_t n = std::min<size_t>(65535, raw_len - off);
            const bool last = final_block && off + n == raw_len;
            bw.put(last ? 1 : 0, 1);
            bw.put(0, 2);
            bw.align_byte();
            const uint16_t le16 = (uint16_t)n, nle = (uint16_t)~le16;
            memcpy(bw.p, &le16, 2); memcpy(bw.p + 2, &nle, 2);
            bw.p += 4;
            memcpy(bw.p, raw + off, n);
            bw.p += n;
            off += n;
        } while (off < raw_len);
        return;
    }
    int hlit = 286, hdist = 30;
    while (hlit > 257 && ll[hlit - 1] == 0) --hlit;
    while (hdist > 1 && dl[hdist - 1] == 0) --hdist;
    // run-length code of the hlit + hdist code lengths
    uint8_t all[316];
    memcpy(all, ll, (size_t)hlit);
    memcpy(all + hlit, dl, (size_t)hdist);
    const int nall = hlit + hdist;
    uint8_t sym[316], ext[316];
    int ns = 0;
    uint32_t cf[19] = {0};
    for (int i = 0; i < nall;) {
        int run = 1;
        while (i + run < nall && all[i + run] == all[i]) ++run;
        const int v = all[i];
        int left = run;
        if (v == 0) {
            while (left >= 11) { const int r = std::min(left, 138); sym[ns] = 18; ext[ns++] = (uint8_t)(r - 11); left -= r; }
            if (left >= 3) { sym[ns] = 17; ext[ns++] = (uint8_t)(left - 3); left = 0; }
            while (left-- > 0) { sym[ns] = 0; ext[ns++] = 0; }
        } else {
            sym[ns] = (uint8_t)v; ext[ns++] = 0; --left;
            while (left >= 3) { const int r = std::min(left, 6); sym[ns] = 16; ext[ns++] = (uint8_t)(r - 3); left -= r; }
            while (left-- > 0) { sym[ns] = (uint8_t)v; ext[ns++] = 0; }
        }
        i += run;
    }
    for (int i = 0; i < ns; ++i) ++cf[sym[i]];
    uint8_t cl[19];
    huffman_lengths(cf, 19, 7, cl);
    int ncl = 0;
    for (int i = 0; i < 19; ++i) ncl += cl[i] != 0;
    if (ncl == 1) {                                // one code-length symbol: give it a partner so that the tree is complete
        for (int i = 0; i < 19; ++i)
            if (!cl[i]) { cl[i] = 1; break; }
    }
    uint16_t cc[19];
    huffman_codes(cl, 19, cc);
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    int hclen = 19;
    while (hclen > 4 && cl[order[hclen - 1]] == 0) --hclen;
    bw.put(final_block ? 1 : 0, 1);
    bw.put(2, 2);
    bw.put((uint32_t)(hlit - 257), 5);
    bw.put((uint32_t)(hdist - 1), 5);
    bw.put((uint32_t)(hclen - 4), 4);
    for (int i = 0; i < hclen; ++i) bw.put(cl[order[i]], 3);
    for (int i = 0; i < ns; ++i) {
        bw.put(cc[sym[i]], cl[sym[i]]);
        if (sym[i] == 16) bw.put(ext[i], 2);
        else if (sym[i] == 17) bw.put(ext[i], 3);
        else if (sym[i] == 18) bw.put(ext[i], 7);
    }
    uint32_t lit[256];                               // code | length << 16 of the literals: one load per byte
    for (int i = 0; i < 256; ++i) lit[i] = (uint32_t)lc[i] | ((uint32_t)ll[i] << 16);
    size_t p = 0;
    for (size_t k = 0; k <= nm; ++k) {
        const size_t e = k < nm ? mt[k].pos : raw_len;
        for (; p + 3 <= e; p += 3) {                 // three literals per put: at most 45 bits
            const uint32_t a = lit[raw[p]], b = lit[raw[p + 1]], c = lit[raw[p + 2]];
            const int la = (int)(a >> 16), lb = (int)(b >> 16);
            bw.put((uint64_t)(a & 0xffffu) | ((uint64_t)(b & 0xffffu) << la) | ((uint64_t)(c & 0xffffu) << (la + lb)), la + lb + (int)(c >> 16));
        }
        for (; p < e; ++p) { const uint32_t a = lit[raw[p]]; bw.put(a & 0xffffu, (int)(a >> 16)); }
        if (k < nm) {
            const uint32_t l3 = mt[k].len3, d1 = mt[k].dist1;
            const int lcd = T.len_code[l3];
            bw.put(lc[257 + lcd], ll[257 + lcd]);
            if (T.len_extra[lcd]) bw.put(l3 + 3 - T.len_base[lcd], T.len_extra[lcd]);
            const int dcd = T.dcode(d1);
            bw.put(dc[dcd], dl[dcd]);
            if (T.dist_extra[dcd]) bw.put(d1 + 1 - T.dist_base[dcd], T.dist_extra[dcd]);
            p = e + l3 + 3;
        }
    }
    bw.put(lc[256], ll[256]);
}

// raw DEFLATE stream of in[0, n) appended at out (which must have room for n + n / 8 + 1024 bytes); returns the end.
// Every position is probed (a candidate must agree in six bytes); after 32 misses in a row every second, after 64 every third
// ... position, up to every eighth, until the next match (a stretch of digits does not repay probing every byte).
inline unsigned char* deflate_fast(const unsigned char* in, size_t n, unsigned char* out) {
    BitWriter bw(out);
    if (n == 0) {                                  // one empty fixed-Huffman block
        bw.put(1, 1); bw.put(1, 2); bw.put(0, 7);
        bw.align_byte();
        return bw.p;
    }
    std::vector<uint32_t> head((size_t)1 << kHashBits, 0xffffffffu);
    std::vector<Match> mt;
    mt.reserve(kBlockBytes / 16);
    const size_t last_hashable = n >= 8 ? n - 8 : 0;        // positions below it have 8 readable bytes
    const size_t kSegment = (size_t)1 << 30;                // positions in the hash table are 32-bit offsets into a 1 GiB segment
    size_t seg0 = 0;
    for (size_t b0 = 0; b0 < n; b0 += kBlockBytes) {
        if (b0 - seg0 >= kSegment) {                         // (a multiple of the block size) forget the window, start over
            seg0 = b0;
            std::fill(head.begin(), head.end(), 0xffffffffu);
        }
        const size_t b1 = std::min(n, b0 + kBlockBytes);
        const size_t probe_end = std::min(b1, last_hashable);
        mt.clear();
        size_t pos = b0;
        uint32_t misses = 0;
        while (pos < probe_end) {
            uint64_t v;
            memcpy(&v, in + pos, 8);
            const uint32_t h = hash6(v);
            const uint32_t rel = (uint32_t)(pos - seg0);
            const uint32_t crel = head[h];
            head[h] = rel;
            if (crel != 0xffffffffu && rel - crel <= kWindow) {
                const size_t cand = seg0 + crel;
                uint64_t w;
                memcpy(&w, in + cand, 8);
                const uint64_t x = v ^ w;
                if ((x & 0xffffffffffffull) == 0) {                // six bytes equal
                    const size_t maxlen = std::min<size_t>(258, b1 - pos);   // a match does not cross the block's end
                    size_t len = x ? (size_t)(__builtin_ctzll(x) >> 3) : 8;
                    if (len == 8) {
                        while (len + 8 <= maxlen) {
                            uint64_t a, c;
                            memcpy(&a, in + pos + len, 8);
                            memcpy(&c, in + cand + len, 8);
                            if (a != c) { len += (size_t)(__builtin_ctzll(a ^ c) >> 3); break; }
                            len += 8;
                        }
                        if (len + 8 > maxlen)
                            while (len < maxlen && in[pos + len] == in[cand + len]) ++len;
                    }
                    if (len > maxlen) len = maxlen;
                    if (len >= 6) {
                        mt.push_back(Match{(uint32_t)(pos - b0), (uint16_t)(len - 3), (uint16_t)(pos - cand - 1)});
                        const size_t end = pos + len;               // index inside the match: later repeats of its tail are found
                        for (size_t q = pos + 1; q < end && q < last_hashable; q += 2) {
                            uint64_t u;
                            memcpy(&u, in + q, 8);
                            head[hash6(u)] = (uint32_t)(q - seg0);
                        }
                        pos = end;
                        misses = 0;
                        continue;
                    }
                }
            }
            pos += 1 + std::min<uint32_t>(misses++ >> 5, 7);
        }
        emit_block(bw, in + b0, b1 - b0, mt.data(), mt.size(), b1 == n);
    }
    bw.align_byte();
    return bw.p;
}

// one gzip member (RFC 1952) holding in[0, n)
inline void gzip_member_fast(const unsigned char* in, size_t n, std::vector<unsigned char>& out) {
    out.resize(n + n / 8 + 1024 + 26);
    static const unsigned char hdr[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 0xff};
    memcpy(out.data(), hdr, 10);
    unsigned char* e = deflate_fast(in, n, out.data() + 10);
    const uint32_t crc = epgcrc::crc32_fast(0, in, n);
    const uint32_t isize = (uint32_t)n;
    memcpy(e, &crc, 4);
    memcpy(e + 4, &isize, 4);
    out.resize((size_t)(e + 8 - out.data()));
}

}  // namespace epgdeflate
