// libepilogos_io.so -- native TSV(.gz) parser and "%.5f"/gzip writer (host C++17, zlib + std::thread).
// Contract: include/epilogos_io.h.
#include "epilogos_io.h"
#include "epg_crc32.h"
#include "epg_deflate.h"
#include "epg_inflate.h"

#include <cmath>
#include <ctime>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <memory>
#include <mutex>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local char g_err[512] = "";

int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return -1;
}

bool ends_with_gz(const char* path) {
    const size_t n = strlen(path);
    return n >= 2 && path[n - 2] == 'g' && path[n - 1] == 'z';   // the reference's test: name.endswith("gz")
}

// Census of the library's runnable threads (a caller inside the library + the workers it started; a caller waiting in a join
// does not count): epgio_thread_census reports the peak, so that a multi-rank run can show that the ranks of one node together
// stay inside the node's CPU budget (driver._host_budget shares it out).
std::atomic<int> g_live{0}, g_peak{0};
thread_local int tl_counted = 0;

void census_add(int d) {
    const int now = g_live.fetch_add(d) + d;
    int pk = g_peak.load();
    while (now > pk && !g_peak.compare_exchange_weak(pk, now)) {}
}

struct Census {
    Census() { if (tl_counted++ == 0) census_add(1); }
    ~Census() { if (--tl_counted == 0) g_live.fetch_sub(1); }
};

// A caller that hands its work to worker threads and waits for them is not runnable meanwhile: declared just before the
// workers are started, gives the caller's place in the census up until the end of the scope.
struct Uncount {
    const bool was = tl_counted > 0;
    Uncount() { if (was) g_live.fetch_sub(1); }
    ~Uncount() { if (was) census_add(1); }
};

void join_all(std::vector<std::thread>& th) {
    for (auto& t : th) t.join();
}

// threads to use when the caller says 0: EPILOGOS_HOST_THREADS when set (the share of the host the driver gives this rank: a
// node's cores divided by the ranks on it, capped by -c), else the hardware threads, capped by the cgroup CPU quota (a container
// may see 256 hardware threads and be allowed 16 of them at a time) and by 64
// The budget is held in an atomic: reader threads ask for it while Python (or torch, importing next to the early readers) may
// be changing the environment, and a getenv beside a setenv is undefined before glibc 2.40.  The variable is read ONCE, when the
// library first needs a budget; after that only epgio_set_host_threads changes it (the binding calls it from host_budget()).
static std::atomic<int> g_host_threads{-1};      // -1: not looked at yet; 0: no budget given (hardware threads / cgroup quota)
int n_threads(int32_t req) {
    if (req > 0) return req;
    int b = g_host_threads.load(std::memory_order_relaxed);
    if (b < 0) {
        const char* e = getenv("EPILOGOS_HOST_THREADS");
        const int v = e ? atoi(e) : 0;
        int expect = -1;
        g_host_threads.compare_exchange_strong(expect, v > 0 ? v : 0);
        b = g_host_threads.load(std::memory_order_relaxed);
    }
    if (b > 0) return std::min(b, 64);
    static const int cached = [] {
        unsigned hc = std::thread::hardware_concurrency();
        if (!hc) hc = 4;
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[64];
            long long per = 0;
            if (fscanf(f, "%63s %lld", q, &per) == 2 && strcmp(q, "max") != 0 && per > 0) {
                const long long quota = (atoll(q) + per - 1) / per;
                if (quota >= 1 && (unsigned)quota < hc) hc = (unsigned)quota;
            }
            fclose(f);
        }
        return (int)std::min(hc, 64u);
    }();
    return cached;
}

// readers inside open_table_impl right now, and how many files the caller reads side by side (epgio_set_reader_plan): with
// threads == 0 ("share") every parallel phase of a file takes the rank's budget divided by the larger of the two -- one thread
// each while sixteen files inflate side by side, more for the last, largest files once the small ones are done
std::atomic<int> g_readers{0}, g_reader_plan{0};
int reader_share() {
    const int active = std::max(1, std::max(g_readers.load(), g_reader_plan.load()));
    return std::max(1, n_threads(0) / active);
}

// The whole (decompressed) file in memory, with 16 readable bytes of slack after the text (the value parser looks a few
// characters ahead).  Plain files are parsed straight from the mapping of the file -- no copy --, gzip files are inflated
// from the mapped compressed bytes with the zlib inflate API into ONE malloc'd buffer whose size comes from the gzip trailer
// (ISIZE; a multi-member or > 4 GiB file simply grows it).  The first version went through gzread into a std::vector that
// doubled and zero-filled as it grew: 0.55 GB/s for plain text, 0.15-0.2 GB/s of text for gzip.
// The inflated text of a big file is gigabytes; handing such a buffer back to the kernel (munmap) and faulting the next one in
// holds the process's memory-map lock for tenths of a second, and every other reader thread -- they all fault pages into their
// own buffers -- stands still meanwhile (profiles/r04l: eight readers returning in the same millisecond, one second after their
// parses had ended).  Buffers of 64 MiB and more are therefore kept and reused by the next file that fits (the driver reads
// the largest files first); they are never returned during a run -- the command line leaves through _exit.
struct BigBufs {
    std::mutex m;
    std::vector<std::pair<char*, size_t>> free_list;
};
BigBufs& big_bufs() { static BigBufs* b = new BigBufs(); return *b; }
constexpr size_t BIG_MIN = (size_t)64 << 20;

void advise_huge(void* p, size_t n);

char* big_alloc(size_t n, size_t* cap) {
    if (n >= BIG_MIN) {
        BigBufs& b = big_bufs();
        std::lock_guard<std::mutex> g(b.m);
        int best = -1;
        for (int i = 0; i < (int)b.free_list.size(); ++i)
            if (b.free_list[i].second >= n && (best < 0 || b.free_list[i].second < b.free_list[best].second)) best = i;
        if (best >= 0) {
            char* p = b.free_list[best].first;
            *cap = b.free_list[best].second;
            b.free_list.erase(b.free_list.begin() + best);
            return p;
        }
    }
    char* p = (char*)malloc(n);
    if (p && n >= BIG_MIN) advise_huge(p, n);
    *cap = n;
    return p;
}

void big_free(char* p, size_t cap) {
    if (!p) return;
    if (cap >= BIG_MIN) {
        // keep the address range, give the PAGES back now: MADV_DONTNEED takes the memory-map lock shared (page faults of the
        // other readers go on) and runs on this reader's own thread, in parallel with the others' -- where munmap, and the
        // teardown of tens of gigabytes at process exit, are exclusive and serial
#ifdef MADV_DONTNEED
        const uintptr_t a = ((uintptr_t)p + 4095) & ~(uintptr_t)4095, e = ((uintptr_t)p + cap) & ~(uintptr_t)4095;
        if (e > a) madvise((void*)a, (size_t)(e - a), MADV_DONTNEED);
#endif
        BigBufs& b = big_bufs();
        std::lock_guard<std::mutex> g(b.m);
        if (b.free_list.size() < 64) { b.free_list.emplace_back(p, cap); return; }
    }
    free(p);
}

char* big_grow(char* p, size_t old_cap, size_t used, size_t n, size_t* cap) {   // a larger buffer with the first `used` bytes kept
    char* q = big_alloc(n, cap);
    if (!q) return nullptr;
    if (used) memcpy(q, p, used);
    big_free(p, old_cap);
    return q;
}

struct Text {
    const char* data = nullptr;
    size_t size = 0;
    void* map = nullptr;           // mmap of the file (plain: the text itself)
    size_t map_len = 0;
    char* heap = nullptr;          // inflated text
    size_t heap_cap = 0;
    ~Text() {
        big_free(heap, heap_cap);
        if (map) munmap(map, map_len);
    }
};

bool is_gzip(const unsigned char* p, size_t n) { return n >= 18 && p[0] == 0x1f && p[1] == 0x8b; }

// Ask for transparent huge pages under a big fresh buffer (the pool runs THP in "madvise" mode): a whole genome's text is 26 GB
// of first-touch pages, 6.5 M faults of 4 KiB.  Harmless where it is not honoured.
void advise_huge(void* p, size_t n) {
#ifdef MADV_HUGEPAGE
    const uintptr_t a = ((uintptr_t)p + (1u << 21) - 1) & ~(uintptr_t)((1u << 21) - 1), e = ((uintptr_t)p + n) & ~(uintptr_t)((1u << 21) - 1);
    if (n >= (8u << 20) && e > a) madvise((void*)a, (size_t)(e - a), MADV_HUGEPAGE);
#else
    (void)p; (void)n;
#endif
}

// gzip members of in[0, flen) -> t.heap through epginflate, then ISIZE and CRC-32 of every member (epg_crc32.h: carry-less
// multiplication, ~20 GB/s where zlib's table-driven crc32 does 1 GB/s; in pieces over the threads).  false = use zlib instead.
bool inflate_own(const unsigned char* in, size_t flen, size_t cap_hint, Text& t, int32_t threads) {
    struct Member { size_t out0, out1; uint32_t crc, isize; };
    std::vector<Member> members;
    epginflate::Out out{nullptr, 0, cap_hint};
    t.heap = big_alloc(cap_hint + 320 + 16, &t.heap_cap);
    if (!t.heap) return false;
    out.cap = t.heap_cap - 320 - 16;                       // (a reused buffer may be larger than asked for)
    out.base = (unsigned char*)t.heap;
    auto grow = [&](size_t min_cap) {
        size_t ncap = out.cap + out.cap / 2 + (1u << 24);
        if (ncap < min_cap) ncap = min_cap + (1u << 24);
        size_t got = 0;
        char* nh = big_grow(t.heap, t.heap_cap, out.pos, ncap + 320 + 16, &got);
        if (!nh) return false;
        t.heap = nh;
        t.heap_cap = got;
        out.base = (unsigned char*)nh;
        out.cap = got - 320 - 16;
        return true;
    };
    size_t pos = 0;
    while (pos < flen && is_gzip(in + pos, flen - pos)) {
        const unsigned char* h = in + pos;
        if (h[2] != 8 || (h[3] & 0xe0)) return false;
        size_t hp = pos + 10;
        const int flg = h[3];
        if (flg & 4) { if (hp + 2 > flen) return false; hp += 2 + ((size_t)in[hp] | ((size_t)in[hp + 1] << 8)); }
        if (flg & 8) { while (hp < flen && in[hp]) ++hp; ++hp; }
        if (flg & 16) { while (hp < flen && in[hp]) ++hp; ++hp; }
        if (flg & 2) hp += 2;
        if (hp + 8 > flen) return false;
        const size_t out0 = out.pos;
        const size_t used = epginflate::inflate_raw(in + hp, flen - hp - 8, out, grow);
        if (!used) return false;
        const unsigned char* tr = in + hp + used;
        Member m;
        m.out0 = out0; m.out1 = out.pos;
        m.crc = (uint32_t)tr[0] | (uint32_t)tr[1] << 8 | (uint32_t)tr[2] << 16 | (uint32_t)tr[3] << 24;
        m.isize = (uint32_t)tr[4] | (uint32_t)tr[5] << 8 | (uint32_t)tr[6] << 16 | (uint32_t)tr[7] << 24;
        if ((uint32_t)(m.out1 - m.out0) != m.isize) return false;
        members.push_back(m);
        pos = hp + used + 8;
    }
    if (members.empty()) return false;
    // CRC-32: pieces of <= 32 MiB over the threads, stitched per member with crc32_combine
    struct Piece { size_t off, len; uint32_t crc; };
    std::vector<Piece> pieces;
    std::vector<size_t> first(members.size() + 1, 0);
    const size_t PIECE = (size_t)32 << 20;
    for (size_t k = 0; k < members.size(); ++k) {
        first[k] = pieces.size();
        for (size_t o = members[k].out0; o < members[k].out1; o += PIECE) pieces.push_back(Piece{o, std::min(PIECE, members[k].out1 - o), 0});
    }
    first[members.size()] = pieces.size();
    const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)(threads > 0 ? threads : reader_share()), pieces.size()));
    Uncount uncount_;
    std::vector<std::thread> th;
    for (int w = 0; w < T; ++w)
        th.emplace_back([&, w] {
            Census census_;
            for (size_t i = (size_t)w; i < pieces.size(); i += (size_t)T)
                pieces[i].crc = epgcrc::crc32_fast(0, (const unsigned char*)t.heap + pieces[i].off, pieces[i].len);
        });
    join_all(th);
    for (size_t k = 0; k < members.size(); ++k) {
        uLong crc = crc32(0L, Z_NULL, 0);
        for (size_t i = first[k]; i < first[k + 1]; ++i) crc = crc32_combine(crc, pieces[i].crc, (z_off_t)pieces[i].len);
        if ((uint32_t)crc != members[k].crc) return false;
    }
    memset(t.heap + out.pos, 0, 16);
    t.data = t.heap;
    t.size = out.pos;
    return true;
}

// BGZF -- the blocked gzip of htslib (bgzip, tabix): every member is at most 64 KiB and carries its own compressed size in an extra
// subfield ('B', 'C'), so the members' boundaries -- and, from their trailers, their places in the output -- are known without
// decoding anything, and the members of ONE file inflate in parallel.  (A plain gzip file is one serial stream: the largest file
// of a genome is then the critical path of a cold run, 3 s of inflate on one core)
struct BgzfBlock { size_t in_off; uint32_t in_len, isize, crc; size_t out_off; };

bool bgzf_header(const unsigned char* h, size_t avail, size_t* hdr_len, size_t* total) {
    if (avail < 18 + 8 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || h[3] != 4) return false;   // FEXTRA and nothing else
    const size_t xlen = (size_t)h[10] | ((size_t)h[11] << 8);
    if (12 + xlen + 8 > avail) return false;
    size_t bsize = 0;
    for (size_t x = 12; x + 4 <= 12 + xlen;) {
        const size_t slen = (size_t)h[x + 2] | ((size_t)h[x + 3] << 8);
        if (x + 4 + slen > 12 + xlen) return false;
        if (h[x] == 'B' && h[x + 1] == 'C' && slen == 2) bsize = ((size_t)h[x + 4] | ((size_t)h[x + 5] << 8)) + 1;
        x += 4 + slen;
    }
    if (bsize < 12 + xlen + 8 || bsize > avail) return false;
    *hdr_len = 12 + xlen;
    *total = bsize;
    return true;
}

bool bgzf_index(const unsigned char* in, size_t flen, std::vector<BgzfBlock>& blocks, size_t* out_total) {
    size_t pos = 0, out = 0;
    while (pos < flen) {
        size_t hl = 0, total = 0;
        if (!bgzf_header(in + pos, flen - pos, &hl, &total)) return false;      // not BGZF all the way: the general reader takes it
        const unsigned char* tr = in + pos + total - 8;
        BgzfBlock b;
        b.in_off = pos + hl;
        b.in_len = (uint32_t)(total - hl - 8);
        b.crc = (uint32_t)tr[0] | (uint32_t)tr[1] << 8 | (uint32_t)tr[2] << 16 | (uint32_t)tr[3] << 24;
        b.isize = (uint32_t)tr[4] | (uint32_t)tr[5] << 8 | (uint32_t)tr[6] << 16 | (uint32_t)tr[7] << 24;
        if (b.isize > 65536) return false;
        b.out_off = out;
        out += b.isize;
        blocks.push_back(b);
        pos += total;
    }
    *out_total = out;
    return !blocks.empty();
}

// false = not BGZF, or a member is not what its header and trailer say: the general reader (and then zlib) judges the file.
bool inflate_bgzf(const unsigned char* in, size_t flen, Text& t, int32_t threads) {
    size_t hl = 0, tot = 0;
    if (!bgzf_header(in, flen, &hl, &tot)) return false;
    std::vector<BgzfBlock> blocks;
    size_t total = 0;
    if (!bgzf_index(in, flen, blocks, &total)) return false;
    t.heap = big_alloc(total + 16, &t.heap_cap);
    if (!t.heap) return false;
    std::atomic<size_t> next{0};
    std::atomic<int> bad{0};
    const size_t ROUND = 1024;                                    // blocks per round (<= 64 MiB of text): the thread count is
    while (next.load() < blocks.size() && !bad.load()) {          // taken anew every round -- cores that other files' readers
        const size_t r0 = next.load(), r1 = std::min(blocks.size(), r0 + ROUND);   // give back join in
        const int T = (int)std::max<size_t>(1, std::min<size_t>((size_t)(threads > 0 ? threads : reader_share()), (r1 - r0 + 15) / 16));
        std::atomic<size_t> cur{r0};
        auto work = [&] {
            std::vector<unsigned char> scratch(65536 + 512);      // a symbol's writes may run past the block's end: not into a neighbour
            for (;;) {
                const size_t k0 = cur.fetch_add(16);
                if (k0 >= r1 || bad.load()) return;
                for (size_t k = k0; k < std::min(r1, k0 + 16); ++k) {
                    const BgzfBlock& b = blocks[k];
                    epginflate::Out o{scratch.data(), 0, 65536};
                    auto no_grow = [](size_t) { return false; };
                    const size_t used = epginflate::inflate_raw(in + b.in_off, b.in_len, o, no_grow);
                    // (inflate_raw returns 0 for "not a deflate stream": a member whose payload is EMPTY must not pass as 0 == 0 --
                    // raw deflate needs two bytes at least, the BGZF end-of-file block has exactly two -- zlib rejects it too)
                    if (b.in_len < 2 || used == 0 || used != b.in_len || o.pos != b.isize || epgcrc::crc32_fast(0, scratch.data(), b.isize) != b.crc) { bad.store(1); return; }
                    memcpy(t.heap + b.out_off, scratch.data(), b.isize);
                }
            }
        };
        if (T == 1) work();
        else {
            Uncount uncount_;
            std::vector<std::thread> th;
            for (int w = 0; w < T; ++w) th.emplace_back([&] { Census census_; work(); });
            join_all(th);
        }
        next.store(r1);
    }
    if (bad.load()) return false;
    memset(t.heap + total, 0, 16);
    t.data = t.heap;
    t.size = total;
    return true;
}

// zlib's inflate over the whole (possibly multi-member) gzip file in memory; the fallback of inflate_own and the reference the
// differential fuzz compares it with.
bool inflate_zlib(const unsigned char* in, size_t flen, size_t cap, Text& t, const char* path) {
    t.heap = big_alloc(cap + 16, &t.heap_cap);
    if (!t.heap) { fail("out of memory reading %s", path); return false; }
    cap = t.heap_cap - 16;
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, 15 + 16) != Z_OK) { fail("zlib init failed"); return false; }
    size_t in_pos = 0, out_pos = 0;
    for (;;) {
        if (cap - out_pos < (1u << 20)) {
            const size_t ncap = cap + cap / 2 + (1u << 24);
            size_t got = 0;
            char* nh = big_grow(t.heap, t.heap_cap, out_pos, ncap + 16, &got);
            if (!nh) { inflateEnd(&zs); fail("out of memory reading %s", path); return false; }
            t.heap = nh;
            t.heap_cap = got;
            cap = got - 16;
        }
        zs.next_in = const_cast<Bytef*>(in + in_pos);
        zs.avail_in = (uInt)std::min<size_t>(flen - in_pos, 1u << 30);
        zs.next_out = (Bytef*)t.heap + out_pos;
        zs.avail_out = (uInt)std::min<size_t>(cap - out_pos, 1u << 30);
        const size_t in0 = zs.avail_in, out0 = zs.avail_out;
        const int rc = inflate(&zs, Z_NO_FLUSH);
        in_pos += in0 - zs.avail_in;
        out_pos += out0 - zs.avail_out;
        if (rc == Z_STREAM_END) {
            if (in_pos >= flen || !is_gzip(in + in_pos, flen - in_pos)) break;      // last member (trailing garbage is ignored like gzip does)
            inflateReset(&zs);                                                          // next member of a multi-member file
            continue;
        }
        if (rc != Z_OK && rc != Z_BUF_ERROR) { inflateEnd(&zs); fail("read error in %s: %s", path, zs.msg ? zs.msg : "corrupt gzip data"); return false; }
        if (rc == Z_BUF_ERROR && zs.avail_in == 0 && in_pos >= flen) { inflateEnd(&zs); fail("read error in %s: truncated gzip data", path); return false; }
    }
    inflateEnd(&zs);
    memset(t.heap + out_pos, 0, 16);
    t.data = t.heap;
    t.size = out_pos;
    return true;
}

bool slurp(const char* path, Text& t, int32_t threads = 0) {
    const int fd = open(path, O_RDONLY);
    if (fd < 0) { fail("cannot open %s", path); return false; }
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); fail("cannot stat %s", path); return false; }
    const size_t flen = (size_t)st.st_size;
    if (flen == 0) { close(fd); t.data = ""; t.size = 0; return true; }
    void* m = mmap(nullptr, flen, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { fail("cannot map %s", path); return false; }
    madvise(m, flen, MADV_SEQUENTIAL);
    t.map = m;
    t.map_len = flen;
    const unsigned char* in = (const unsigned char*)m;
    if (!is_gzip(in, flen)) {
        // plain text: parse in place when the mapping leaves slack after the last byte (it does unless the size is a multiple
        // of the page size), else through a copy
        if (flen % 4096 != 0 && 4096 - flen % 4096 >= 16) { t.data = (const char*)m; t.size = flen; return true; }
        t.heap = big_alloc(flen + 16, &t.heap_cap);
        if (!t.heap) { fail("out of memory reading %s", path); return false; }
        memcpy(t.heap, m, flen);
        memset(t.heap + flen, 0, 16);
        t.data = t.heap; t.size = flen;
        return true;
    }
    size_t cap = (size_t)((uint32_t)in[flen - 4] | (uint32_t)in[flen - 3] << 8 | (uint32_t)in[flen - 2] << 16 | (uint32_t)in[flen - 1] << 24);
    if (cap < flen) cap = flen * 4;                     // ISIZE is the LAST member's size mod 2^32: a hint only
    cap += 64;
    // The library's own inflate first (epg_inflate.h, ~2-3x zlib's rate); every member's ISIZE and CRC-32 are checked, and on
    // any doubt the file is read again with zlib below.  EPGIO_INFLATE=zlib skips it.
    {
        static const bool use_own = [] { const char* e = getenv("EPGIO_INFLATE"); return !(e && e[0] == 'z'); }();
        if (use_own && inflate_bgzf(in, flen, t, threads)) return true;      // blocked gzip: the members in parallel
        big_free(t.heap, t.heap_cap);
        t.heap = nullptr;
        t.heap_cap = 0;
        if (use_own && inflate_own(in, flen, cap, t, threads)) return true;
        big_free(t.heap, t.heap_cap);
        t.heap = nullptr;
        t.heap_cap = 0;
    }
    return inflate_zlib(in, flen, cap, t, path);
}

}  // namespace

#if defined(__x86_64__)
// One row's states with AVX-512 (VBMI2): 64 bytes of "\t18\t7\t18..." per step.  Every byte position is treated as if a value
// started behind it -- first digit d0 = B[i+1] - '0', second d1 = B[i+2] - '0', value = two digits ? 10 d0 + d1 : d0 -- and the
// results at the TAB positions are packed to the front (vpcompressb) and stored: ~1 cycle per value where the scalar loop
// below spends ~12 (a data-dependent branch on one or two digits).  Anything that is not a plain row of one- or two-digit
// values separated by single tabs (a sign, three digits, an empty field, other characters, a '\r' that is not the last
// byte) makes it return -1 and the scalar code parses -- and judges -- the row.  Loads are masked to the row (through its
// '\n'), so nothing behind it is touched.  q = the tab in front of the first value, nl = the row's '\n'.
__attribute__((target("avx512f,avx512bw,avx512vbmi2,bmi2,popcnt")))
int parse_row_avx512(const char* q, const char* nl, int8_t* out, int cols, __m512i& vmin, __m512i& vmax, int max0) {
    const __m512i c_tab = _mm512_set1_epi8('\t'), c_nl = _mm512_set1_epi8('\n'), c_cr = _mm512_set1_epi8('\r');
    const __m512i c_0 = _mm512_set1_epi8('0'), c_9 = _mm512_set1_epi8(9), c_30 = _mm512_set1_epi8((char)max0), c_1 = _mm512_set1_epi8(1);
    const __m512i c_ff = _mm512_set1_epi8((char)0xff);
    const __m512i lut10 = _mm512_broadcast_i32x4(_mm_setr_epi8(0, 10, 20, 30, 40, 50, 60, 70, 80, 90, 0, 0, 0, 0, 0, 0));
    int c = 0;
    while (q < nl) {
        const long left = nl - q;                                  // bytes before the '\n'
        const unsigned n = left < 64 ? (unsigned)left : 64u;
        const unsigned long long in_row = _bzhi_u64(~0ull, n);
        const long valid = left + 1;                               // readable: through the '\n'
        const __m512i b0 = _mm512_maskz_loadu_epi8(_bzhi_u64(~0ull, (unsigned)(valid > 64 ? 64 : valid)), q);
        const __m512i b1 = _mm512_maskz_loadu_epi8(_bzhi_u64(~0ull, (unsigned)(valid - 1 > 64 ? 64 : valid - 1)), q + 1);
        const __m512i b2 = _mm512_maskz_loadu_epi8(_bzhi_u64(~0ull, (unsigned)(valid - 2 > 64 ? 64 : (valid - 2 < 0 ? 0 : valid - 2))), q + 2);
        const __m512i b3 = _mm512_maskz_loadu_epi8(_bzhi_u64(~0ull, (unsigned)(valid - 3 > 64 ? 64 : (valid - 3 < 0 ? 0 : valid - 3))), q + 3);
        const unsigned long long tab = _mm512_cmpeq_epi8_mask(b0, c_tab) & in_row;
        const __m512i d0 = _mm512_sub_epi8(b1, c_0), d1 = _mm512_sub_epi8(b2, c_0);
        const unsigned long long dig_here = _mm512_cmple_epu8_mask(_mm512_sub_epi8(b0, c_0), c_9);
        const unsigned long long cr = _mm512_cmpeq_epi8_mask(b0, c_cr) & in_row;
        if ((~(tab | dig_here | cr)) & in_row) return -1;          // some other character
        if (cr && (cr != (1ull << (n - 1)) || left > 64)) return -1;   // a '\r' anywhere but right in front of the '\n'
        const unsigned long long dig0 = _mm512_cmple_epu8_mask(d0, c_9), two = _mm512_cmple_epu8_mask(d1, c_9);
        const unsigned long long t2 = _mm512_cmpeq_epi8_mask(b2, c_tab) | _mm512_cmpeq_epi8_mask(b2, c_nl) | _mm512_cmpeq_epi8_mask(b2, c_cr);
        const unsigned long long t3 = _mm512_cmpeq_epi8_mask(b3, c_tab) | _mm512_cmpeq_epi8_mask(b3, c_nl) | _mm512_cmpeq_epi8_mask(b3, c_cr);
        const unsigned long long ok = dig0 & ((~two & t2) | (two & t3));
        if (tab & ~ok) return -1;                                   // empty field, sign, three digits, ...
        const __m512i v2 = _mm512_add_epi8(_mm512_shuffle_epi8(lut10, d0), d1);
        const __m512i v = _mm512_mask_blend_epi8(two, d0, v2);     // the value as written (1-based)
        vmin = _mm512_mask_min_epu8(vmin, tab, vmin, v);
        vmax = _mm512_mask_max_epu8(vmax, tab, vmax, v);
        __m512i o = _mm512_sub_epi8(v, c_1);                        // 0-based; 0 wraps to 255
        o = _mm512_mask_mov_epi8(o, _mm512_cmpgt_epu8_mask(o, c_30), c_ff);   // outside 0..max0: "not a state" (-1)
        const int cnt = (int)_mm_popcnt_u64(tab);
        if (c + cnt > cols) return -1;
        _mm512_mask_storeu_epi8(out + c, _bzhi_u64(~0ull, (unsigned)cnt), _mm512_maskz_compress_epi8(tab, o));
        c += cnt;
        q += n;
    }
    return c;
}

bool have_avx512_parser() {
    static const bool ok = [] {
        const char* e = getenv("EPGIO_SIMD");
        if (e && e[0] == '0') return false;
        return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vbmi2") &&
               __builtin_cpu_supports("bmi2");
    }();
    return ok;
}
#endif

#if defined(__x86_64__)
__attribute__((target("avx512f,avx512bw")))
void init_minmax512(void* mn, void* mx, int T) {
    for (int i = 0; i < T; ++i) {
        _mm512_store_si512((char*)mn + 64 * i, _mm512_set1_epi8((char)0xff));
        _mm512_store_si512((char*)mx + 64 * i, _mm512_setzero_si512());
    }
}
// smallest / largest value byte a thread's vector parser saw, folded into its scalar range
void fold_minmax512(const void* mn, const void* mx, int T, int* vlo, int* vhi) {
    for (int i = 0; i < T; ++i) {
        const unsigned char* a = (const unsigned char*)mn + 64 * i;
        const unsigned char* b = (const unsigned char*)mx + 64 * i;
        int lo = 255, hi = 0;
        for (int k = 0; k < 64; ++k) { lo = a[k] < lo ? a[k] : lo; hi = b[k] > hi ? b[k] : hi; }
        if (lo <= hi) {                                            // the vector parser saw at least one value
            if (lo < vlo[i]) vlo[i] = lo;
            if (hi > vhi[i]) vhi[i] = hi;
        }
    }
}
#endif

struct epgio_table {
    int64_t rows = 0;
    int32_t cols = 0;
    std::unique_ptr<int8_t[]> states;   // [rows * cols], not zero-filled; empty when the caller supplied the destination
    const int8_t* ext = nullptr;        // the caller's destination (epgio_open_table_into), row pitch ext_ld
    int64_t ext_ld = 0;
    std::vector<char> loc;          // concatenated "chr\tstart\tend"
    std::vector<int64_t> loc_off;   // [rows + 1]
    int32_t state_lo = 0, state_hi = 0;   // smallest / largest state value as written in the file (1-based); 0, 0 when empty
};

extern "C" {

const char* epgio_last_error(void) { return g_err; }

int64_t epgio_count_rows(const char* path) {
    Census census_;
    gzFile f = gzopen(path, "rb");
    if (!f) return fail("cannot open %s", path);
    gzbuffer(f, 1 << 20);
    std::vector<char> buf(1 << 22);
    int64_t total = 0;
    for (;;) {
        const int got = gzread(f, buf.data(), (unsigned)buf.size());
        if (got < 0) { gzclose(f); return fail("read error in %s", path); }
        if (got == 0) break;
        const char* p = buf.data();
        const char* e = p + got;
        while ((p = (const char*)memchr(p, '\n', (size_t)(e - p)))) { ++total; ++p; }
    }
    gzclose(f);
    (void)ends_with_gz;
    return total;
}

static double now_s() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

epgio_table* epgio_open_table(const char* path, int64_t row_lo, int64_t row_hi, int32_t threads) {
    return epgio_open_table_ex(path, row_lo, row_hi, threads, 31);
}

void epgio_thread_census(int32_t* live, int32_t* peak, int32_t reset) {
    if (live) *live = g_live.load();
    if (peak) *peak = g_peak.load();
    if (reset) g_peak.store(g_live.load());
}

int32_t epgio_default_threads(void) { return n_threads(0); }
void epgio_set_host_threads(int32_t n) { g_host_threads.store(n > 0 ? n : 0, std::memory_order_relaxed); }


int64_t epgio_inflate_mem(const void* in, int64_t n, void* out, int64_t cap, int32_t own) {
    if (!in || n < 0 || cap < 0 || (cap > 0 && !out)) return fail("inflate_mem: bad argument");
    const unsigned char* p = (const unsigned char*)in;
    if (!is_gzip(p, (size_t)n)) return fail("inflate_mem: not a gzip stream");
    Text t;
    size_t hint = (size_t)((uint32_t)p[n - 4] | (uint32_t)p[n - 3] << 8 | (uint32_t)p[n - 2] << 16 | (uint32_t)p[n - 1] << 24);
    if (hint < (size_t)n || hint > ((size_t)1 << 28)) hint = (size_t)n * 4;
    hint += 64;
    const bool ok = own == 2 ? inflate_bgzf(p, (size_t)n, t, 2) : own ? inflate_own(p, (size_t)n, hint, t, 1) : inflate_zlib(p, (size_t)n, hint, t, "memory");
    if (!ok) return own ? fail("inflate_mem: the library's inflate declined the stream") : -1;
    if ((int64_t)t.size > cap) return fail("inflate_mem: output %lld > cap %lld", (long long)t.size, (long long)cap);
    if (t.size) memcpy(out, t.data, t.size);
    return (int64_t)t.size;
}

static epgio_table* open_table_impl(const char* path, int64_t row_lo, int64_t row_hi, int32_t threads, int32_t max_state,
                                    epgio_alloc_fn alloc, void* user);

epgio_table* epgio_open_table_ex(const char* path, int64_t row_lo, int64_t row_hi, int32_t threads, int32_t max_state) {
    return open_table_impl(path, row_lo, row_hi, threads, max_state, nullptr, nullptr);
}

epgio_table* epgio_open_table_into(const char* path, int64_t row_lo, int64_t row_hi, int32_t threads, int32_t max_state,
                                   epgio_alloc_fn alloc, void* user) {
    if (!alloc) { fail("open_table_into: no allocator"); return nullptr; }
    return open_table_impl(path, row_lo, row_hi, threads, max_state, alloc, user);
}

extern "C" void epgio_set_reader_plan(int32_t n) { g_reader_plan.store(n > 0 ? n : 0); }

// The kept text buffers (their pages are released after every file already) go back to the kernel -- on a thread of its own
// when `background`.  For a process that is done reading and goes on living.
extern "C" void epgio_release_buffers(int32_t background) {
    std::vector<std::pair<char*, size_t>> bufs;
    {
        BigBufs& b = big_bufs();
        std::lock_guard<std::mutex> g(b.m);
        bufs.swap(b.free_list);
    }
    if (bufs.empty()) return;
    auto work = [bufs] { for (auto& e : bufs) free(e.first); };
    if (background) std::thread(work).detach();
    else work();
}
struct ReaderScope {
    ReaderScope() { g_readers.fetch_add(1); }
    ~ReaderScope() { g_readers.fetch_sub(1); }
};

static epgio_table* open_table_impl(const char* path, int64_t row_lo, int64_t row_hi, int32_t threads, int32_t max_state,
                                    epgio_alloc_fn alloc, void* user) {
    Census census_;                                           // the caller: the inflate of a file is one serial thread
    ReaderScope reader_;
    const int max0 = (max_state > 31 ? 127 : 31) - 1;        // largest 0-based state kept: two classes, like the kernels
    static const bool timing = getenv("EPGIO_TIMING") != nullptr;
    double t0 = now_s();
    const char* fname = strrchr(path, '/') ? strrchr(path, '/') + 1 : path;
    auto lap = [&](const char* what) {            // EPGIO_TIMING: the phase, its file, its time and when it ended (CLOCK_MONOTONIC)
        if (timing) {
            const double t1 = now_s();
            fprintf(stderr, "    [epgio] %-26s %-26s %7.3f s  ends %.3f\n", fname, what, t1 - t0, t1);
            t0 = t1;
        }
    };
    Text buf;
    if (!slurp(path, buf, threads)) return nullptr;
    lap("read / inflate");
    const char* base = buf.data;
    const char* end = base + buf.size;
    // complete lines only (the reference counts newlines, helpers.py:94: a dangling last line does not exist)
    const char* last_nl = nullptr;
    for (const char* p = end; p > base;) { --p; if (*p == '\n') { last_nl = p; break; } }
    if (!last_nl) { auto* t = new epgio_table(); t->loc_off.assign(1, 0); return t; }
    end = last_nl + 1;

    const int T = threads > 0 ? threads : reader_share();
    // segment borders on line starts
    std::vector<const char*> seg(T + 1);
    seg[0] = base;
    for (int i = 1; i < T; ++i) {
        const char* p = base + (size_t)((end - base) / T) * i;
        if (p < seg[i - 1]) p = seg[i - 1];
        const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
        seg[i] = nl ? nl + 1 : end;
    }
    seg[T] = end;
    std::vector<int64_t> seg_rows(T, 0);
    {
        Uncount uncount_;
        std::vector<std::thread> th;
        for (int i = 0; i < T; ++i)
            th.emplace_back([&, i] {
            Census census_;
                int64_t n = 0;
                const char* p = seg[i];
                while (p < seg[i + 1] && (p = (const char*)memchr(p, '\n', (size_t)(seg[i + 1] - p)))) { ++n; ++p; }
                seg_rows[i] = n;
            });
        join_all(th);
    }
    lap("count lines");
    std::vector<int64_t> seg_first(T + 1, 0);
    for (int i = 0; i < T; ++i) seg_first[i + 1] = seg_first[i] + seg_rows[i];
    const int64_t total = seg_first[T];
    if (row_hi < 0 || row_hi > total) row_hi = total;
    if (row_lo < 0) row_lo = 0;
    if (row_lo > row_hi) row_lo = row_hi;

    // number of state columns from the first line
    int32_t cols = 0;
    {
        const char* nl = (const char*)memchr(base, '\n', (size_t)(end - base));
        int tabs = 0;
        for (const char* p = base; p < nl; ++p) tabs += *p == '\t';
        cols = tabs - 2;
        if (cols < 1) { fail("%s: expected at least 4 tab-separated columns, found %d", path, tabs + 1); return nullptr; }
    }
    auto* t = new epgio_table();
    t->rows = row_hi - row_lo;
    t->cols = cols;
    int8_t* sbase = nullptr;
    int64_t sld = cols;
    if (alloc) {
        // the caller's destination (a pinned, row-padded staging buffer): rows are parsed straight into it -- no 1 GB intermediate
        // matrix, no copy of it (round 3: both, single-threaded, after the parse)
        int64_t ld = 0;
        sbase = alloc(t->rows, cols, &ld, user);
        if (!sbase || ld < cols) { fail("%s: no destination for %lld x %d states", path, (long long)t->rows, cols); delete t; return nullptr; }
        sld = ld;
        t->ext = sbase;
        t->ext_ld = ld;
    } else {
        t->states.reset(new int8_t[(size_t)t->rows * cols + 1]);
        advise_huge(t->states.get(), (size_t)t->rows * cols);
        sbase = t->states.get();
    }
    t->loc_off.assign((size_t)t->rows + 1, 0);

    // pass 1: location text lengths; pass 2 (after prefix sum): states + location text.  Both per segment.
    std::vector<int> err(T, 0);
    std::vector<int64_t> bad_row(T, -1);
    std::vector<int> vlo(T, 1 << 30), vhi(T, -(1 << 30));
    auto for_rows = [&](int i, auto&& fn) {
        const char* p = seg[i];
        int64_t r = seg_first[i];
        while (p < seg[i + 1]) {
            const char* nl = (const char*)memchr(p, '\n', (size_t)(seg[i + 1] - p));
            if (r >= row_hi) break;
            if (r >= row_lo) fn(r - row_lo, p, nl);
            p = nl + 1;
            ++r;
        }
    };
    {
        Uncount uncount_;
        std::vector<std::thread> th;
        for (int i = 0; i < T; ++i)
            th.emplace_back([&, i] {
            Census census_;
                for_rows(i, [&](int64_t r, const char* p, const char* nl) {
                    int tabs = 0;
                    const char* q = p;
                    for (; q < nl; ++q)
                        if (*q == '\t' && ++tabs == 3) break;
                    t->loc_off[(size_t)r + 1] = q - p + 1; // length (+ the '\n' that terminates the row's text) for now
                });
            });
        join_all(th);
    }
    lap("location lengths");
    for (int64_t r = 0; r < t->rows; ++r) t->loc_off[(size_t)r + 1] += t->loc_off[(size_t)r];
    t->loc.resize((size_t)t->loc_off[(size_t)t->rows]);
    {
#if defined(__x86_64__)
        struct alignas(64) V512 { __m512i v; };
        const bool simd = have_avx512_parser();
        std::vector<V512> vmin512((size_t)T), vmax512((size_t)T);
        if (simd) init_minmax512(vmin512.data(), vmax512.data(), T);
#endif
        Uncount uncount_;
        std::vector<std::thread> th;
        for (int i = 0; i < T; ++i)
            th.emplace_back([&, i] {
            Census census_;
                for_rows(i, [&](int64_t r, const char* p, const char* nl) {
                    const int64_t len = t->loc_off[(size_t)r + 1] - t->loc_off[(size_t)r] - 1;
                    memcpy(t->loc.data() + t->loc_off[(size_t)r], p, (size_t)len);
                    t->loc[(size_t)(t->loc_off[(size_t)r] + len)] = '\n';
                    const char* q = p + len;
                    int8_t* out = sbase + (size_t)r * (size_t)sld;
                    if (sld > cols) memset(out + cols, 0xff, (size_t)(sld - cols));    // pad bytes: not a state
                    int c = 0;
#if defined(__x86_64__)
                    if (simd) {
                        if (parse_row_avx512(q, nl, out, cols, vmin512[i].v, vmax512[i].v, max0) == cols) return;
                        // anything unusual: the scalar code below parses the row again from its first value
                    }
#endif
                    // fast path: one- or two-digit values ("\t7", "\t18"), which is every state of a 1..31-state model; the
                    // text has 16 readable bytes after its end, a row ends in '\n' (not a digit), so looking three characters
                    // ahead is safe.  Anything else (sign, three digits, '\r', malformed) leaves the loop for the general one.
                    {
                        int lo = vlo[i], hi = vhi[i];
                        while (c < cols && *q == '\t') {
                            const unsigned d0 = (unsigned)(unsigned char)q[1] - '0';
                            if (d0 > 9) break;
                            const unsigned d1 = (unsigned)(unsigned char)q[2] - '0';
                            int v;
                            if (d1 > 9) { v = (int)d0; q += 2; }
                            else {
                                if ((unsigned)(unsigned char)q[3] - '0' <= 9) break;       // three or more digits
                                v = (int)(d0 * 10 + d1);
                                q += 3;
                            }
                            lo = v < lo ? v : lo;
                            hi = v > hi ? v : hi;
                            --v;                                                        // file states are 1-based (helpers.py:155)
                            out[c++] = (int8_t)(((unsigned)v > (unsigned)max0) ? -1 : v);  // outside 0..max0: "not a state" (see below)
                        }
                        vlo[i] = lo; vhi[i] = hi;
                        if (c == cols && q < nl && *q == '\r') ++q;                    // CRLF line ends
                    }
                    while (q < nl && c < cols) {
                        // a value is introduced by a TAB and by nothing else: the fast path above can stop on any byte
                        // ("1.5", "1x5", "1 5"), and that byte must not be swallowed as if it were the separator
                        if (*q != '\t') { err[i] = 1; if (bad_row[i] < 0) bad_row[i] = r; return; }
                        ++q;                                // the tab before the value
                        int v = 0;
                        bool neg = false, any = false;
                        if (q < nl && *q == '-') { neg = true; ++q; }
                        while (q < nl && *q >= '0' && *q <= '9') { v = v * 10 + (*q - '0'); ++q; any = true; if (v > 100000) break; }
                        if (q < nl && *q == '\r') ++q;
                        if (!any || (q < nl && *q != '\t')) { err[i] = 1; if (bad_row[i] < 0) bad_row[i] = r; return; }
                        v = neg ? -v : v;
                        if (v < vlo[i]) vlo[i] = v;
                        if (v > vhi[i]) vhi[i] = v;
                        v -= 1;                             // file states are 1-based (helpers.py:155)
                        // the kernels of models up to 31 states decode five bits and treat 31 as "not a state": anything
                        // outside 0..max0 (30, or 126 for the wide models) is stored as -1 so that it cannot alias a state; the caller sees it in epgio_table_state_range and in the
                        // count check (sum of counts != rows * columns)
                        out[c++] = (int8_t)((v < 0 || v > max0) ? -1 : v);
                    }
                    if (c != cols || q != nl) { err[i] = 1; if (bad_row[i] < 0) bad_row[i] = r; }
                });
            });
        join_all(th);
#if defined(__x86_64__)
        if (simd) fold_minmax512(vmin512.data(), vmax512.data(), T, vlo.data(), vhi.data());
#endif
    }
    lap("parse states + locations");
    for (int i = 0; i < T; ++i) {
        if (vlo[i] <= vhi[i]) {
            if (t->state_lo == 0 && t->state_hi == 0) { t->state_lo = vlo[i]; t->state_hi = vhi[i]; }
            t->state_lo = std::min(t->state_lo, vlo[i]);
            t->state_hi = std::max(t->state_hi, vhi[i]);
        }
    }
    for (int i = 0; i < T; ++i)
        if (err[i]) {
            fail("%s: malformed line at row %lld (expected %d integer state columns)", path, (long long)(bad_row[i] + row_lo), cols);
            delete t;
            return nullptr;
        }
    return t;
}

int64_t epgio_table_rows(const epgio_table* t) { return t ? t->rows : fail("null table"); }
int32_t epgio_table_cols(const epgio_table* t) { return t ? t->cols : fail("null table"); }

int epgio_table_state_range(const epgio_table* t, int32_t* lo, int32_t* hi) {
    if (!t) return fail("null table");
    if (lo) *lo = t->state_lo;
    if (hi) *hi = t->state_hi;
    return 0;
}

int epgio_table_copy_states(const epgio_table* t, int8_t* out, int64_t ldx) {
    if (!t || !out || ldx < t->cols) return fail("copy_states: bad argument");
    if (t->ext) return t->ext == out && t->ext_ld == ldx ? 0 : fail("copy_states: the states were parsed into the caller's own destination");
    for (int64_t r = 0; r < t->rows; ++r) {
        memcpy(out + r * ldx, t->states.get() + (size_t)r * t->cols, (size_t)t->cols);
        if (ldx > t->cols) memset(out + r * ldx + t->cols, 0xff, (size_t)(ldx - t->cols));
    }
    return 0;
}

const char* epgio_table_locations(const epgio_table* t, const int64_t** offsets) {
    if (!t) { fail("null table"); return nullptr; }
    if (offsets) *offsets = t->loc_off.data();
    return t->loc.data();
}

void epgio_close_table(epgio_table* t) { delete t; }

}  // extern "C"

namespace {

// Exact "%.5f" of a float32: round-half-even of the exact binary value times 1e5 (what Python's float formatting does
// for float(v)), sign kept for -0.0 and for negatives that round to zero.
inline char* fmt_f5(float f, char* o) {
    uint32_t u;
    memcpy(&u, &f, 4);
    const bool neg = u >> 31;
    const int ex = (int)((u >> 23) & 0xff);
    const uint32_t man = u & 0x7fffffu;
    if (ex == 255) {
        if (man) { memcpy(o, "nan", 3); return o + 3; }
        if (neg) *o++ = '-';
        memcpy(o, "inf", 3);
        return o + 3;
    }
    const uint64_t m = ex ? (man | 0x800000u) : man;
    const int e = ex ? ex - 150 : -149;
    uint64_t q64 = 0;
    bool wide = false;
    unsigned __int128 q = 0;
    if (e >= 0) {
        if (e > 80) { return o + sprintf(o, "%.5f", (double)f); }
        if (e <= 22) q64 = (m * 100000ull) << e;                // < 2^41 * 2^22: fits
        else { q = ((unsigned __int128)(m * 100000ull)) << e; wide = true; }
    } else {
        const int sh = -e;
        const uint64_t num = m * 100000ull;                 // < 2^41
        if (sh >= 64) {
            q64 = 0;                                        // num < 2^41 <= half of 2^sh: rounds to 0
        } else {
            uint64_t qq = num >> sh;
            const uint64_t rem = num & ((1ull << sh) - 1), half = 1ull << (sh - 1);
            if (rem > half || (rem == half && (qq & 1))) ++qq;
            q64 = qq;
        }
    }
    if (neg) *o++ = '-';
    uint64_t frac;
    if (!wide) {                                            // every score this code has seen: 64-bit division, no __udivti3
        frac = q64 % 100000u;
        uint64_t ip = q64 / 100000u;
        if (ip < 10) {
            *o++ = (char)('0' + (int)ip);
        } else {
            char tmp[24];
            int n = 0;
            while (ip) { tmp[n++] = (char)('0' + (int)(ip % 10)); ip /= 10; }
            while (n) *o++ = tmp[--n];
        }
    } else {
        frac = (uint64_t)(q % 100000u);
        unsigned __int128 ip = q / 100000u;
        char tmp[48];
        int n = 0;
        if (ip == 0) tmp[n++] = '0';
        while (ip) { tmp[n++] = (char)('0' + (int)(ip % 10)); ip /= 10; }
        while (n) *o++ = tmp[--n];
    }
    *o++ = '.';
    o[4] = (char)('0' + frac % 10); o[3] = (char)('0' + frac / 10 % 10); o[2] = (char)('0' + frac / 100 % 10);
    o[1] = (char)('0' + frac / 1000 % 10); o[0] = (char)('0' + frac / 10000 % 10);
    return o + 5;
}

// EPILOGOS_BGZF=1: the writers emit BGZF (blocks of <= 65 280 bytes of text, each a gzip member with its compressed size in a 'BC'
// extra subfield, an empty block as end marker) -- the same decompressed bytes, readable by gzip and zlib like any multi-member
// file, and by bgzip / tabix with random access; this library's reader inflates such a file's blocks in parallel.
bool bgzf_on() { const char* e = getenv("EPILOGOS_BGZF"); return e && e[0] && e[0] != '0'; }
const unsigned char BGZF_EOF[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};

bool bgzf_blocks(const char* in, size_t n, int level, std::vector<unsigned char>& out) {
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (deflateInit2(&zs, level >= 1 ? level : 1, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
    out.clear();
    out.reserve(n / 3 + 64);
    unsigned char blk[65536 + 64];
    bool ok = true;
    for (size_t off = 0; off < n && ok; off += 65280) {
        const size_t len = std::min<size_t>(65280, n - off);
        deflateReset(&zs);
        zs.next_in = (Bytef*)(in + off);
        zs.avail_in = (uInt)len;
        zs.next_out = blk + 18;
        zs.avail_out = 65536 - 18 - 8;
        if (deflate(&zs, Z_FINISH) != Z_STREAM_END) { ok = false; break; }
        const size_t total = 18 + (size_t)zs.total_out + 8;
        static const unsigned char hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
        memcpy(blk, hdr, 16);
        blk[16] = (unsigned char)((total - 1) & 0xff);
        blk[17] = (unsigned char)((total - 1) >> 8);
        const uint32_t crc = epgcrc::crc32_fast(0, (const unsigned char*)in + off, len), isize = (uint32_t)len;
        memcpy(blk + total - 8, &crc, 4);
        memcpy(blk + total - 4, &isize, 4);
        out.insert(out.end(), blk, blk + total);
    }
    deflateEnd(&zs);
    return ok;
}

// level 1..9: zlib; level 0: the writer's own fast compressor (epg_deflate.h)
bool gzip_member(const std::vector<char>& in, int level, std::vector<unsigned char>& out) {
    if (bgzf_on()) return bgzf_blocks(in.data(), in.size(), level, out);
    if (level == 0) {
        epgdeflate::gzip_member_fast(reinterpret_cast<const unsigned char*>(in.data()), in.size(), out);
        return true;
    }
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (deflateInit2(&zs, level, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
    out.resize(deflateBound(&zs, (uLong)in.size()) + 64);
    zs.next_in = (Bytef*)in.data();
    zs.avail_in = (uInt)in.size();
    zs.next_out = out.data();
    zs.avail_out = (uInt)out.size();
    const int rc = deflate(&zs, Z_FINISH);
    out.resize(zs.total_out);
    deflateEnd(&zs);
    return rc == Z_STREAM_END;
}

}  // namespace

extern "C" {

int epgio_parse_locations(const char* loc, const int64_t* loc_off, int64_t R, int64_t* start, int64_t* end, int32_t* same_chrom,
                          int32_t threads) {
    if (R < 0 || (R > 0 && (!loc || !loc_off || !start || !end))) return fail("parse_locations: bad argument");
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads(threads), (R + 65535) / 65536));
    std::vector<int> bad(T, 0), same(T, 1);
    const char* c0 = loc + (R ? loc_off[0] : 0);
    const char* c0e = R ? (const char*)memchr(c0, '\t', (size_t)(loc_off[1] - loc_off[0])) : c0;
    const size_t c0n = c0e ? (size_t)(c0e - c0) : 0;
    Uncount uncount_;
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
        th.emplace_back([&, t] {
            Census census_;
            for (int64_t r = R * t / T; r < R * (t + 1) / T; ++r) {
                const char* p = loc + loc_off[r];
                const char* e = loc + loc_off[r + 1];
                const char* tab = (const char*)memchr(p, '\t', (size_t)(e - p));
                if (!tab) { bad[t] = 1; return; }
                if ((size_t)(tab - p) != c0n || memcmp(p, c0, c0n) != 0) same[t] = 0;
                int64_t v[2];
                const char* q = tab + 1;
                for (int k = 0; k < 2; ++k) {
                    int64_t x = 0;
                    bool any = false, neg = false;
                    if (q < e && *q == '-') { neg = true; ++q; }
                    while (q < e && *q >= '0' && *q <= '9') { x = x * 10 + (*q - '0'); ++q; any = true; }
                    if (!any) { bad[t] = 1; return; }
                    v[k] = neg ? -x : x;
                    if (k == 0) { if (q >= e || *q != '\t') { bad[t] = 1; return; } ++q; }
                }
                if (q < e && *q == '\r') ++q;
                if (q >= e || *q != '\n') { bad[t] = 1; return; }   // anything else (a float, a third tab): not plain integers
                start[r] = v[0];
                end[r] = v[1];
            }
        });
    join_all(th);
    int all_same = 1;
    for (int t = 0; t < T; ++t) {
        if (bad[t]) return fail("parse_locations: a row is not 'name<TAB>integer<TAB>integer'");
        all_same &= same[t];
    }
    if (same_chrom) *same_chrom = all_same;
    return 0;
}

int epgio_row_sums_f32(const float* a, int64_t R, int32_t S, int64_t lda, float* out, int32_t threads) {
    if (R < 0 || S < 1 || S > 128 || lda < S || (R > 0 && (!a || !out))) return fail("row_sums: bad argument (S must be 1..128)");
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads(threads), (R + 65535) / 65536));
    Uncount uncount_;
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
        th.emplace_back([&, t] {
            Census census_;
            const int n8 = S - S % 8;
            for (int64_t r = R * t / T; r < R * (t + 1) / T; ++r) {
                const float* p = a + r * lda;
                float res;
                if (S < 8) {
                    res = 0.f;
                    for (int i = 0; i < S; ++i) res += p[i];
                } else {
                    float q[8];
                    for (int j = 0; j < 8; ++j) q[j] = p[j];
                    for (int i = 8; i < n8; i += 8)
                        for (int j = 0; j < 8; ++j) q[j] += p[i + j];
                    res = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
                    for (int i = n8; i < S; ++i) res += p[i];
                }
                out[r] = res;
            }
        });
    join_all(th);
    return 0;
}

int epgio_rolling_max_f64(const double* x, int64_t n, int32_t W, double* out, int32_t threads) {
    if (n < 0 || W < 1 || (n > 0 && (!x || !out))) return fail("rolling_max: bad argument");
    const int64_t back = W / 2, fwd = (W - 1) / 2;
    const double nan = std::nan("");
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads(threads), (n + (1 << 20) - 1) >> 20));
    Uncount uncount_;
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
        th.emplace_back([&, t] {
            Census census_;
            const int64_t i0 = n * t / T, i1 = n * (t + 1) / T;
            // monotonic deque of candidate indices (values decreasing; a later equal value replaces an earlier one, like
            // pandas) over the window [i - back, i + fwd], in a power-of-two ring
            size_t ring = 1;
            while (ring < (size_t)W + 1) ring <<= 1;
            const size_t mask = ring - 1;
            std::vector<int64_t> dq(ring);
            size_t head = 0, tail = 0;                               // size = tail - head
            int64_t next = std::max<int64_t>(0, i0 - back);          // next element to push
            for (int64_t i = i0; i < i1; ++i) {
                const int64_t lo = i - back, hi = i + fwd;
                if (lo < 0 || hi >= n) { out[i] = nan; continue; }
                if (next < lo) next = lo;
                for (; next <= hi; ++next) {
                    const double v = x[next];
                    while (tail != head && !(x[dq[(tail - 1) & mask]] > v)) --tail;
                    dq[tail & mask] = next;
                    ++tail;
                }
                while (dq[head & mask] < lo) ++head;
                out[i] = x[dq[head & mask]];
            }
        });
    join_all(th);
    return 0;
}

int64_t epgio_gzip_fast(const void* in, int64_t n, void* out, int64_t cap) {
    if (n < 0 || (n > 0 && !in) || !out || cap < n + n / 8 + 1100) return fail("gzip_fast: bad argument");
    std::vector<unsigned char> z;
    epgdeflate::gzip_member_fast(static_cast<const unsigned char*>(in), (size_t)n, z);
    memcpy(out, z.data(), z.size());
    return (int64_t)z.size();
}

int64_t epgio_format_f5(const float* v, int64_t n, char sep, char* buf, int64_t cap) {
    if (cap < 48 * n) return fail("format_f5: buffer too small");
    char* o = buf;
    for (int64_t i = 0; i < n; ++i) { o = fmt_f5(v[i], o); *o++ = sep; }
    return o - buf;
}

int epgio_write_scores(const char* path, const char* loc, const int64_t* loc_off, const float* scores, int64_t R, int32_t S,
                       int32_t threads, int32_t gzip_level) {
    Census census_;
    if (!path || (R > 0 && (!loc || !loc_off || !scores)) || S < 1) return fail("write_scores: bad argument");
    if (gzip_level < 0 || gzip_level > 9) gzip_level = 6;
    FILE* f = fopen(path, "wb");
    if (!f) return fail("cannot create %s", path);
    const int T = n_threads(threads);
    const int64_t CH = 32768;                                   // rows per gzip member
    const int64_t nchunks = (R + CH - 1) / CH;
    if (R == 0) {                                               // an empty gzip stream, like gzip.open(..).close()
        std::vector<unsigned char> z;
        gzip_member(std::vector<char>(), gzip_level, z);
        fwrite(z.data(), 1, z.size(), f);
    }
    bool ok = true;
    for (int64_t c0 = 0; c0 < nchunks && ok; c0 += T) {
        const int nb = (int)std::min<int64_t>(T, nchunks - c0);
        std::vector<std::vector<unsigned char>> z(nb);
        std::vector<int> good(nb, 0);
        Uncount uncount_;
        std::vector<std::thread> th;
        for (int k = 0; k < nb; ++k)
            th.emplace_back([&, k] {
            Census census_;
                const int64_t r0 = (c0 + k) * CH, r1 = std::min(R, r0 + CH);
                std::vector<char> txt((size_t)((loc_off[r1] - loc_off[r0]) + (r1 - r0) * (2 + 48 * (int64_t)S)));
                char* o = txt.data();
                for (int64_t r = r0; r < r1; ++r) {
                    const int64_t len = loc_off[r + 1] - loc_off[r] - 1;   // without the row's '\n'
                    memcpy(o, loc + loc_off[r], (size_t)len);
                    o += len;
                    const float* row = scores + r * S;
                    for (int s = 0; s < S; ++s) { *o++ = '\t'; o = fmt_f5(row[s], o); }
                    *o++ = '\n';
                }
                txt.resize((size_t)(o - txt.data()));
                good[k] = gzip_member(txt, gzip_level, z[k]);
            });
        join_all(th);
        for (int k = 0; k < nb && ok; ++k) {
            if (!good[k] || fwrite(z[k].data(), 1, z[k].size(), f) != z[k].size()) ok = false;
        }
    }
    if (ok && bgzf_on() && ends_with_gz(path) && fwrite(BGZF_EOF, 1, sizeof(BGZF_EOF), f) != sizeof(BGZF_EOF)) ok = false;
    if (fclose(f) != 0) ok = false;
    return ok ? 0 : fail("write error on %s", path);
}

int epgio_write_states(const char* path, const char* chrom, int64_t start0, int64_t step, const int8_t* states, int64_t R, int32_t N,
                       int64_t ldx, int32_t threads, int32_t gzip_level) {
    Census census_;
    if (!path || !chrom || (R > 0 && !states) || N < 1 || ldx < N || step < 1) return fail("write_states: bad argument");
    if (gzip_level < 0 || gzip_level > 9) gzip_level = 6;
    FILE* f = fopen(path, "wb");
    if (!f) return fail("cannot create %s", path);
    const bool gz = ends_with_gz(path);
    const int T = n_threads(threads);
    const int64_t CH = 8192;                                    // rows per gzip member (~1.7 KB of text per row at N = 833)
    const int64_t nchunks = (R + CH - 1) / CH;
    const size_t clen = strlen(chrom);
    if (R == 0 && gz) {
        std::vector<unsigned char> z;
        gzip_member(std::vector<char>(), gzip_level, z);
        fwrite(z.data(), 1, z.size(), f);
    }
    bool ok = true;
    for (int64_t c0 = 0; c0 < nchunks && ok; c0 += T) {
        const int nb = (int)std::min<int64_t>(T, nchunks - c0);
        std::vector<std::vector<unsigned char>> z(nb);
        std::vector<std::vector<char>> plain(nb);
        std::vector<int> good(nb, 0);
        Uncount uncount_;
        std::vector<std::thread> th;
        for (int k = 0; k < nb; ++k)
            th.emplace_back([&, k] {
            Census census_;
                const int64_t r0 = (c0 + k) * CH, r1 = std::min(R, r0 + CH);
                std::vector<char>& txt = plain[k];
                txt.resize((size_t)(r1 - r0) * (clen + 48 + 5 * (size_t)N));
                char* o = txt.data();
                for (int64_t r = r0; r < r1; ++r) {
                    memcpy(o, chrom, clen);
                    o += clen;
                    o += sprintf(o, "\t%lld\t%lld", (long long)(start0 + r * step), (long long)(start0 + (r + 1) * step));
                    const int8_t* row = states + r * ldx;
                    for (int c = 0; c < N; ++c) {
                        int v = (int)row[c] + 1;                // file states are 1-based
                        *o++ = '\t';
                        if (v < 0) { *o++ = '-'; v = -v; }
                        if (v >= 100) { *o++ = (char)('0' + v / 100); v %= 100; *o++ = (char)('0' + v / 10); *o++ = (char)('0' + v % 10); }
                        else if (v >= 10) { *o++ = (char)('0' + v / 10); *o++ = (char)('0' + v % 10); }
                        else *o++ = (char)('0' + v);
                    }
                    *o++ = '\n';
                }
                txt.resize((size_t)(o - txt.data()));
                good[k] = gz ? gzip_member(txt, gzip_level, z[k]) : 1;
            });
        join_all(th);
        for (int k = 0; k < nb && ok; ++k) {
            if (!good[k]) { ok = false; break; }
            const void* p = gz ? (const void*)z[k].data() : (const void*)plain[k].data();
            const size_t n = gz ? z[k].size() : plain[k].size();
            if (fwrite(p, 1, n, f) != n) ok = false;
        }
    }
    if (ok && bgzf_on() && ends_with_gz(path) && fwrite(BGZF_EOF, 1, sizeof(BGZF_EOF), f) != sizeof(BGZF_EOF)) ok = false;
    if (fclose(f) != 0) ok = false;
    return ok ? 0 : fail("write error on %s", path);
}

int epgio_write_metrics(const char* path, const char* chrom, const int64_t* chrom_off, const int32_t* chrom_idx, const int64_t* start,
                        const int64_t* end, const char* names, const int64_t* names_off, const int32_t* maxdiff, const float* dist,
                        const double* pvals, const double* mh, int64_t R, int32_t threads, int32_t gzip_level) {
    Census census_;
    if (!path || (R > 0 && (!chrom || !chrom_off || !chrom_idx || !start || !end || !names || !names_off || !maxdiff || !dist)) ||
        ((pvals == nullptr) != (mh == nullptr)))
        return fail("write_metrics: bad argument");
    if (gzip_level < 0 || gzip_level > 9) gzip_level = 6;
    FILE* f = fopen(path, "wb");
    if (!f) return fail("cannot create %s", path);
    const int T = n_threads(threads);
    const int64_t CH = 32768;
    const int64_t nchunks = (R + CH - 1) / CH;
    if (R == 0) {
        std::vector<unsigned char> z;
        gzip_member(std::vector<char>(), gzip_level, z);
        fwrite(z.data(), 1, z.size(), f);
    }
    bool ok = true;
    for (int64_t c0 = 0; c0 < nchunks && ok; c0 += T) {
        const int nb = (int)std::min<int64_t>(T, nchunks - c0);
        std::vector<std::vector<unsigned char>> z(nb);
        std::vector<int> good(nb, 0);
        Uncount uncount_;
        std::vector<std::thread> th;
        for (int k = 0; k < nb; ++k)
            th.emplace_back([&, k] {
            Census census_;
                const int64_t r0 = (c0 + k) * CH, r1 = std::min(R, r0 + CH);
                std::vector<char> txt;
                txt.reserve((size_t)(r1 - r0) * 96);
                char num[64];
                for (int64_t r = r0; r < r1; ++r) {
                    const int32_t ci = chrom_idx[r], mi = maxdiff[r] - 1;
                    txt.insert(txt.end(), chrom + chrom_off[ci], chrom + chrom_off[ci + 1]);
                    int n = snprintf(num, sizeof num, "\t%lld\t%lld\t", (long long)start[r], (long long)end[r]);
                    txt.insert(txt.end(), num, num + n);
                    txt.insert(txt.end(), names + names_off[mi], names + names_off[mi + 1]);
                    txt.push_back('\t');
                    const float d = dist[r];
                    char* e = fmt_f5(std::fabs(d), num);                     // "{:.5f}".format(abs(distance)) of the float32
                    txt.insert(txt.end(), num, e);
                    txt.push_back('\t');
                    txt.push_back(d >= 0.0f ? '+' : '-');                   // helpers.findSign: x >= 0
                    if (pvals) {
                        n = snprintf(num, sizeof num, "\t%.5e\t%.5e", pvals[r], mh[r]);
                        txt.insert(txt.end(), num, num + n);
                    }
                    txt.push_back('\n');
                }
                good[k] = gzip_member(txt, gzip_level, z[k]);
            });
        join_all(th);
        for (int k = 0; k < nb && ok; ++k) {
            if (!good[k] || fwrite(z[k].data(), 1, z[k].size(), f) != z[k].size()) ok = false;
        }
    }
    if (ok && bgzf_on() && ends_with_gz(path) && fwrite(BGZF_EOF, 1, sizeof(BGZF_EOF), f) != sizeof(BGZF_EOF)) ok = false;
    if (fclose(f) != 0) ok = false;
    return ok ? 0 : fail("write error on %s", path);
}

}  // extern "C"
