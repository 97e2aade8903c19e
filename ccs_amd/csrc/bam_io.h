// bam_io.h — minimal PacBio-BAM reader / writer for the `ccs` driver (App. B).
//
// BGZF = concatenated gzip members (<= 64 KiB each, BC extra field carries the block size): blocks are
// independent, so inflate / deflate run on a thread pool and are re-ordered (the reference's "-j" threads,
// docs/faq/parallelize.md:17).  Records are unaligned (FLAG 4) PacBio subreads with the tags this path needs:
// zm:i hole number, sn:B,f SNR (A,C,G,T), pw:B,C|S pulse widths, ip:B,C|S, cx:i local context
// (docs/faq/bam-output.md:9-30, docs/faq/missing-adapters.md:11-12).  htslib/pbbam are not in the image;
// zlib is — and so is the runtime library of libdeflate (no headers): when libdeflate.so.0 can be dlopen-ed its whole-buffer
// DEFLATE routines replace zlib's streaming ones for the 64 KiB BGZF blocks (the host side of `ccs` is inflate-bound:
// DESIGN.md §7); CCS_NO_LIBDEFLATE=1 forces zlib.
#pragma once
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <future>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace bamio {

// ---------------------------------------------------------------------------------------------- thread pool
class ThreadPool {
public:
    explicit ThreadPool(int n)
    {
        if (n < 1) n = 1;
        for (int i = 0; i < n; ++i) workers_.emplace_back([this] { run(); });
    }
    ~ThreadPool()
    {
        { std::lock_guard<std::mutex> l(m_); stop_ = true; }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }
    template <class F> auto submit(F f) -> std::future<decltype(f())>
    {
        auto task = std::make_shared<std::packaged_task<decltype(f())()>>(std::move(f));
        auto fut = task->get_future();
        { std::lock_guard<std::mutex> l(m_); q_.emplace_back([task] { (*task)(); }); }
        cv_.notify_one();
        return fut;
    }
    int size() const { return (int)workers_.size(); }

private:
    void run()
    {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [this] { return stop_ || !q_.empty(); });
                if (stop_ && q_.empty()) return;
                f = std::move(q_.front()); q_.pop_front();
            }
            f();
        }
    }
    std::vector<std::thread> workers_;
    std::deque<std::function<void()>> q_;
    std::mutex m_;
    std::condition_variable cv_;
    bool stop_ = false;
};

// ---------------------------------------------------------------------------------------------- BGZF
// libdeflate's public C API ([RECALL] libdeflate.h, stable since 1.0), bound at run time from libdeflate.so.0
struct LibDeflate {
    void *(*alloc_decompressor)(void) = nullptr;
    int (*deflate_decompress)(void *, const void *, size_t, void *, size_t, size_t *) = nullptr;   // 0 = success
    void (*free_decompressor)(void *) = nullptr;
    void *(*alloc_compressor)(int) = nullptr;
    size_t (*deflate_compress)(void *, const void *, size_t, void *, size_t) = nullptr;            // 0 = does not fit
    void (*free_compressor)(void *) = nullptr;
    uint32_t (*crc32_)(uint32_t, const void *, size_t) = nullptr;                                  // optional (carry-less multiply: ~10 GB/s)
    bool ok = false;
    // CRC32 of a BGZF block's payload: libdeflate's when the library exports it, zlib's otherwise
    uint32_t crc(const uint8_t *p, size_t n) const { return crc32_ ? crc32_(0, p, n) : (uint32_t)::crc32(::crc32(0L, Z_NULL, 0), p, (uInt)n); }
    static const LibDeflate &get()
    {
        static const LibDeflate L = [] {
            LibDeflate l;
            const char *no = std::getenv("CCS_NO_LIBDEFLATE");
            if (no && *no && *no != '0') return l;
            void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
            if (!h) return l;
            l.alloc_decompressor = (void *(*)(void))dlsym(h, "libdeflate_alloc_decompressor");
            l.deflate_decompress = (int (*)(void *, const void *, size_t, void *, size_t, size_t *))dlsym(h, "libdeflate_deflate_decompress");
            l.free_decompressor = (void (*)(void *))dlsym(h, "libdeflate_free_decompressor");
            l.alloc_compressor = (void *(*)(int))dlsym(h, "libdeflate_alloc_compressor");
            l.deflate_compress = (size_t (*)(void *, const void *, size_t, void *, size_t))dlsym(h, "libdeflate_deflate_compress");
            l.free_compressor = (void (*)(void *))dlsym(h, "libdeflate_free_compressor");
            l.crc32_ = (uint32_t (*)(uint32_t, const void *, size_t))dlsym(h, "libdeflate_crc32");
            l.ok = l.alloc_decompressor && l.deflate_decompress && l.free_decompressor && l.alloc_compressor && l.deflate_compress && l.free_compressor;
            return l;
        }();
        return L;
    }
};

inline std::vector<uint8_t> deflate_block(const uint8_t *data, size_t n, int level)
{
    std::vector<uint8_t> out(18 + compressBound((uLong)n) + 8);
    static const uint8_t hdr[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
    std::memcpy(out.data(), hdr, 16);
    size_t clen = 0;
    const LibDeflate &ld = LibDeflate::get();
    if (ld.ok) {
        thread_local struct Comp { void *c = nullptr; int level = -1; ~Comp() { if (c) LibDeflate::get().free_compressor(c); } } tc;
        if (!tc.c || tc.level != level) { if (tc.c) ld.free_compressor(tc.c); tc.c = ld.alloc_compressor(level); tc.level = level; }
        if (tc.c) clen = ld.deflate_compress(tc.c, data, n, out.data() + 18, out.size() - 18 - 8);
    }
    if (clen == 0) {                                            // zlib (no libdeflate, or it declined)
    z_stream zs; std::memset(&zs, 0, sizeof(zs));
    if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw std::runtime_error("deflateInit2 failed");
    zs.next_in = const_cast<Bytef *>(data); zs.avail_in = (uInt)n;
    zs.next_out = out.data() + 18; zs.avail_out = (uInt)(out.size() - 18 - 8);
    const int rc = deflate(&zs, Z_FINISH);
    clen = zs.total_out;
    deflateEnd(&zs);
    if (rc != Z_STREAM_END) throw std::runtime_error("deflate failed");
    }
    const size_t total = 18 + clen + 8;
    if (total > 65536) throw std::runtime_error("BGZF block too large");
    out[16] = (uint8_t)((total - 1) & 0xff); out[17] = (uint8_t)((total - 1) >> 8);
    const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), data, (uInt)n);
    uint8_t *t = out.data() + 18 + clen;
    for (int i = 0; i < 4; ++i) t[i] = (uint8_t)(crc >> (8 * i));
    for (int i = 0; i < 4; ++i) t[4 + i] = (uint8_t)((uint32_t)n >> (8 * i));
    out.resize(total);
    return out;
}

// Sequential BGZF input.  The file is memory-mapped; the reader thread only hops from block header to block header and hands
// runs of ~1 MB of compressed blocks ("slabs") to the pool, each of which inflates into ONE contiguous buffer.  Consumers
// either copy bytes out (read) or walk the inflated slab in place (take_slab / the record framer below), so the byte stream is
// never copied by the reader thread.
class BgzfReader {
public:
    typedef std::shared_ptr<std::vector<uint8_t>> Slab;
    BgzfReader(const std::string &path, ThreadPool &pool, int depth = 0) : pool_(pool), depth_(depth > 0 ? depth : 2 * pool.size() + 4)
    {
        fd_ = ::open(path.c_str(), O_RDONLY);
        if (fd_ < 0) throw std::runtime_error("cannot open " + path);
        struct stat st;
        if (fstat(fd_, &st) != 0) throw std::runtime_error("cannot stat " + path);
        size_ = (size_t)st.st_size;
        if (size_ > 0) {
            void *m = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
            if (m == MAP_FAILED) throw std::runtime_error("cannot mmap " + path + " (regular files only)");
            map_ = (const uint8_t *)m;
            (void)madvise(m, size_, MADV_SEQUENTIAL);
        }
    }
    ~BgzfReader()
    {
        for (auto &f : pending_) { try { f.get(); } catch (...) {} }       // tasks reference the mapping
        if (map_) munmap((void *)map_, size_);
        if (fd_ >= 0) ::close(fd_);
    }
    // read exactly n bytes; returns false on clean EOF at a record boundary (n bytes not available)
    bool read(void *dst, size_t n)
    {
        uint8_t *d = (uint8_t *)dst;
        while (n > 0) {
            if (!cur_ || pos_ == cur_->size()) { if (!next_slab()) return false; continue; }
            const size_t k = std::min(n, cur_->size() - pos_);
            std::memcpy(d, cur_->data() + pos_, k);
            d += k; pos_ += k; n -= k;
        }
        return true;
    }
    // the inflated slab the stream is currently in (fetches the next one when the current is used up); null at EOF
    Slab current() { while (!cur_ || pos_ == cur_->size()) if (!next_slab()) return Slab(); return cur_; }
    size_t pos() const { return pos_; }
    void advance(size_t n) { pos_ += n; }
    // continue at a BGZF virtual offset (compressed block start << 16 | offset in the inflated block), e.g. from a .pbi
    void seek(uint64_t voff)
    {
        for (auto &f : pending_) { try { f.get(); } catch (...) {} }
        pending_.clear(); cur_.reset(); pos_ = 0;
        foff_ = (size_t)(voff >> 16);
        if (foff_ > size_) throw std::runtime_error("virtual offset beyond the end of the file");
        const size_t skip = (size_t)(voff & 0xffff);
        if (skip) {
            if (!next_slab() || skip > cur_->size()) throw std::runtime_error("virtual offset does not point into a BGZF block");
            pos_ = skip;
        }
    }

private:
    struct Blk { size_t off, size; };
    void schedule()
    {
        while (foff_ < size_ && (int)pending_.size() < depth_) {
            std::vector<Blk> blks;
            size_t comp = 0, raw = 0;
            while (foff_ < size_ && comp < (1u << 20)) {
                if (size_ - foff_ < 18) throw std::runtime_error("truncated BGZF header");
                const uint8_t *h = map_ + foff_;
                if (h[0] != 0x1f || h[1] != 0x8b || !(h[3] & 4)) throw std::runtime_error("not a BGZF file");
                const size_t xlen = h[10] | (h[11] << 8);
                if (size_ - foff_ < 12 + xlen) throw std::runtime_error("truncated BGZF header");
                size_t bsize = 0;
                for (size_t p = 12; p + 4 <= 12 + xlen;) {                  // find the BC subfield
                    const size_t sl = h[p + 2] | (h[p + 3] << 8);
                    if (h[p] == 'B' && h[p + 1] == 'C' && p + 6 <= 12 + xlen) bsize = (size_t)(h[p + 4] | (h[p + 5] << 8)) + 1;
                    p += 4 + sl;
                }
                if (!bsize) throw std::runtime_error("BGZF block without BC field");
                if (size_ - foff_ < bsize || bsize < 12 + xlen + 8) throw std::runtime_error("truncated BGZF block");
                const uint8_t *t = h + bsize - 4;
                raw += (size_t)t[0] | ((size_t)t[1] << 8) | ((size_t)t[2] << 16) | ((size_t)t[3] << 24);
                blks.push_back({foff_, bsize});
                foff_ += bsize; comp += bsize;
            }
            const uint8_t *map = map_;
            pending_.push_back(pool_.submit([map, blks, raw]() -> Slab {
                Slab out = std::make_shared<std::vector<uint8_t>>(raw);
                size_t at = 0;
                const LibDeflate &ld = LibDeflate::get();
                if (ld.ok) {                                    // whole-buffer inflate of every block of the slab
                    thread_local struct Dec { void *d = nullptr; ~Dec() { if (d) LibDeflate::get().free_decompressor(d); } } td;
                    if (!td.d) td.d = ld.alloc_decompressor();
                    if (!td.d) throw std::runtime_error("libdeflate_alloc_decompressor failed");
                    for (const Blk &b : blks) {
                        const uint8_t *h = map + b.off;
                        const size_t xlen = h[10] | (h[11] << 8), off = 12 + xlen;
                        const uint8_t *t = h + b.size - 4;
                        const size_t isize = (size_t)t[0] | ((size_t)t[1] << 8) | ((size_t)t[2] << 16) | ((size_t)t[3] << 24);
                        if (isize == 0) continue;
                        if (at + isize > out->size()) throw std::runtime_error("BGZF block sizes are inconsistent");
                        size_t got = 0;
                        if (ld.deflate_decompress(td.d, h + off, b.size - off - 8, out->data() + at, isize, &got) != 0 || got != isize)
                            throw std::runtime_error("BGZF block does not inflate");
                        const uint8_t *c = h + b.size - 8;         // the block's CRC32 is verified: a flipped bit is an error, not data
                        if (ld.crc(out->data() + at, isize) != ((uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16) | ((uint32_t)c[3] << 24)))
                            throw std::runtime_error("BGZF block fails its CRC32");
                        at += isize;
                    }
                    return out;
                }
                z_stream zs; std::memset(&zs, 0, sizeof(zs));
                if (inflateInit2(&zs, -15) != Z_OK) throw std::runtime_error("inflateInit2 failed");
                for (const Blk &b : blks) {
                    const uint8_t *h = map + b.off;
                    const size_t xlen = h[10] | (h[11] << 8), off = 12 + xlen;
                    const uint8_t *t = h + b.size - 4;
                    const size_t isize = (size_t)t[0] | ((size_t)t[1] << 8) | ((size_t)t[2] << 16) | ((size_t)t[3] << 24);
                    if (isize == 0) continue;
                    if (at + isize > out->size()) { inflateEnd(&zs); throw std::runtime_error("BGZF block sizes are inconsistent"); }
                    if (inflateReset(&zs) != Z_OK) { inflateEnd(&zs); throw std::runtime_error("inflateReset failed"); }
                    zs.next_in = const_cast<Bytef *>(h + off); zs.avail_in = (uInt)(b.size - off - 8);
                    zs.next_out = out->data() + at; zs.avail_out = (uInt)isize;
                    const int rc = inflate(&zs, Z_FINISH);
                    if (rc != Z_STREAM_END || zs.avail_out != 0) { inflateEnd(&zs); throw std::runtime_error("BGZF block does not inflate"); }
                    const uint8_t *c = h + b.size - 8;
                    if (ld.crc(out->data() + at, isize) != ((uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16) | ((uint32_t)c[3] << 24)))
                        { inflateEnd(&zs); throw std::runtime_error("BGZF block fails its CRC32"); }
                    at += isize;
                }
                inflateEnd(&zs);
                return out;
            }));
        }
    }
    bool next_slab()
    {
        schedule();
        if (pending_.empty()) return false;
        cur_ = pending_.front().get(); pending_.pop_front(); pos_ = 0;
        schedule();
        return true;
    }
    ThreadPool &pool_;
    int depth_;
    int fd_ = -1;
    const uint8_t *map_ = nullptr;
    size_t size_ = 0, foff_ = 0;
    std::deque<std::future<Slab>> pending_;
    Slab cur_;
    size_t pos_ = 0;
};

class BgzfWriter {   // pooled deflate, in-order write
public:
    BgzfWriter(const std::string &path, ThreadPool &pool, int level = 4) : pool_(pool), level_(level)
    {
        f_ = std::fopen(path.c_str(), "wb");
        if (!f_) throw std::runtime_error("cannot create " + path);
    }
    ~BgzfWriter() { try { close(); } catch (...) {} }
    // position of the next byte as (block number << 16 | offset in that block); virtual_offset() resolves it after close()
    uint64_t mark() const { return ((uint64_t)nblocks_ << 16) | (uint64_t)buf_.size(); }
    uint64_t virtual_offset(uint64_t mark) const
    {
        const size_t blk = (size_t)(mark >> 16);
        if (blk >= block_off_.size()) throw std::runtime_error("BGZF mark beyond the written blocks");
        return (block_off_[blk] << 16) | (mark & 0xffff);
    }
    void write(const void *src, size_t n)
    {
        const uint8_t *s = (const uint8_t *)src;
        while (n > 0) {
            const size_t k = std::min(n, kBlock - buf_.size());
            buf_.insert(buf_.end(), s, s + k);
            s += k; n -= k;
            if (buf_.size() == kBlock) flush_block();
        }
    }
    void close()
    {
        if (!f_) return;
        if (!buf_.empty()) flush_block();
        drain(0);
        block_off_.push_back(written_);      // a mark taken at the very end resolves to the EOF block
        static const uint8_t eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const bool ok = std::fwrite(eof, 1, 28, f_) == 28;
        const bool closed = std::fclose(f_) == 0;
        f_ = nullptr;
        if (!ok || !closed) throw std::runtime_error("short write (final BGZF block / close)");
    }

private:
    static constexpr size_t kBlock = 0xff00;
    void flush_block()
    {
        auto sp = std::make_shared<std::vector<uint8_t>>(std::move(buf_));
        buf_.clear(); buf_.reserve(kBlock);
        ++nblocks_;
        const int lvl = level_;
        pending_.push_back(pool_.submit([sp, lvl] { return deflate_block(sp->data(), sp->size(), lvl); }));
        drain(128);
    }
    void drain(size_t keep)
    {
        while (pending_.size() > keep) {
            const std::vector<uint8_t> c = pending_.front().get(); pending_.pop_front();
            block_off_.push_back(written_);
            written_ += c.size();
            if (std::fwrite(c.data(), 1, c.size(), f_) != c.size()) throw std::runtime_error("short write");
        }
    }
    ThreadPool &pool_;
    int level_;
    FILE *f_ = nullptr;
    std::vector<uint8_t> buf_;
    std::deque<std::future<std::vector<uint8_t>>> pending_;
    size_t nblocks_ = 0;                     // blocks handed to the pool so far
    uint64_t written_ = 0;                   // compressed bytes written
    std::vector<uint64_t> block_off_;        // file offset of every written block
};

// ---------------------------------------------------------------------------------------------- PacBio BAM index (.pbi)
// "the .pbi file enables random access by ZMW" (docs/faq/parallelize.md:9-13).  The format specification is not part of the
// reference mount; this is the layout of pbbam 3.x as recalled ([RECALL], unverified against pbindex): a BGZF stream with
// magic "PBI\1", u32 version, u16 section flags (0 = basic only), u32 n_reads, 18 reserved bytes, then the basic section
// column by column: rgId i32[n], qStart i32[n], qEnd i32[n], holeNumber i32[n], readQual f32[n], ctxtFlag u8[n],
// fileOffset i64[n] (BGZF virtual offsets of the records).  Extra sections (mapped / reference / barcode) are ignored.
struct PbiIndex {
    std::vector<int32_t> rg_id, q_start, q_end, hole;
    std::vector<float> read_qual;
    std::vector<uint8_t> ctxt;
    std::vector<int64_t> file_offset;
    size_t size() const { return hole.size(); }
};

inline bool read_pbi(const std::string &path, ThreadPool &pool, PbiIndex &x)
{
    struct stat st;
    if (::stat(path.c_str(), &st) != 0) return false;
    BgzfReader in(path, pool);
    uint8_t h[32];
    if (!in.read(h, 32) || std::memcmp(h, "PBI\1", 4) != 0) throw std::runtime_error(path + ": not a PacBio BAM index");
    const uint32_t n = h[10] | (h[11] << 8) | (h[12] << 16) | ((uint32_t)h[13] << 24);
    if ((uint64_t)n * 29 > (uint64_t)st.st_size * 1100 + 64) throw std::runtime_error(path + ": record count does not fit the file size");   // (deflate expands < 1100x)
    auto col = [&](void *dst, size_t bytes) { if (bytes && !in.read(dst, bytes)) throw std::runtime_error(path + ": truncated PacBio BAM index"); };
    x.rg_id.resize(n); x.q_start.resize(n); x.q_end.resize(n); x.hole.resize(n); x.read_qual.resize(n); x.ctxt.resize(n); x.file_offset.resize(n);
    col(x.rg_id.data(), 4 * (size_t)n); col(x.q_start.data(), 4 * (size_t)n); col(x.q_end.data(), 4 * (size_t)n); col(x.hole.data(), 4 * (size_t)n);
    col(x.read_qual.data(), 4 * (size_t)n); col(x.ctxt.data(), (size_t)n); col(x.file_offset.data(), 8 * (size_t)n);
    return true;
}

inline void write_pbi(const std::string &path, ThreadPool &pool, const PbiIndex &x)
{
    BgzfWriter out(path, pool);
    const uint32_t n = (uint32_t)x.size();
    uint8_t h[32]; std::memset(h, 0, sizeof(h));
    std::memcpy(h, "PBI\1", 4);
    h[4] = 1; h[5] = 0; h[6] = 3; h[7] = 0;                  // version 3.0.1 as 0x00030001
    h[10] = (uint8_t)n; h[11] = (uint8_t)(n >> 8); h[12] = (uint8_t)(n >> 16); h[13] = (uint8_t)(n >> 24);
    out.write(h, 32);
    out.write(x.rg_id.data(), 4 * (size_t)n); out.write(x.q_start.data(), 4 * (size_t)n); out.write(x.q_end.data(), 4 * (size_t)n);
    out.write(x.hole.data(), 4 * (size_t)n); out.write(x.read_qual.data(), 4 * (size_t)n); out.write(x.ctxt.data(), (size_t)n);
    out.write(x.file_offset.data(), 8 * (size_t)n);
    out.close();
}

// ---------------------------------------------------------------------------------------------- BAM records
struct Subread {
    // Round 3: a subread is a VIEW of its record in the inflated BGZF bytes (kept alive by `keep`) — the filters need only lengths
    // and tags; bases / pw / ip are decoded exactly once, straight into the page-locked staging of the batch (decode_* below).
    // Decoding every record into three vectors first cost a third of the host time and a second copy at packing.
    std::string name;
    int32_t zm = -1;
    int32_t cx = -1;            // -1: tag absent
    float snr[4] = {0, 0, 0, 0};
    bool has_snr = false, has_n = false;
    uint8_t strand = 0;         // 0 forward / 1 reverse pass (set by the driver: cx direction bits, else alternation)
    uint8_t partial = 0;        // ccsx_batch.flags bits 1-2: 2 = partial pass with the adapter at its start, 6 = ... at its end
    uint32_t len = 0;           // bases
    const uint8_t *seq = nullptr;                              // BAM nibbles, two per byte
    const uint8_t *pw_p = nullptr, *ip_p = nullptr;            // payload of the B arrays as stored
    char pw_t = 0, ip_t = 0;                                   // their element types ('C': CodecV1 bytes pass through)
    uint32_t pw_n = 0, ip_n = 0;
    std::shared_ptr<const void> keep;
    size_t size() const { return len; }
    inline void decode_bases(uint8_t *dst) const;              // 0..3 (a non-ACGT nibble decodes as 0; has_n says so)
    inline void decode_pw(uint8_t *dst) const;                 // CodecV1 codes; a missing / mismatched array = all 2
    inline void decode_ip(uint8_t *dst) const;                 // ... = all 1
};

struct BamHeader {
    std::string text;
    std::vector<std::pair<std::string, uint32_t>> refs;
};

inline uint32_t rd32(const uint8_t *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }

inline bool read_header(BgzfReader &in, BamHeader &h)
{
    uint8_t magic[4];
    if (!in.read(magic, 4) || std::memcmp(magic, "BAM\1", 4)) throw std::runtime_error("not a BAM file");
    auto must = [&](void *dst, size_t n) { if (n && !in.read(dst, n)) throw std::runtime_error("truncated BAM header"); };
    uint8_t b4[4];
    must(b4, 4);
    const uint32_t l_text = rd32(b4);
    if (l_text > (1u << 30)) throw std::runtime_error("malformed BAM header (text length)");
    h.text.resize(l_text);
    if (!h.text.empty()) must(&h.text[0], h.text.size());
    must(b4, 4);
    const uint32_t nref = rd32(b4);
    if (nref > (1u << 24)) throw std::runtime_error("malformed BAM header (reference count)");
    for (uint32_t i = 0; i < nref; ++i) {
        must(b4, 4);
        const uint32_t l_name = rd32(b4);
        if (l_name > (1u << 16)) throw std::runtime_error("malformed BAM header (reference name)");
        std::string nm(l_name, '\0');
        if (l_name) must(&nm[0], nm.size());
        must(b4, 4);
        h.refs.emplace_back(nm.c_str(), rd32(b4));
    }
    return true;
}

// pw / ip travel as CodecV1 codes (include/ccsx.h ccsx_batch): B,C tags are passed through untouched, raw-frame
// B,S tags are encoded (nearest representable value, ties up, clamp at 952 frames)
inline uint8_t codec_v1_encode(int64_t f)
{
    if (f < 0) f = 0;
    if (f < 64) return (uint8_t)f;
    if (f < 192) return (uint8_t)(64 + (f - 64 + 1) / 2);
    if (f < 448) return (uint8_t)(128 + (f - 192 + 2) / 4);
    const int64_t c = 192 + (f - 448 + 4) / 8;
    return (uint8_t)(c > 255 ? 255 : c);
}

inline size_t tag_value_size(char t)
{
    switch (t) { case 'c': case 'C': case 'A': return 1; case 's': case 'S': return 2; case 'i': case 'I': case 'f': return 4; default: return 0; }
}

// one table look-up per packed byte: two base codes + "contains a non-ACGT nibble"
struct NibLut {
    uint16_t v[256]; uint8_t bad[256];
    NibLut()
    {
        static const int8_t nib2code[16] = {-1, 0, 1, -1, 2, -1, -1, -1, 3, -1, -1, -1, -1, -1, -1, -1};
        for (int b = 0; b < 256; ++b) {
            const int hi = nib2code[b >> 4], lo = nib2code[b & 15];
            v[b] = (uint16_t)((hi < 0 ? 0 : hi) | ((lo < 0 ? 0 : lo) << 8)); bad[b] = (uint8_t)((hi < 0 ? 1 : 0) | (lo < 0 ? 2 : 0));
        }
    }
};
inline const NibLut &nib_lut() { static const NibLut lut; return lut; }

inline void Subread::decode_bases(uint8_t *dst) const
{
    const NibLut &lut = nib_lut();
    const uint32_t pairs = len >> 1;
    for (uint32_t k = 0; k < pairs; ++k) std::memcpy(dst + 2 * k, &lut.v[seq[k]], 2);
    if (len & 1) dst[len - 1] = (uint8_t)(lut.v[seq[pairs]] & 0xff);
}

inline void decode_codes(const uint8_t *q, char st, uint32_t n, uint32_t len, uint8_t deflt, uint8_t *dst)
{
    if (!q || n != len) { std::memset(dst, deflt, len); return; }      // (an array of the wrong length is ignored, as before)
    if (st == 'c' || st == 'C') { std::memcpy(dst, q, n); return; }     // CodecV1 bytes pass through
    const size_t es = tag_value_size(st);
    for (uint32_t i = 0; i < n; ++i) {
        const uint8_t *e = q + (size_t)i * es;
        int64_t f;
        switch (st) {
            case 's': f = (int16_t)(e[0] | (e[1] << 8)); break; case 'S': f = (uint16_t)(e[0] | (e[1] << 8)); break;
            case 'i': f = (int32_t)rd32(e); break; case 'I': f = rd32(e); break;
            default: f = 0;
        }
        dst[i] = codec_v1_encode(f);
    }
}
inline void Subread::decode_pw(uint8_t *dst) const { decode_codes(pw_p, pw_t, pw_n, len, 2, dst); }
inline void Subread::decode_ip(uint8_t *dst) const { decode_codes(ip_p, ip_t, ip_n, len, 1, dst); }

// decode one record body (the bytes after block_size)
// Every length field of the record is checked against the record's own size before it is used (a lying l_seq, tag count or
// an unterminated string must end in "malformed BAM record", not in an out-of-bounds read on a pool thread).
inline void parse_subread(const uint8_t *p, uint32_t bs, Subread &r)
{
    auto bad = [] { throw std::runtime_error("malformed BAM record"); };
    if (bs < 32) bad();
    const uint32_t l_name = p[8], n_cig = p[12] | (p[13] << 8), l_seq = rd32(p + 16);
    if ((uint64_t)32 + l_name + (uint64_t)4 * n_cig + ((uint64_t)l_seq + 1) / 2 + l_seq > bs) bad();
    r = Subread();
    r.name.assign((const char *)p + 32, l_name ? l_name - 1 : 0);
    const uint8_t *seq = p + 32 + l_name + 4 * n_cig;
    r.len = l_seq; r.seq = seq;
    {   // only "does it hold a non-ACGT nibble" is decided here (one table look-up per packed byte, no writes)
        const NibLut &lut = nib_lut();
        const uint32_t pairs = l_seq >> 1;
        unsigned anybad = 0;
        for (uint32_t k = 0; k < pairs; ++k) anybad |= lut.bad[seq[k]];
        if (l_seq & 1) anybad |= (lut.bad[seq[pairs]] & 1);
        if (anybad) r.has_n = true;
    }
    const uint8_t *t = seq + (l_seq + 1) / 2 + l_seq, *end = p + bs;
    while (t + 3 <= end) {
        const char t0 = (char)t[0], t1 = (char)t[1], ty = (char)t[2];
        t += 3;
        auto rdint = [&](char tt, const uint8_t *q) -> int64_t {
            switch (tt) {
                case 'c': return (int8_t)q[0]; case 'C': return q[0];
                case 's': return (int16_t)(q[0] | (q[1] << 8)); case 'S': return (uint16_t)(q[0] | (q[1] << 8));
                case 'i': return (int32_t)rd32(q); case 'I': return rd32(q);
                default: return 0;
            }
        };
        if (ty == 'Z' || ty == 'H') { while (t < end && *t) ++t; if (t >= end) bad(); ++t; continue; }
        if (ty == 'B') {
            if (t + 5 > end) bad();
            const char st = (char)t[0];
            const uint32_t n = rd32(t + 1);
            const uint8_t *q = t + 5;
            const size_t es = tag_value_size(st);
            if (!es || (uint64_t)n * es > (uint64_t)(end - q)) bad();
            if (t0 == 's' && t1 == 'n' && st == 'f' && n == 4) { std::memcpy(r.snr, q, 16); r.has_snr = true; }
            else if (t0 == 'p' && t1 == 'w') { r.pw_p = q; r.pw_t = st; r.pw_n = n; }
            else if (t0 == 'i' && t1 == 'p') { r.ip_p = q; r.ip_t = st; r.ip_n = n; }
            t = q + (size_t)n * es;
            continue;
        }
        const size_t vs = tag_value_size(ty);
        if (!vs) throw std::runtime_error("unknown BAM tag type");
        if (t + vs > end) bad();
        if (t0 == 'z' && t1 == 'm') r.zm = (int32_t)rdint(ty, t);
        else if (t0 == 'c' && t1 == 'x') r.cx = (int32_t)rdint(ty, t);
        t += vs;
    }
}

// a run of raw records: framed by the reader thread (pointer hops inside one inflated slab, no copy), decoded on the pool.
// Only a record that straddles two slabs is copied (into `carry`).
struct RawChunk {
    BgzfReader::Slab slab;                                    // keeps the referenced bytes alive
    std::deque<std::vector<uint8_t>> carry;
    std::vector<std::pair<const uint8_t *, uint32_t>> recs;   // (body, size) of every record
};

inline bool read_raw_chunk(BgzfReader &in, RawChunk &c)
{
    c.slab.reset(); c.carry.clear(); c.recs.clear();
    c.slab = in.current();
    if (!c.slab) return false;
    const BgzfReader::Slab slab = c.slab;
    for (;;) {
        const size_t pos = in.pos(), left = slab->size() - pos;
        if (left >= 4) {
            const uint32_t bs = rd32(slab->data() + pos);
            if (left >= 4 + (size_t)bs) {                      // whole record inside this slab
                c.recs.emplace_back(slab->data() + pos + 4, bs);
                in.advance(4 + (size_t)bs);
                if (in.pos() == slab->size()) break;
                continue;
            }
        }
        if (left == 0) break;
        // the record continues in the next slab(s): copy it out, then end the chunk (the stream has moved on)
        uint8_t b4[4];
        if (!in.read(b4, 4)) throw std::runtime_error("truncated BAM record");
        const uint32_t bs = rd32(b4);
        if (bs > (1u << 28)) throw std::runtime_error("malformed BAM record (block size)");   // no subread record is 256 MB
        c.carry.emplace_back(bs);
        if (bs && !in.read(c.carry.back().data(), bs)) throw std::runtime_error("truncated BAM record");
        c.recs.emplace_back(c.carry.back().data(), bs);
        break;
    }
    return !c.recs.empty();
}

inline std::vector<Subread> decode_chunk(const std::shared_ptr<const RawChunk> &c)
{
    std::vector<Subread> out(c->recs.size());
    for (size_t i = 0; i < c->recs.size(); ++i) { parse_subread(c->recs[i].first, c->recs[i].second, out[i]); out[i].keep = c; }   // the views live as long as the chunk
    return out;
}

// ---- record builders -------------------------------------------------------------------------------
struct RecordBuilder {
    std::vector<uint8_t> b;
    void u32(uint32_t v) { for (int i = 0; i < 4; ++i) b.push_back((uint8_t)(v >> (8 * i))); }
    void begin(const std::string &name, const uint8_t *bases, const uint8_t *qual, uint32_t n)
    {
        b.clear();
        u32(0);                               // block size, patched in finish()
        u32((uint32_t)-1); u32((uint32_t)-1); // refID, pos
        b.push_back((uint8_t)(name.size() + 1)); b.push_back(255);   // l_read_name, mapq
        b.push_back(4680 & 0xff); b.push_back(4680 >> 8);            // bin
        b.push_back(0); b.push_back(0);                              // n_cigar
        b.push_back(4); b.push_back(0);                              // flag = unmapped
        u32(n); u32((uint32_t)-1); u32((uint32_t)-1); u32(0);
        b.insert(b.end(), name.begin(), name.end()); b.push_back(0);
        static const uint8_t code2nib[4] = {1, 2, 4, 8};
        const size_t at = b.size();
        b.resize(at + (n + 1) / 2);                              // (one resize, plain stores: the writer thread packs ~10 kb per record)
        uint8_t *dst = b.data() + at;
        for (uint32_t i = 0; i + 1 < n; i += 2) dst[i >> 1] = (uint8_t)((code2nib[bases[i] & 3] << 4) | code2nib[bases[i + 1] & 3]);
        if (n & 1) dst[n >> 1] = (uint8_t)(code2nib[bases[n - 1] & 3] << 4);
        if (qual) b.insert(b.end(), qual, qual + n); else b.insert(b.end(), n, 0xff);
    }
    void tagZ(const char *t, const std::string &v) { b.push_back(t[0]); b.push_back(t[1]); b.push_back('Z'); b.insert(b.end(), v.begin(), v.end()); b.push_back(0); }
    void tagi(const char *t, int32_t v) { b.push_back(t[0]); b.push_back(t[1]); b.push_back('i'); u32((uint32_t)v); }
    void tagC(const char *t, uint8_t v) { b.push_back(t[0]); b.push_back(t[1]); b.push_back('C'); b.push_back(v); }
    void tagf(const char *t, float v) { uint32_t u; std::memcpy(&u, &v, 4); b.push_back(t[0]); b.push_back(t[1]); b.push_back('f'); u32(u); }
    void tagBf(const char *t, const float *v, uint32_t n)
    {
        b.push_back(t[0]); b.push_back(t[1]); b.push_back('B'); b.push_back('f'); u32(n);
        for (uint32_t i = 0; i < n; ++i) { uint32_t u; std::memcpy(&u, v + i, 4); u32(u); }
    }
    void tagBC(const char *t, const uint8_t *v, uint32_t n)
    {
        b.push_back(t[0]); b.push_back(t[1]); b.push_back('B'); b.push_back('C'); u32(n);
        b.insert(b.end(), v, v + n);
    }
    void finish(BgzfWriter &out)
    {
        const uint32_t bs = (uint32_t)b.size() - 4;
        for (int i = 0; i < 4; ++i) b[i] = (uint8_t)(bs >> (8 * i));
        out.write(b.data(), b.size());
    }
};

inline void write_header(BgzfWriter &out, const std::string &text)
{
    out.write("BAM\1", 4);
    uint8_t b4[4];
    const uint32_t n = (uint32_t)text.size();
    for (int i = 0; i < 4; ++i) b4[i] = (uint8_t)(n >> (8 * i));
    out.write(b4, 4); out.write(text.data(), n);
    std::memset(b4, 0, 4); out.write(b4, 4);   // n_ref = 0
}

}  // namespace bamio
