// trk_vcf.cpp -- native VCF / BGZF reader (host C++17, zlib): decodes records straight into
// the packed batch layout of include/trk.h.  See include/trk_vcf.h for the contract and the
// reference call sites it stands in for (cyvcf2 behind trtools/utils/utils.py:19-67).
#include <unistd.h>
#include <zlib.h>
#include <dlfcn.h>
#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <charconv>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/trk_vcf.h"

// ---- options (include/trk_test.h): a process-wide table of name -> value, set by trk_test_set_option ----
// Values are never freed (a reader of an option may still hold the pointer): tests set a handful.
namespace {
struct OptTable {
    std::mutex m;
    std::vector<std::pair<std::string, const char*>> kv;
    std::atomic<int> n_set{0};
};
OptTable& opt_table() {
    static OptTable* t = new OptTable;
    return *t;
}
}  // namespace
extern "C" __attribute__((visibility("hidden"))) const char* trk_opt(const char* name) {
    OptTable& t = opt_table();
    if (t.n_set.load(std::memory_order_acquire) > 0) {
        std::lock_guard<std::mutex> g(t.m);
        for (auto& e : t.kv)
            if (e.first == name) return e.second;
    }
#ifdef TRK_LAB
    return getenv(name);   // the lab build (tools/): options from the environment too
#else
    return nullptr;
#endif
}
extern "C" int trk_test_set_option(const char* name, const char* value) {
    if (!name || !*name) return 2;
    OptTable& t = opt_table();
    std::lock_guard<std::mutex> g(t.m);
    const char* v = value ? strdup(value) : nullptr;
    for (auto& e : t.kv)
        if (e.first == name) {
            e.second = v;
            return 0;
        }
    t.kv.emplace_back(name, v);
    t.n_set.fetch_add(1, std::memory_order_release);
    return 0;
}
extern "C" const char* trk_test_get_option(const char* name) { return name ? trk_opt(name) : nullptr; }

namespace {

constexpr int32_t INT_MISSING = INT32_MIN;
constexpr int32_t INT_VECTOR_END = INT32_MIN + 1;
std::string g_open_error;

// ---------------------------------------------------------------------------------------
// The decompressed text: a growable byte buffer WITHOUT value initialisation (std::string::resize zero-fills,
// i.e. touches every new page on one thread before the inflate workers get to write it) that grows by realloc
// (glibc remaps large blocks instead of copying them) and asks for huge pages.  At 120 MB of text per batch the
// zero-fill, the page faults and the doubling copies of a std::string were most of the reader's time.
// ---------------------------------------------------------------------------------------
class TextBuf {
    char* p_ = nullptr;
    size_t n_ = 0, cap_ = 0;
    bool ext_ = false;      // the memory is the caller's (adopt(): pinned pages for the upload of the text): never freed,
                            // never reallocated, never cached here; growing beyond it moves the bytes to memory of our own
    void drop() {           // let go of the current memory
        if (ext_ || !p_) {
            p_ = nullptr;
        } else if (cap_ >= kCacheMin) {
            Cache& c = cache();
            std::lock_guard<std::mutex> g(c.m);
            if (c.bytes + cap_ <= c.limit) {
                c.free_list.emplace_back(p_, cap_);
                c.bytes += cap_;
            } else {
                free(p_);
            }
            p_ = nullptr;
        } else {
            free(p_);
            p_ = nullptr;
        }
        cap_ = 0;
        ext_ = false;
    }

    // Big buffers are kept for the next reader of the process instead of going back to the C library: giving ~500 MB
    // of huge-page text back is 50 ms of trk_vcf_close after a 1 GB file (and the next reader faults the same pages
    // in again).  At most TRK_VCF_BUF_CACHE_MB (default 4096; 0: off) are held; a one-shot command line never frees
    // them before it exits.
    struct Cache {
        std::mutex m;
        std::vector<std::pair<char*, size_t>> free_list;
        size_t bytes = 0, limit = (size_t)4096 << 20;
        Cache() {
            if (const char* e = getenv("TRK_VCF_BUF_CACHE_MB")) limit = (size_t)std::max(0L, atol(e)) << 20;
        }
    };
    static Cache& cache() {
        static Cache* c = new Cache;     // never destroyed: buffers may be handed back during static destruction
        return *c;
    }
    static constexpr size_t kCacheMin = (size_t)16 << 20;

public:
    static constexpr size_t npos = std::string::npos;
    TextBuf() = default;
    TextBuf(const TextBuf&) = delete;
    TextBuf& operator=(const TextBuf&) = delete;
    ~TextBuf() { drop(); }
    // continue in the caller's memory [q, q + cap): the bytes held now are copied over (false, nothing changed: they do not fit)
    bool adopt(char* q, size_t cap) {
        if (!q || n_ > cap) return false;
        if (n_) memcpy(q, p_, n_);
        const size_t n = n_;
        drop();
        p_ = q;
        cap_ = cap;
        n_ = n;
        ext_ = true;
        return true;
    }
    bool external() const { return ext_; }
    size_t size() const { return n_; }
    size_t capacity() const { return cap_; }
    const char* data() const { return p_; }
    char* data() { return p_; }
    char& operator[](size_t i) { return p_[i]; }
    const char& operator[](size_t i) const { return p_[i]; }
    void reserve(size_t c) {
        if (c <= cap_) return;
        if (ext_) {                       // the caller's memory is too small after all: move to memory of our own
            char* old = p_;
            const size_t n = n_;
            p_ = nullptr;
            cap_ = 0;
            ext_ = false;
            n_ = 0;
            reserve(c);
            if (n) memcpy(p_, old, n);
            n_ = n;
            return;
        }
        if (!p_ && c >= kCacheMin) {      // a buffer a closed reader left behind, the largest one
            Cache& ch = cache();
            std::lock_guard<std::mutex> g(ch.m);
            size_t best = SIZE_MAX;
            for (size_t i = 0; i < ch.free_list.size(); ++i)
                if (best == SIZE_MAX || ch.free_list[i].second > ch.free_list[best].second) best = i;
            if (best != SIZE_MAX) {
                p_ = ch.free_list[best].first;
                cap_ = ch.free_list[best].second;
                ch.bytes -= cap_;
                ch.free_list.erase(ch.free_list.begin() + (long)best);
                if (c <= cap_) return;
            }
        }
        size_t nc = std::max<size_t>(c, cap_ + cap_ / 2);
        nc = (nc + ((size_t)1 << 21) - 1) & ~(((size_t)1 << 21) - 1);
        if (p_ && n_ == 0) {              // nothing to keep: no copy of the old bytes
            free(p_);
            p_ = nullptr;
        }
        char* q = static_cast<char*>(realloc(p_, nc));
        if (!q) throw std::bad_alloc();
        p_ = q;
        cap_ = nc;
#ifdef MADV_HUGEPAGE
        (void)madvise(p_, cap_, MADV_HUGEPAGE);
#endif
    }
    void resize(size_t n) {   // new bytes are NOT initialised
        reserve(n);
        n_ = n;
    }
    void clear() { n_ = 0; }
    void push_back(char c) {
        reserve(n_ + 1);
        p_[n_++] = c;
    }
    void erase(size_t pos, size_t n) {   // only ever used on the consumed front
        if (pos != 0 || n == 0) return;
        n = std::min(n, n_);
        memmove(p_, p_ + n, n_ - n);
        n_ -= n;
    }
    size_t find(char c, size_t from) const {
        if (from >= n_) return npos;
        const void* r = memchr(p_ + from, c, n_ - from);
        return r ? (size_t)(static_cast<const char*>(r) - p_) : npos;
    }
    int compare(size_t pos, size_t n, const char* s) const {
        const size_t m = strlen(s);
        const size_t have = pos < n_ ? std::min(n, n_ - pos) : 0;
        const int c = memcmp(p_ + pos, s, std::min(have, m));
        return c ? c : (have < m ? -1 : (have > m ? 1 : 0));
    }
    std::string substr(size_t pos, size_t n) const { return std::string(p_ + pos, std::min(n, n_ - pos)); }
    void swap(TextBuf& o) {
        std::swap(p_, o.p_);
        std::swap(n_, o.n_);
        std::swap(cap_, o.cap_);
        std::swap(ext_, o.ext_);
    }
};

// libdeflate (2-3x zlib's inflate rate) is in the image as a runtime library without its header: bound by hand at
// run time, zlib when it is absent.
struct Deflater {
    void* handle = nullptr;
    void* (*alloc)() = nullptr;
    int (*decompress)(void*, const void*, size_t, void*, size_t, size_t*) = nullptr;
    void (*release)(void*) = nullptr;
    Deflater() {
        if (trk_opt("TRK_VCF_ZLIB")) return;
        handle = dlopen("libdeflate.so.0", RTLD_NOW);
        if (!handle) return;
        alloc = reinterpret_cast<void* (*)()>(dlsym(handle, "libdeflate_alloc_decompressor"));
        decompress = reinterpret_cast<int (*)(void*, const void*, size_t, void*, size_t, size_t*)>(
            dlsym(handle, "libdeflate_deflate_decompress"));
        release = reinterpret_cast<void (*)(void*)>(dlsym(handle, "libdeflate_free_decompressor"));
        if (!alloc || !decompress || !release) decompress = nullptr;
        // (round 6) the compressor, for BGZF output
        c_alloc = reinterpret_cast<void* (*)(int)>(dlsym(handle, "libdeflate_alloc_compressor"));
        c_compress = reinterpret_cast<size_t (*)(void*, const void*, size_t, void*, size_t)>(dlsym(handle, "libdeflate_deflate_compress"));
        c_release = reinterpret_cast<void (*)(void*)>(dlsym(handle, "libdeflate_free_compressor"));
        c_crc32 = reinterpret_cast<uint32_t (*)(uint32_t, const void*, size_t)>(dlsym(handle, "libdeflate_crc32"));
        if (!c_alloc || !c_compress || !c_release || !c_crc32) c_compress = nullptr;
    }
    void* (*c_alloc)(int) = nullptr;
    size_t (*c_compress)(void*, const void*, size_t, void*, size_t) = nullptr;
    void (*c_release)(void*) = nullptr;
    uint32_t (*c_crc32)(uint32_t, const void*, size_t) = nullptr;
};
const Deflater& deflater() {
    static Deflater d;
    return d;
}

// CPUs' worth of time the container's cgroup grants per scheduling period (cpu.max of cgroup v2, cfs_quota_us / cfs_period_us
// of v1); 0: no quota.  The GPU boxes of this project show 256 hardware threads and grant 16 (cpu.max "1600000 100000"):
// 64 reader threads burn the period's budget in bursts and the whole process is throttled until the next period
// (profiles/r04_notes.md section 13), so the default thread counts follow the quota, not the visible CPUs.
static int cpu_quota() {
    static const int q = [] {
        long quota = -1, period = 0;
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char a[32] = {0};
            if (fscanf(f, "%31s %ld", a, &period) == 2 && strcmp(a, "max") != 0) quota = atol(a);
            fclose(f);
        } else {
            if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
                if (fscanf(g, "%ld", &quota) != 1) quota = -1;
                fclose(g);
            }
            if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (fscanf(g, "%ld", &period) != 1) period = 0;
                fclose(g);
            }
        }
        return (quota > 0 && period > 0) ? (int)std::max(1L, (quota + period - 1) / period) : 0;
    }();
    return q;
}
// default number of worker threads for a phase that would like `want`: the visible CPUs, and twice the quota when
// there is one (bursts above the quota pay while the other phases of the command line idle; far above it they stall)
static int default_threads(int want) {
    const int hw = (int)std::thread::hardware_concurrency();
    int n = std::min(want, hw > 0 ? hw : 8);
    const int q = cpu_quota();
    if (q > 0) n = std::min(n, std::max(8, 2 * q));
    return std::max(1, n);
}

// ---------------------------------------------------------------------------------------
// input: plain text, gzip stream, or BGZF (block-parallel inflate)
// ---------------------------------------------------------------------------------------
// A fixed set of worker threads that run one job at a time together with the caller: fill_bgzf inflates ~50 MB of text
// per call, twenty calls per GB -- spawning and joining 63 threads each time was about half of the inflate time
// (1 GB of text: 102 ms at 64 threads).
class WorkerPool {
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void()>* job_ = nullptr;
    uint64_t gen_ = 0;
    int pending_ = 0;
    bool stop_ = false;

    void loop(uint64_t seen) {   // seen: the generation at the time the thread was created (earlier jobs are not its)
        for (;;) {
            const std::function<void()>* job;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
                job = job_;
            }
            (*job)();
            {
                std::lock_guard<std::mutex> lk(m_);
                if (--pending_ == 0) done_.notify_one();
            }
        }
    }

public:
    WorkerPool() = default;
    WorkerPool(const WorkerPool&) = delete;
    WorkerPool& operator=(const WorkerPool&) = delete;
    ~WorkerPool() { shutdown(); }
    void shutdown() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : th_) t.join();
        th_.clear();
        stop_ = false;
    }
    // runs `job` on up to n - 1 pool threads and on the caller; returns when every one of them has returned
    void run(int n, const std::function<void()>& job) {
        const int want = std::max(0, n - 1);
        if ((int)th_.size() < want) {
            // a pool that grows between two jobs: the new threads must not take the finished job for a new one
            uint64_t g0;
            {
                std::lock_guard<std::mutex> lk(m_);
                g0 = gen_;
            }
            while ((int)th_.size() < want) th_.emplace_back([this, g0] { loop(g0); });
        }
        if (!th_.empty()) {
            {
                std::lock_guard<std::mutex> lk(m_);
                job_ = &job;
                pending_ = (int)th_.size();
                ++gen_;
            }
            cv_.notify_all();
        }
        job();
        if (!th_.empty()) {
            std::unique_lock<std::mutex> lk(m_);
            done_.wait(lk, [&] { return pending_ == 0; });
        }
    }
};

// The caller-side phases (harmonise, statSTR's rows, the record writers) run their per-chunk jobs on ONE process-wide pool
// instead of threads made and joined per call (16-32 threads, twice per batch in the writers: ~1-2 ms per batch of
// creation alone).  One job at a time (the lock): two callers at once take turns -- a writer thread's members, CRCs and
// newline scans among them (a pool of their own was measured in round 6: dumpSTR --zip no faster, level 1 5 % slower).
struct SharedPool {
    WorkerPool* pool = new WorkerPool;      // (never destroyed: no join of parked threads at process exit)
    std::mutex* mu = new std::mutex;
    pid_t owner = getpid();
    void run(int nt, const std::function<void()>& job) {
        if (nt <= 1) {
            job();
            return;
        }
        if (getpid() != owner) {     // a forked child: the parent's threads do not exist here -- a pool and a lock of its own
            pool = new WorkerPool;
            mu = new std::mutex;
            owner = getpid();
        }
        std::lock_guard<std::mutex> g(*mu);
        pool->run(nt, job);
    }
};
static void run_on_caller_pool(int nt, const std::function<void()>& job) {
    static SharedPool* sp = new SharedPool;
    sp->run(nt, job);
}

struct Source {
    FILE* fp = nullptr;
    gzFile gz = nullptr;
    bool bgzf = false, plain = false, eof = false;
    bool src_eof = false;             // the last read of compressed bytes came back short (end of file)
    int n_threads = 1;
    // compressed bytes not yet consumed (BGZF).  Not a std::vector: resize() must not zero 8 MB that fread overwrites.
    struct CBuf {
        TextBuf b;
        size_t size() const { return b.size(); }
        unsigned char* data() { return reinterpret_cast<unsigned char*>(b.data()); }
        const unsigned char* data() const { return reinterpret_cast<const unsigned char*>(b.data()); }
        void clear() { b.clear(); }
        void resize(size_t n) { b.resize(n); }
        void drop_front(size_t n) { b.erase(0, n); }
    } cbuf;
    size_t cpos = 0;
    // contiguous shards (trk_vcf_shard): the blocks from file offset end_coff on belong to the next rank.  The first of
    // them is still inflated (the last line this rank owns may end in it), one more per call after that; limit_pos is
    // the index in the text buffer where the foreign data begins (SIZE_MAX: not met yet).
    uint64_t cbuf_foff = 0;           // file offset of cbuf[0]
    uint64_t end_coff = UINT64_MAX;
    size_t limit_pos = SIZE_MAX;
    uint64_t n_inflated = 0, n_compressed = 0, n_blocks = 0;   // counters: bytes out / in, blocks
    // trk_vcf_set_inflate_hook: the members are inflated by the caller (on the device); the text buffer then holds the
    // heads of the lines only and the newlines come from the hook
    trk_vcf_inflate_hook hook = {nullptr, nullptr, nullptr};
    uint64_t abs_end = 0;             // stream offset of the end of the text handed to the reader so far
    int line_state = 0;               // tabs seen in the unfinished last line (the hook's carry)
    std::vector<uint64_t> dev_nls;    // newlines reported and not yet behind the reader: stream offsets | bit 63 = after '\r'

    bool open(const char* path, int threads, std::string& err) {
        n_threads = threads;
        fp = fopen(path, "rb");
        if (!fp) {
            err = std::string("cannot open ") + path;
            return false;
        }
        unsigned char h[18];
        size_t n = fread(h, 1, sizeof h, fp);
        rewind(fp);
        if (n >= 2 && h[0] == 0x1f && h[1] == 0x8b) {
            // BGZF: FEXTRA set, extra subfield 'B','C' of length 2
            bgzf = n >= 18 && (h[3] & 4) && h[12] == 'B' && h[13] == 'C' && h[14] == 2 && h[15] == 0;
            if (!bgzf) {
                fclose(fp);
                fp = nullptr;
                gz = gzopen(path, "rb");
                if (!gz) {
                    err = std::string("gzopen failed for ") + path;
                    return false;
                }
                gzbuffer(gz, 1 << 20);
            }
        } else {
            plain = true;
        }
        return true;
    }
    void close() {
        if (fp) fclose(fp);
        if (gz) gzclose(gz);
        fp = nullptr;
        gz = nullptr;
    }

    // append at least `want` decompressed bytes to out (fewer only at end of file)
    bool fill(TextBuf& out, size_t want, std::string& err) {
        size_t start = out.size();
        // (a shard that has reached the next rank's blocks takes one block per call: the caller only needs the end of
        // its last line)
        // (with an inflate hook: one run of members per call -- the hook says how many it likes at a time)
        while (!eof && out.size() - start < want && !(limit_pos != SIZE_MAX && out.size() > start) &&
               !(hook.inflate && hook.max_members > 0 && out.size() > start)) {
            if (plain || gz) {
                size_t chunk = std::max<size_t>(want, 1 << 22);
                size_t old = out.size();
                out.resize(old + chunk);
                long got = plain ? (long)fread(&out[old], 1, chunk, fp) : (long)gzread(gz, &out[old], (unsigned)chunk);
                if (got < 0) {
                    err = "read error";
                    return false;
                }
                out.resize(old + (size_t)got);
                if ((size_t)got < chunk) eof = true;
            } else {
                if (!fill_bgzf(out, want, err)) return false;
            }
        }
        return true;
    }

    WorkerPool pool;                   // the inflater threads (created with the first BGZF fill, kept until close)
    double t_fread = 0, t_resize = 0, t_inflate = 0;   // TRK_VCF_TIMING: where fill_bgzf's time goes (seconds, cumulative)
    struct Blk { size_t off, csize, isize, dst; };
    // Top up the compressed buffer and find the complete members that follow cpos (at most hook.max_members with a hook):
    // blks, their end p in cbuf and the bytes of text they hold.  `out_size`: where that text will start in the reader's
    // buffer (a shard's limit_pos).  false: an error (err).
    bool scan_run(size_t want, size_t out_size, std::vector<Blk>& blks, size_t& p, size_t& total, std::string& err) {
        auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double tf0 = now();
        // top up the compressed buffer
        if (cpos > 0 && cpos == cbuf.size()) {
            cbuf_foff += cbuf.size();
            cbuf.clear();
            cpos = 0;
        }
        size_t target = std::max<size_t>(want / 3, 8u << 20);
        // (TRK_VCF_READ_BYTES: an option of the tests -- many small fills on files of a few megabytes)
        if (const char* e = trk_opt("TRK_VCF_READ_BYTES")) target = (size_t)std::max(1L, atol(e));
        if (end_coff != UINT64_MAX) {
            // a shard reads up to its end plus one block; once there, one more block's worth per call
            const uint64_t have_to = cbuf_foff + cbuf.size();
            const uint64_t stop = end_coff + 2 * 65536;
            target = have_to < stop ? (size_t)std::min<uint64_t>(target, stop - have_to) : (size_t)65536 + 64;
            if (target < 65536 + 64) target = 65536 + 64;   // always room for one whole block
        }
        if (cbuf.size() - cpos < target) {
            if (cpos > 0) {
                cbuf.drop_front(cpos);
                cbuf_foff += cpos;
                cpos = 0;
            }
            size_t old = cbuf.size();
            cbuf.resize(old + target);
            size_t got = 0;
            const off_t foff = ftello(fp);
            if (target >= ((size_t)4 << 20) && n_threads > 1 && foff >= 0) {
                // the compressed bytes by several threads (pread of 1 MB-aligned slices): a single fread copies the
                // page cache at ~8 GB/s, 40 ms per GB of text that the inflaters wait for
                const int fd = fileno(fp);
                const int nt = std::min(n_threads, 16);
                const size_t slice = (((target + (size_t)nt - 1) / (size_t)nt) + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
                const int n_sl = (int)((target + slice - 1) / slice);
                std::vector<size_t> got_k((size_t)n_sl, 0);
                std::atomic<int> nx{0};
                unsigned char* dst = cbuf.data() + old;
                auto rd = [&]() {
                    for (;;) {
                        const int k = nx.fetch_add(1);
                        if (k >= n_sl) break;
                        const size_t b0 = (size_t)k * slice, len = std::min(slice, target - b0);
                        size_t g = 0;
                        while (g < len) {
                            const ssize_t r = pread(fd, dst + b0 + g, len - g, foff + (off_t)(b0 + g));
                            if (r <= 0) break;
                            g += (size_t)r;
                        }
                        got_k[(size_t)k] = g;
                    }
                };
                pool.run(std::min(nt, n_sl), rd);
                for (int k = 0; k < n_sl; ++k) {
                    got += got_k[(size_t)k];
                    if (got_k[(size_t)k] < std::min(slice, target - (size_t)k * slice)) break;
                }
                (void)fseeko(fp, foff + (off_t)got, SEEK_SET);
            } else {
                got = fread(cbuf.data() + old, 1, target, fp);
            }
            src_eof = got < target;
            cbuf.resize(old + got);
        }
        t_fread += now() - tf0;
        // index complete blocks
        blks.clear();
        p = cpos;
        total = 0;
        int beyond = 0;                   // blocks of the next shard taken by this call
        while (p + 18 <= cbuf.size()) {
            if (cbuf_foff + p >= end_coff) {
                if (limit_pos == SIZE_MAX) limit_pos = out_size + total;
                if (beyond++ >= 1) break;
            }
            const unsigned char* h = cbuf.data() + p;
            // gzip member with an extra field that holds the 'BC' subfield (BSIZE).  Nothing of the header is
            // trusted: a block must have room for its own header, extra field, a deflate stream and the trailer,
            // and inflates to at most 64 KiB (the BGZF limit) -- a corrupt or hostile file must not turn into an
            // out-of-bounds read of cbuf or a multi-gigabyte allocation.
            if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) {
                err = "corrupt BGZF block header";
                return false;
            }
            const size_t xlen = (size_t)h[10] | ((size_t)h[11] << 8);
            if (p + 12 + xlen > cbuf.size()) {
                if (xlen > 1024) {   // htslib writes 6; anything this long is not BGZF
                    err = "corrupt BGZF block (extra field)";
                    return false;
                }
                break;               // the rest of the header is not in the buffer yet
            }
            size_t bsize = 0;
            for (size_t q = 12; q + 4 <= 12 + xlen;) {
                const size_t slen = (size_t)h[q + 2] | ((size_t)h[q + 3] << 8);
                if (h[q] == 'B' && h[q + 1] == 'C' && slen == 2 && q + 6 <= 12 + xlen) {
                    bsize = ((size_t)h[q + 4] | ((size_t)h[q + 5] << 8)) + 1;
                    break;
                }
                q += 4 + slen;
            }
            if (bsize == 0 || bsize < 12 + xlen + 2 + 8) {
                err = "corrupt BGZF block (no BC subfield / impossible block size)";
                return false;
            }
            if (p + bsize > cbuf.size()) break;
            size_t isize = (size_t)h[bsize - 4] | ((size_t)h[bsize - 3] << 8) | ((size_t)h[bsize - 2] << 16) |
                           ((size_t)h[bsize - 1] << 24);
            if (isize > 65536) {
                err = "corrupt BGZF block (inflated size above 64 KiB)";
                return false;
            }
            if (hook.inflate && hook.max_members > 0 && (int64_t)blks.size() >= (int64_t)hook.max_members) break;
            blks.push_back({p, bsize, isize, total});
            total += isize;
            n_inflated += isize;
            n_compressed += bsize;
            ++n_blocks;
            p += bsize;
        }
        return true;
    }
    void hook_blocks(const std::vector<Blk>& blks, std::vector<trk_vcf_iblock>& ib) const {
        ib.resize(blks.size());
        const size_t c0 = blks[0].off;
        for (size_t i = 0; i < blks.size(); ++i) {
            const unsigned char* h = cbuf.data() + blks[i].off;
            const size_t xlen = (size_t)h[10] | ((size_t)h[11] << 8);
            ib[i].payload_off = blks[i].off - c0 + 12 + xlen;
            ib[i].payload_len = (uint32_t)(blks[i].csize - 12 - xlen - 8);
            ib[i].isize = (uint32_t)blks[i].isize;
            ib[i].dst = blks[i].dst;
        }
    }
    // A hook with submit / collect: TWO runs of members in flight -- run k + 1 is read and handed over (its upload, its
    // kernel's launch) before run k is waited for, so the file read and the upload go on behind run k's inflate.
    std::deque<size_t> inflight;       // bytes of text of the runs submitted and not collected
    uint64_t abs_submit = 0;           // stream offset behind the last submitted run
    bool src_done = false;             // no complete member is left in the file
    bool fill_bgzf_pipelined(TextBuf& out, size_t want, std::string& err) {
        auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double ti0 = now();
        std::vector<Blk> blks;
        std::vector<trk_vcf_iblock> ib;
        while (inflight.size() < 2 && !src_done) {
            size_t p = 0, total = 0;
            if (!scan_run(want, 0, blks, p, total, err)) return false;
            if (blks.empty()) {
                if (cbuf.size() - cpos == 0 || src_eof) {
                    src_done = true;
                    break;
                }
                err = "truncated BGZF block";
                return false;
            }
            hook_blocks(blks, ib);
            const size_t c0 = blks[0].off;
            const int rc = hook.submit(hook.user, cbuf.data() + c0, p - c0, ib.data(), (int)ib.size(), abs_submit, total);
            if (rc != 0) {
                err = "the inflate hook failed (" + std::to_string(rc) + ")";
                return false;
            }
            abs_submit += total;
            inflight.push_back(total);
            cpos = p;
        }
        if (inflight.empty()) {
            eof = true;
            return true;
        }
        const size_t total = inflight.front();
        inflight.pop_front();
        const size_t base = out.size();
        out.resize(base + total);
        const uint64_t* nl = nullptr;
        size_t n_nl = 0;
        const int rc = hook.collect(hook.user, total ? &out[base] : nullptr, &line_state, &nl, &n_nl);
        t_inflate += now() - ti0;
        if (rc != 0) {
            err = "the inflate hook failed (" + std::to_string(rc) + ")";
            return false;
        }
        for (size_t i = 0; i < n_nl; ++i) dev_nls.push_back(((nl[i] & ~(1ull << 63)) + abs_end) | (nl[i] & (1ull << 63)));
        abs_end += total;
        if (inflight.empty() && (src_done || (cpos == cbuf.size() && src_eof))) eof = true;
        return true;
    }
    bool fill_bgzf(TextBuf& out, size_t want, std::string& err) {
        if (hook.inflate && hook.submit && hook.collect) return fill_bgzf_pipelined(out, want, err);
        auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        std::vector<Blk> blks;
        size_t p = 0, total = 0;
        if (!scan_run(want, out.size(), blks, p, total, err)) return false;
        if (blks.empty()) {
            if (cbuf.size() - cpos == 0 || src_eof) {
                eof = true;
                return true;
            }
            err = "truncated BGZF block";
            return false;
        }
        size_t base = out.size();
        const double tr0 = now();
        out.resize(base + total);
        const double ti0 = now();
        t_resize += ti0 - tr0;
        if (hook.inflate) {
            std::vector<trk_vcf_iblock> ib;
            hook_blocks(blks, ib);
            const size_t c0 = blks[0].off;
            const uint64_t* nl = nullptr;
            size_t n_nl = 0;
            const int rc = hook.inflate(hook.user, cbuf.data() + c0, p - c0, ib.data(), (int)ib.size(), abs_end, total,
                                        total ? &out[base] : nullptr, &line_state, &nl, &n_nl);
            t_inflate += now() - ti0;
            if (rc != 0) {
                err = "the inflate hook failed (" + std::to_string(rc) + ")";
                return false;
            }
            for (size_t i = 0; i < n_nl; ++i) dev_nls.push_back(((nl[i] & ~(1ull << 63)) + abs_end) | (nl[i] & (1ull << 63)));
            abs_end += total;
            cpos = p;
            if (cpos == cbuf.size() && src_eof) eof = true;
            return true;
        }
        std::atomic<size_t> next{0};
        std::atomic<bool> bad{false};
        auto work = [&]() {
            void* ld = deflater().decompress ? deflater().alloc() : nullptr;   // one decompressor per worker
            struct Rel {
                void* p;
                ~Rel() { if (p) deflater().release(p); }
            } rel{ld};
            for (;;) {
                size_t i = next.fetch_add(1);
                if (i >= blks.size()) break;
                const Blk& b = blks[i];
                if (b.isize == 0) continue;
                if (ld) {
                    const unsigned char* h = cbuf.data() + b.off;
                    const size_t xlen = (size_t)h[10] | ((size_t)h[11] << 8);
                    size_t got = 0;
                    if (deflater().decompress(ld, h + 12 + xlen, b.csize - 12 - xlen - 8, &out[base + b.dst], b.isize,
                                              &got) != 0 || got != b.isize)
                        bad = true;
                    continue;
                }
                z_stream zs;
                memset(&zs, 0, sizeof zs);
                if (inflateInit2(&zs, -15) != Z_OK) {
                    bad = true;
                    return;
                }
                const unsigned char* h = cbuf.data() + b.off;
                size_t xlen = (size_t)h[10] | ((size_t)h[11] << 8);
                zs.next_in = const_cast<unsigned char*>(h + 12 + xlen);
                zs.avail_in = (unsigned)(b.csize - 12 - xlen - 8);
                zs.next_out = reinterpret_cast<unsigned char*>(&out[base + b.dst]);
                zs.avail_out = (unsigned)b.isize;
                int rc = inflate(&zs, Z_FINISH);
                inflateEnd(&zs);
                if (rc != Z_STREAM_END) bad = true;
            }
        };
        int nt = std::max(1, std::min<int>(n_threads, (int)blks.size()));
        pool.run(nt, work);
        t_inflate += now() - ti0;
        if (bad) {
            err = "BGZF inflate failed";
            return false;
        }
        cpos = p;
        if (cpos == cbuf.size() && src_eof) eof = true;
        return true;
    }
};

struct PlaneSel {
    std::string key;
    int kind, ncol;
};

// results of trk_vcf_harmonize, owned by the reader (valid until the next call)
struct HzStore {
    std::vector<int32_t> allele_off, n_str_classes, n_len_classes, hrun, period;
    std::vector<uint16_t> len_class, str_class;
    std::vector<double> len_class_value, allele_len;
    std::vector<int64_t> pos, tr_pos, end, key_off;
    std::vector<uint8_t> passing, status;
    std::string keys;
};

}  // namespace

struct trk_vcf {
    Source src;
    HzStore hz2[2];      // results of trk_vcf_harmonize, two sets taken in turn: a batch's tables stay valid during the NEXT
    int hz_i = 0;        // call too (the read-ahead thread harmonises batch n + 1 while batch n's tables are in use)
    std::string err;
    std::string header;
    std::vector<std::string> samples;
    std::vector<PlaneSel> planes;
    TextBuf buf;         // decompressed text: [consumed .. pending)
    size_t pos = 0;      // start of unconsumed text in buf
    std::vector<int64_t> line_off, line_end;     // of the batch being read
    std::vector<int32_t> field_off;
    // A batch stays valid during the NEXT trk_vcf_read_batch call (a caller's thread reads batch n + 1 while batch n
    // is harmonised, filtered and written): its text is left in `prev_text` when the unconsumed tail moves on, its
    // line tables in one of two kept sets.
    TextBuf prev_text;
    struct Kept {
        std::vector<int64_t> line_off, line_end;
        std::vector<int32_t> field_off;
        std::vector<int8_t> fmt_idx;
    } kept[2];
    // trk_vcf_skip_samples: read_batch stops a record at its FORMAT keys -- the caller parses the sample columns itself
    // (on the device: trk_parse_samples, include/trk.h) or asks for them later (trk_vcf_parse_samples).  fmt_idx
    // [n_records][1 + planes]: index of GT, then of every selected plane's key, among the record's FORMAT keys.
    bool skip_samples = false;
    std::vector<int8_t> fmt_idx;
    int kept_i = 0;
    int n_threads = 1;
    // contiguous shard of the file (trk_vcf_shard)
    bool sharded = false, skip_partial = false, shard_done = false;
    uint64_t plain_end = UINT64_MAX;   // plain text: file offset where the next rank's lines begin
    // trk_vcf_set_sample_map: sample s of the file -> column sample_map[s] of trk_vcf_batch.gt_mapped (-1: dropped)
    std::vector<int32_t> sample_map;
    int map_out = 0;
    uint64_t abs0 = 0;                 // inflate hook: stream offset of buf[0]
};

namespace {

inline const char* find_ch(const char* p, const char* e, char c) {
    const void* r = memchr(p, c, (size_t)(e - p));
    return r ? static_cast<const char*>(r) : e;
}

inline int32_t parse_int(const char* p, const char* e) {
    if (e - p >= 2 && *p == '+' && (unsigned)(p[1] - '0') <= 9u) ++p;      // an explicit plus sign (strtol takes it)
    if (p == e || (e - p == 1 && *p == '.')) return INT_MISSING;
    long long v = 0;
    auto r = std::from_chars(p, e, v);
    if (r.ec != std::errc()) return INT_MISSING;
    return (int32_t)v;
}

inline float parse_float(const char* p, const char* e) {
    if (p == e || (e - p == 1 && *p == '.')) return NAN;
    double d = 0;
    auto r = std::from_chars(p, e, d);  // text -> double -> float (htslib / python float() path)
    if (r.ec != std::errc()) {
        std::string s(p, e);
        d = strtod(s.c_str(), nullptr);  // nan / inf spellings
    }
    return (float)d;
}

// one subfield -> ncol values
inline void parse_list(const char* p, const char* e, int kind, int ncol, void* dst) {
    if (kind == TRK_VCF_FLOAT) {
        float* o = static_cast<float*>(dst);
        int j = 0;
        while (j < ncol) {
            const char* q = find_ch(p, e, ',');
            o[j++] = parse_float(p, q);
            if (q == e) break;
            p = q + 1;
        }
        for (; j < ncol; ++j) o[j] = NAN;
    } else {
        int32_t* o = static_cast<int32_t*>(dst);
        int j = 0;
        if (e - p == 1 && *p == '.') {
            o[0] = INT_MISSING;
            j = 1;
        } else {
            while (j < ncol) {
                const char* q = p;
                // a leading '-' belongs to the number; later '-' separate ranges when asked to
                if (q < e && *q == '-') ++q;
                while (q < e && *q != ',' && !(kind == TRK_VCF_INT_RANGES && *q == '-')) ++q;
                o[j++] = parse_int(p, q);
                if (q == e) break;
                p = q + 1;
            }
        }
        for (; j < ncol; ++j) o[j] = (j == 0) ? INT_MISSING : INT_VECTOR_END;
    }
}

// HipSTR minimum supporting reads of one call (filters.py:519-567 pre-parse)
inline int32_t min_supp(const char* gb, const char* gbe, const char* ar, const char* are) {
    if (!ar || ar == are || (are - ar == 1 && *ar == '.')) return 0;
    if (!gb || gb == gbe) return 0;
    int32_t best = INT32_MAX;
    const char* p = gb;
    while (p <= gbe) {
        const char* q = p;
        if (q < gbe && *q == '-') ++q;
        while (q < gbe && *q != '|' && *q != '/') ++q;
        int32_t allele = parse_int(p, q);
        int32_t count = 0;
        const char* a = ar;
        while (a < are) {
            const char* semi = find_ch(a, are, ';');
            const char* bar = find_ch(a, semi, '|');
            if (bar < semi && parse_int(a, bar) == allele) {
                count = parse_int(bar + 1, semi);
                break;
            }
            a = semi + 1;
        }
        best = std::min(best, count);
        if (q >= gbe) break;
        p = q + 1;
    }
    return best == INT32_MAX ? 0 : best;
}

struct RecordJob {
    trk_vcf* v;
    const char* text;
    int S, P;
    trk_vcf_batch* out;
    const int64_t* line_off;
    const int64_t* line_end;
    int32_t* field_off;
    int8_t* fmt_idx = nullptr;     // [n][1 + planes] (skip_samples)
    bool samples = true;           // false: stop at the FORMAT keys
    std::atomic<int> error{0};
    std::atomic<int> error_rec{-1};   // a record the error was met in
};

void parse_record(RecordJob& job, int rec) {
    trk_vcf* v = job.v;
    const int S = job.S, P = job.P;
    const char* line = job.text + job.line_off[rec];
    const char* end = job.text + job.line_end[rec];
    int32_t* foff = &job.field_off[(size_t)rec * 10];
    const char* p = line;
    for (int k = 0; k < 10; ++k) {
        foff[k] = (int32_t)(p - line);
        if (k == 9) break;
        const char* t = find_ch(p, end, '\t');
        if (t == end) {
            for (int kk = k + 1; kk < 10; ++kk) foff[kk] = (int32_t)(end - line);
            break;
        }
        p = t + 1;
    }
    int16_t* gt = job.out->gt + (size_t)rec * S * P;
    uint8_t* ph = job.out->phased ? job.out->phased + (size_t)rec * S : nullptr;
    if (job.samples)
        for (size_t i = 0; i < (size_t)S * P; ++i) gt[i] = -2;
    // the same genotypes a second time with the samples in the caller's column order (trk_vcf_set_sample_map):
    // columns no sample maps to are no-calls
    const int32_t* smap = (job.out->gt_mapped && !v->sample_map.empty()) ? v->sample_map.data() : nullptr;
    int16_t* gtm = smap ? job.out->gt_mapped + (size_t)rec * v->map_out * P : nullptr;
    if (gtm && job.samples)
        for (size_t i = 0; i < (size_t)v->map_out * P; ++i) gtm[i] = -1;
    if (ph && job.samples) memset(ph, 0, (size_t)S);
    const int np = (int)v->planes.size();
    // FORMAT keys -> subfield index of GT and of every selected plane's inputs
    const char* fmt = line + foff[8];
    const char* fmt_end = (foff[9] > foff[8]) ? line + foff[9] - 1 : end;
    int gt_idx = -1;
    std::vector<int> pidx(np, -1), pidx2(np, -1);
    {
        int k = 0;
        const char* a = fmt;
        while (a <= fmt_end && a < end) {
            const char* b = find_ch(a, fmt_end, ':');
            size_t n = (size_t)(b - a);
            if (n == 2 && a[0] == 'G' && a[1] == 'T') gt_idx = k;
            for (int i = 0; i < np; ++i) {
                const PlaneSel& ps = v->planes[i];
                if (ps.kind == TRK_VCF_MINSUPP) {
                    if (n == 2 && a[0] == 'G' && a[1] == 'B') pidx[i] = k;
                    if (n == 8 && memcmp(a, "ALLREADS", 8) == 0) pidx2[i] = k;
                } else if (ps.key.size() == n && memcmp(ps.key.data(), a, n) == 0) {
                    pidx[i] = k;
                }
            }
            ++k;
            if (b >= fmt_end) break;
            a = b + 1;
        }
    }
    if (job.fmt_idx) {
        int8_t* fi = job.fmt_idx + (size_t)rec * (size_t)(1 + np);
        fi[0] = (int8_t)(gt_idx > 126 ? -1 : gt_idx);
        for (int i = 0; i < np; ++i) fi[1 + i] = (int8_t)(pidx[i] > 126 ? -1 : pidx[i]);
    }
    if (!job.samples) return;
    int max_needed = gt_idx;
    for (int i = 0; i < np; ++i) max_needed = std::max(max_needed, std::max(pidx[i], pidx2[i]));
    int maxpl = 1;
    const char* sp = line + foff[9];
    std::vector<const char*> sub_b((size_t)max_needed + 2), sub_e((size_t)max_needed + 2);
    // The fast form of a sample (round 4): ONE forward scan of the token up to the last subfield that is needed --
    // alleles of at most five digits, scalar Integer planes of at most nine, scalar Float planes as digits[.digits] of
    // at most fifteen (w / 10^k in float64 is the correctly rounded double of such a decimal, Clinger's fast path: the
    // same double from_chars gives, then the same cast) -- no memchr per token and subfield, no from_chars.  Anything
    // else about a token (exponents, signs on alleles, vectors, inf / nan, more alleles than the tensor holds) sends
    // THAT sample through the general code below.  Needs the byte at `end` readable: the line's newline.
    static const double kP10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                    1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    int8_t plane_at[32];
    bool fast_rec = np <= 32 && max_needed >= 0 && max_needed < 32 && (*end == '\n' || *end == '\r') && !(trk_opt("TRK_VCF_PARSE_GENERIC") && atoi(trk_opt("TRK_VCF_PARSE_GENERIC")) != 0);
    if (fast_rec) {
        for (int k = 0; k < 32; ++k) plane_at[k] = -1;
        for (int i = 0; i < np && fast_rec; ++i) {
            const PlaneSel& ps = v->planes[i];
            if ((ps.kind != TRK_VCF_INT && ps.kind != TRK_VCF_FLOAT) || ps.ncol != 1) fast_rec = false;
            else if (pidx[i] >= 0) {
                if (plane_at[pidx[i]] >= 0 || pidx[i] == gt_idx) fast_rec = false;    // two planes of one subfield
                else plane_at[pidx[i]] = (int8_t)i;
            }
        }
    }
#define TRK_DIGIT(ch) ((unsigned)((ch) - '0') <= 9u)
    for (int s = 0; s < S; ++s) {
        if (fast_rec) {
            const char* c = sp;
            int k = 0, j = 0;
            bool phased = false, ok = true;
            uint32_t got = 0;                    // planes parsed from this token
            for (;;) {
                if (k == gt_idx) {
                    for (;;) {
                        int a;
                        if (*c == '.') {
                            a = -1;
                            ++c;
                        } else if (TRK_DIGIT(*c)) {
                            const char* a0 = c;
                            a = 0;
                            do a = a * 10 + (*c++ - '0'); while (TRK_DIGIT(*c));
                            if (c - a0 > 4) { ok = false; break; }
                        } else if (*c == '/' || *c == '|' || *c == ':' || *c == '\t' || c == end) {
                            a = -1;              // an empty allele
                        } else {
                            ok = false;
                            break;
                        }
                        if (j >= P) { ok = false; break; }
                        gt[(size_t)s * P + j++] = (int16_t)a;
                        if (*c == '|') phased = true;
                        else if (*c != '/') break;
                        ++c;
                    }
                    if (!ok) break;
                } else if (plane_at[k] >= 0) {
                    const int i = plane_at[k];
                    char* base = static_cast<char*>(job.out->planes[i]);
                    if (v->planes[i].kind == TRK_VCF_INT) {
                        int32_t x;
                        if (*c == '.' && (c[1] == ':' || c[1] == '\t' || c + 1 == end)) {
                            x = INT_MISSING;
                            ++c;
                        } else {
                            const bool neg = *c == '-';
                            if (neg) ++c;
                            if (!TRK_DIGIT(*c)) { ok = false; break; }
                            const char* a0 = c;
                            uint32_t u = 0;
                            do u = u * 10 + (uint32_t)(*c++ - '0'); while (TRK_DIGIT(*c));
                            if (c - a0 > 9) { ok = false; break; }
                            x = neg ? -(int32_t)u : (int32_t)u;
                        }
                        reinterpret_cast<int32_t*>(base)[(size_t)rec * S + s] = x;
                    } else {
                        float x;
                        if (*c == '.' && (c[1] == ':' || c[1] == '\t' || c + 1 == end)) {
                            x = NAN;
                            ++c;
                        } else {
                            const bool neg = *c == '-';
                            if (neg) ++c;
                            const char* a0 = c;
                            uint64_t w = 0;
                            while (TRK_DIGIT(*c)) w = w * 10 + (uint64_t)(*c++ - '0');
                            int nd = (int)(c - a0), kf = 0;
                            if (*c == '.') {
                                const char* f0 = ++c;
                                while (TRK_DIGIT(*c)) w = w * 10 + (uint64_t)(*c++ - '0');
                                kf = (int)(c - f0);
                                nd += kf;
                            }
                            if (nd < 1 || nd > 15) { ok = false; break; }     // ('.' alone, or more digits than a double holds)
                            const double d = (double)w / kP10[kf];
                            x = (float)(neg ? -d : d);
                        }
                        reinterpret_cast<float*>(base)[(size_t)rec * S + s] = x;
                    }
                    got |= 1u << i;
                } else {
                    while (*c != ':' && *c != '\t' && *c != '\n' && *c != '\r') ++c;
                }
                if (*c == ':') {
                    if (k == max_needed) break;          // the rest of the token is not needed
                    ++c;
                    ++k;
                    continue;
                }
                if (*c == '\t' || c == end) break;
                ok = false;                              // something the scan does not know (an exponent, a comma ...)
                break;
            }
            if (ok) {
                if (gt_idx >= 0 && j == 0) gt[(size_t)s * P] = -1;     // the token ends before its GT: a missing call (see below)
                if (ph) ph[s] = phased ? 1 : 0;
                maxpl = std::max(maxpl, j);
                if (gtm && smap[s] >= 0)
                    for (int jj = 0; jj < P; ++jj) gtm[(size_t)smap[s] * P + jj] = gt[(size_t)s * P + jj];
                for (int i = 0; i < np; ++i)
                    if (!((got >> i) & 1u)) {            // absent from FORMAT, or dropped at the end of the token
                        char* base = static_cast<char*>(job.out->planes[i]);
                        if (v->planes[i].kind == TRK_VCF_INT) reinterpret_cast<int32_t*>(base)[(size_t)rec * S + s] = INT_MISSING;
                        else reinterpret_cast<float*>(base)[(size_t)rec * S + s] = NAN;
                    }
                const char* se = (*c == '\t' || c == end) ? c : find_ch(c, end, '\t');
                if (se == end) {
                    if (s + 1 < S) { job.error = 3; job.error_rec = rec; }
                    break;
                }
                sp = se + 1;
                continue;
            }
            for (int jj = 0; jj < P; ++jj) gt[(size_t)s * P + jj] = -2;      // the general code starts from padding
        }
        const char* se = find_ch(sp, end, '\t');
        // subfield boundaries up to the last one we need
        int nsub = 0;
        {
            const char* a = sp;
            while (nsub <= max_needed) {
                const char* b = find_ch(a, se, ':');
                sub_b[nsub] = a;
                sub_e[nsub] = b;
                ++nsub;
                if (b == se) break;
                a = b + 1;
            }
        }
        if (gt_idx >= 0 && gt_idx < nsub) {
            const char* a = sub_b[gt_idx];
            const char* e = sub_e[gt_idx];
            int j = 0;
            bool phased = false;
            while (a <= e) {
                const char* b = a;
                while (b < e && *b != '/' && *b != '|') ++b;
                if (j >= P) {
                    job.error = 2;  // ploidy above the tensor's P
                    job.error_rec = rec;
                    break;
                }
                gt[(size_t)s * P + j] = (b - a == 1 && *a == '.') || b == a ? (int16_t)-1 : (int16_t)parse_int(a, b);
                ++j;
                if (b >= e) break;
                if (*b == '|') phased = true;
                a = b + 1;
            }
            if (ph) ph[s] = phased ? 1 : 0;
            maxpl = std::max(maxpl, j);
        } else if (gt_idx >= 0) {
            // the record has a GT key and this sample's token ends before it (GT is not the first key and the trailing
            // fields are dropped): a missing call, '.', as htslib fills a dropped field
            gt[(size_t)s * P] = -1;
            if (ph) ph[s] = 0;
        }
        if (gtm && smap[s] >= 0)
            for (int j = 0; j < P; ++j) gtm[(size_t)smap[s] * P + j] = gt[(size_t)s * P + j];
        for (int i = 0; i < np; ++i) {
            const PlaneSel& ps = v->planes[i];
            char* base = static_cast<char*>(job.out->planes[i]);
            if (ps.kind == TRK_VCF_MINSUPP) {
                int32_t* o = reinterpret_cast<int32_t*>(base) + ((size_t)rec * S + s);
                const bool have_gb = pidx[i] >= 0 && pidx[i] < nsub;
                const bool have_ar = pidx2[i] >= 0 && pidx2[i] < nsub;
                *o = (have_gb && have_ar) ? min_supp(sub_b[pidx[i]], sub_e[pidx[i]], sub_b[pidx2[i]], sub_e[pidx2[i]]) : 0;
                continue;
            }
            const size_t esz = 4;
            void* o = base + ((size_t)rec * S + s) * ps.ncol * esz;
            if (pidx[i] >= 0 && pidx[i] < nsub) {
                parse_list(sub_b[pidx[i]], sub_e[pidx[i]], ps.kind, ps.ncol, o);
            } else {  // field absent from FORMAT or dropped at the end of the sample column: missing
                static const char dot = '.';
                parse_list(&dot, &dot + 1, ps.kind, ps.ncol, o);
            }
        }
        if (se == end) {
            if (s + 1 < S) { job.error = 3; job.error_rec = rec; }  // fewer sample columns than the header announces
            break;
        }
        sp = se + 1;
    }
#undef TRK_DIGIT
    job.out->locus_ploidy[rec] = (uint8_t)maxpl;
}

}  // namespace

extern "C" {

const char* trk_vcf_last_error(trk_vcf* v) { return v ? v->err.c_str() : g_open_error.c_str(); }

int trk_vcf_open(const char* path, int n_threads, trk_vcf** out) {
    if (!out || !path) return 2;
    *out = nullptr;
    trk_vcf* v = new trk_vcf();
    // all cores, but not more than 64 threads: the workers are started per batch, and beyond that their start-up
    // costs more than the extra hands bring (a 50 MB batch parses in ~3 ms on 64 threads)
    if (n_threads <= 0)
        if (const char* e = getenv("TRK_VCF_THREADS")) n_threads = atoi(e);     // inflate / parse threads of a reader
    if (n_threads <= 0) n_threads = default_threads(64);
    if (n_threads < 1) n_threads = 1;
    v->n_threads = n_threads;
    if (!v->src.open(path, n_threads, g_open_error)) {
        delete v;
        return 1;
    }
    // header: everything up to and including the #CHROM line
    for (;;) {
        size_t scan = v->pos;
        bool done = false;
        while (scan < v->buf.size()) {
            size_t nl = v->buf.find('\n', scan);
            if (nl == std::string::npos) break;
            if (v->buf[scan] != '#') {
                done = true;
                break;
            }
            bool chrom = v->buf.compare(scan, 6, "#CHROM") == 0;
            v->header.append(v->buf.data() + scan, nl - scan + 1);
            if (chrom) {
                std::string line = v->buf.substr(scan, nl - scan);
                if (!line.empty() && line.back() == '\r') line.pop_back();
                size_t a = 0;
                int col = 0;
                while (a <= line.size()) {
                    size_t b = line.find('\t', a);
                    if (b == std::string::npos) b = line.size();
                    if (col >= 9) v->samples.push_back(line.substr(a, b - a));
                    ++col;
                    a = b + 1;
                }
                scan = nl + 1;
                v->pos = scan;
                done = true;
                break;
            }
            scan = nl + 1;
            v->pos = scan;
        }
        if (done) break;
        if (v->src.eof) break;
        if (!v->src.fill(v->buf, 1 << 20, g_open_error)) {
            v->src.close();
            delete v;
            return 1;
        }
    }
    if (v->header.find("#CHROM") == std::string::npos) {
        g_open_error = std::string(path) + " does not look like a VCF (no #CHROM line)";
        v->src.close();
        delete v;
        return 1;
    }
    *out = v;
    return 0;
}

int trk_vcf_seek(trk_vcf* v, uint64_t voffset) {
    if (!v) return 2;
    if (v->src.hook.inflate) {
        v->err = "no seek while an inflate hook is installed";
        return 1;
    }
    if (!v->src.bgzf || !v->src.fp) {
        v->err = "seek needs a BGZF (bgzip) file";
        return 1;
    }
    const uint64_t coff = voffset >> 16;
    const size_t uoff = (size_t)(voffset & 0xffffu);
    // the offset must be the start of a BGZF block of THIS file (a stale .tbi points elsewhere):
    // checked before any reader state is touched, so that the caller can fall back to a scan
    const off_t here = ftello(v->src.fp);
    unsigned char h[18];
    const bool ok = fseeko(v->src.fp, (off_t)coff, SEEK_SET) == 0 && fread(h, 1, sizeof h, v->src.fp) == sizeof h &&
                    h[0] == 0x1f && h[1] == 0x8b && (h[3] & 4) && h[12] == 'B' && h[13] == 'C' &&
                    uoff <= 0xff00u;
    if (!ok) {
        (void)fseeko(v->src.fp, here, SEEK_SET);
        v->err = "virtual offset does not point at a BGZF block (stale index?)";
        return 1;
    }
    if (fseeko(v->src.fp, (off_t)coff, SEEK_SET) != 0) {
        v->err = "fseek failed";
        return 1;
    }
    v->src.cbuf.clear();
    v->src.cpos = 0;
    v->src.cbuf_foff = coff;
    v->src.end_coff = UINT64_MAX;
    v->src.limit_pos = SIZE_MAX;
    v->sharded = v->skip_partial = v->shard_done = false;
    v->src.eof = v->src.src_eof = false;
    v->buf.clear();
    v->pos = 0;
    v->line_off.clear();
    v->line_end.clear();
    // the first block must be inflated before the in-block offset can be applied
    while (v->buf.size() < uoff && !v->src.eof)
        if (!v->src.fill(v->buf, 1 << 16, v->err)) return 1;
    if (v->buf.size() < uoff) {
        v->err = "virtual offset beyond the end of its block";
        return 1;
    }
    v->pos = uoff;
    return 0;
}

// ---- contiguous shards --------------------------------------------------------------------
namespace {
struct BlockHdr { size_t bsize, isize, xlen; };
// a BGZF block header at p[0..n): sizes, or false
bool bgzf_header(const unsigned char* p, size_t n, BlockHdr& h) {
    if (n < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return false;
    h.xlen = (size_t)p[10] | ((size_t)p[11] << 8);
    if (h.xlen > 1024 || 12 + h.xlen > n) return false;
    h.bsize = 0;
    for (size_t q = 12; q + 4 <= 12 + h.xlen;) {
        const size_t slen = (size_t)p[q + 2] | ((size_t)p[q + 3] << 8);
        if (p[q] == 'B' && p[q + 1] == 'C' && slen == 2 && q + 6 <= 12 + h.xlen) {
            h.bsize = ((size_t)p[q + 4] | ((size_t)p[q + 5] << 8)) + 1;
            break;
        }
        q += 4 + slen;
    }
    return h.bsize >= 12 + h.xlen + 2 + 8;
}
// the first BGZF block that starts at or after file offset c: a candidate header whose BSIZE chain reaches two more
// valid headers (or the end of the file).  fsize when there is none.
uint64_t next_block(FILE* fp, uint64_t c, uint64_t fsize) {
    if (c >= fsize) return fsize;
    std::vector<unsigned char> w((size_t)std::min<uint64_t>(fsize - c, 4 * 65536 + 4096));
    if (fseeko(fp, (off_t)c, SEEK_SET) != 0 || fread(w.data(), 1, w.size(), fp) != w.size()) return fsize;
    for (size_t p = 0; p + 18 <= w.size() && p <= 65536 + 64; ++p) {
        size_t q = p;
        int ok = 0;
        while (ok < 3) {
            BlockHdr h;
            if (c + q == fsize) { ok = 3; break; }
            if (q + 18 > w.size()) { ok = ok >= 1 ? 3 : 0; break; }   // ran out of window after >= 1 full block
            if (!bgzf_header(w.data() + q, w.size() - q, h) || c + q + h.bsize > fsize) { ok = 0; break; }
            q += h.bsize;
            ++ok;
        }
        if (ok >= 3) return c + p;
    }
    return fsize;
}
// start of the block that ends exactly at file offset b, found by WALKING the block chain forward from a validated
// block start a few blocks earlier (no guessing from bytes that merely look like a header); UINT64_MAX: not found
uint64_t block_before(FILE* fp, uint64_t b, uint64_t fsize) {
    if (b == 0) return UINT64_MAX;
    uint64_t p = next_block(fp, b > 3 * 65536 + 4096 ? b - (3 * 65536 + 4096) : 0, fsize);
    unsigned char h[18 + 1024];
    while (p < b) {
        const size_t want = (size_t)std::min<uint64_t>(sizeof h, fsize - p);
        if (fseeko(fp, (off_t)p, SEEK_SET) != 0 || fread(h, 1, want, fp) != want) return UINT64_MAX;
        BlockHdr hd;
        if (!bgzf_header(h, want, hd)) return UINT64_MAX;
        if (p + hd.bsize == b) return p;
        p += hd.bsize;
    }
    return UINT64_MAX;
}

// last byte of the uncompressed data that precedes file offset b (the block(s) ending exactly at b); -1: nothing before
int last_byte_before(FILE* fp, uint64_t b, uint64_t fsize) {
    // first by the block chain (exact); the backward scan below only if the chain does not land on b
    for (uint64_t e = b; e > 0;) {
        const uint64_t q = block_before(fp, e, fsize);
        if (q == UINT64_MAX) break;
        std::vector<unsigned char> w((size_t)(e - q));
        if (fseeko(fp, (off_t)q, SEEK_SET) != 0 || fread(w.data(), 1, w.size(), fp) != w.size()) break;
        BlockHdr h;
        if (!bgzf_header(w.data(), w.size(), h) || h.bsize != w.size()) break;
        const size_t isize = (size_t)w[w.size() - 4] | ((size_t)w[w.size() - 3] << 8) | ((size_t)w[w.size() - 2] << 16) |
                             ((size_t)w[w.size() - 1] << 24);
        if (isize > 65536) break;
        if (isize == 0) {          // an empty block: the data before it decides
            e = q;
            continue;
        }
        std::vector<unsigned char> o(isize);
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit2(&zs, -15) != Z_OK) break;
        zs.next_in = w.data() + 12 + h.xlen;
        zs.avail_in = (unsigned)(w.size() - 12 - h.xlen - 8);
        zs.next_out = o.data();
        zs.avail_out = (unsigned)isize;
        const int rc = inflate(&zs, Z_FINISH);
        inflateEnd(&zs);
        if (rc != Z_STREAM_END) break;
        return o[isize - 1];
    }
    while (b > 0) {
        const uint64_t lo = b > 65536 + 1024 ? b - (65536 + 1024) : 0;
        std::vector<unsigned char> w((size_t)(b - lo));
        if (fseeko(fp, (off_t)lo, SEEK_SET) != 0 || fread(w.data(), 1, w.size(), fp) != w.size()) return -1;
        bool found = false;
        for (size_t back = 28; back <= w.size(); ++back) {     // (an empty block is 28 bytes)
            const size_t p = w.size() - back;
            BlockHdr h;
            if (!bgzf_header(w.data() + p, back, h) || h.bsize != back) continue;
            const size_t isize = (size_t)w[p + back - 4] | ((size_t)w[p + back - 3] << 8) | ((size_t)w[p + back - 2] << 16) |
                                 ((size_t)w[p + back - 1] << 24);
            if (isize > 65536) continue;
            if (isize == 0) {          // empty block: look further back
                b = lo + p;
                found = true;
                break;
            }
            std::vector<unsigned char> o(isize);
            z_stream zs;
            memset(&zs, 0, sizeof zs);
            if (inflateInit2(&zs, -15) != Z_OK) return -1;
            zs.next_in = w.data() + p + 12 + h.xlen;
            zs.avail_in = (unsigned)(back - 12 - h.xlen - 8);
            zs.next_out = o.data();
            zs.avail_out = (unsigned)isize;
            const int rc = inflate(&zs, Z_FINISH);
            inflateEnd(&zs);
            if (rc != Z_STREAM_END) continue;
            return o[isize - 1];
        }
        if (!found) return -1;
    }
    return -1;
}
}  // namespace

extern "C" int trk_vcf_shard(trk_vcf* v, int rank, int world, uint64_t* begin_off, uint64_t* end_off) {
    if (!v || world < 1 || rank < 0 || rank >= world) return 2;
    if (world == 1) return 0;
    if (v->src.hook.inflate) {
        v->err = "no shard while an inflate hook is installed";
        return 1;
    }
    if (!v->src.fp || !(v->src.bgzf || v->src.plain)) {
        v->err = "contiguous shards need a BGZF (bgzip) or plain-text file";
        return 1;
    }
    FILE* fp = v->src.fp;
    if (fseeko(fp, 0, SEEK_END) != 0) { v->err = "fseek failed"; return 1; }
    const uint64_t fsize = (uint64_t)ftello(fp);
    auto cut = [&](int r) -> uint64_t {
        if (r <= 0) return 0;
        if (r >= world) return fsize;
        const uint64_t c = (uint64_t)((unsigned __int128)fsize * (unsigned)r / (unsigned)world);
        return v->src.bgzf ? next_block(fp, c, fsize) : c;
    };
    const uint64_t b0 = cut(rank), b1 = cut(rank + 1);
    // a line belongs to the rank in whose range it STARTS: the shard begins inside a foreign line unless the byte
    // before it is a newline
    bool partial = false;
    if (b0 > 0) {
        int last;
        if (v->src.bgzf) {
            last = last_byte_before(fp, b0, fsize);
        } else {
            unsigned char ch = 0;
            last = (fseeko(fp, (off_t)(b0 - 1), SEEK_SET) == 0 && fread(&ch, 1, 1, fp) == 1) ? ch : -1;
        }
        partial = last >= 0 && last != '\n';
    }
    if (fseeko(fp, (off_t)b0, SEEK_SET) != 0) { v->err = "fseek failed"; return 1; }
    v->src.cbuf.clear();
    v->src.cpos = 0;
    v->src.cbuf_foff = b0;
    v->src.eof = v->src.src_eof = false;
    v->src.limit_pos = SIZE_MAX;
    v->src.n_inflated = v->src.n_compressed = v->src.n_blocks = 0;
    v->buf.clear();
    v->pos = 0;
    v->line_off.clear();
    v->line_end.clear();
    v->sharded = true;
    v->shard_done = b0 >= b1;
    v->skip_partial = partial;
    if (v->src.bgzf) {
        v->src.end_coff = rank + 1 < world ? b1 : UINT64_MAX;
    } else {
        v->src.end_coff = UINT64_MAX;
        if (rank + 1 < world) v->src.limit_pos = (size_t)(b1 - b0);
    }
    if (begin_off) *begin_off = b0;
    if (end_off) *end_off = b1;
    return 0;
}

extern "C" void trk_vcf_counters(trk_vcf* v, uint64_t out[3]) {
    if (!v || !out) return;
    out[0] = v->src.n_inflated;
    out[1] = v->src.n_compressed;
    out[2] = v->src.n_blocks;
}

void trk_vcf_close(trk_vcf* v) {
    if (!v) return;
    v->src.close();
    delete v;
}

const char* trk_vcf_header(trk_vcf* v, size_t* len) {
    if (len) *len = v->header.size();
    return v->header.c_str();
}
int trk_vcf_n_samples(trk_vcf* v) { return (int)v->samples.size(); }
int trk_vcf_set_sample_map(trk_vcf* v, const int32_t* map, int32_t n_out) {
    if (!v) return 2;
    v->sample_map.clear();
    v->map_out = 0;
    if (!map) return 0;
    const int S = (int)v->samples.size();
    if (n_out < 1) { v->err = "sample map: n_out must be positive"; return 1; }
    std::vector<uint8_t> seen((size_t)n_out, 0);
    for (int s = 0; s < S; ++s) {
        if (map[s] < -1 || map[s] >= n_out) { v->err = "sample map: column out of range"; return 1; }
        if (map[s] >= 0) {
            if (seen[(size_t)map[s]]) { v->err = "sample map: two samples on one column"; return 1; }
            seen[(size_t)map[s]] = 1;
        }
    }
    v->sample_map.assign(map, map + S);
    v->map_out = n_out;
    return 0;
}
const char* trk_vcf_sample_name(trk_vcf* v, int i) {
    return (i >= 0 && i < (int)v->samples.size()) ? v->samples[(size_t)i].c_str() : "";
}

int trk_vcf_select_format(trk_vcf* v, const char* key, int kind, int ncol) {
    if (!v || !key || ncol < 1 || kind < 0 || kind > TRK_VCF_MINSUPP) return -1;
    if (kind == TRK_VCF_MINSUPP) ncol = 1;
    v->planes.push_back({key, kind, ncol});
    return (int)v->planes.size() - 1;
}

int trk_vcf_read_batch(trk_vcf* v, int max_records, int max_ploidy, trk_vcf_batch* out) {
    if (!v || !out || max_records < 1 || max_ploidy < 1) return 2;
    const int S = (int)v->samples.size();
    out->n_records = 0;
    out->max_ploidy = max_ploidy;
    // drop what the previous call handed out
    if (v->pos > 0) {
        // (the handed-out text is not moved: the tail is copied to the other buffer and the two change places)
        const size_t tail = v->buf.size() - v->pos;
        v->prev_text.resize(tail);
        if (tail) memcpy(v->prev_text.data(), v->buf.data() + v->pos, tail);
        v->buf.swap(v->prev_text);
        if (v->src.limit_pos != SIZE_MAX) v->src.limit_pos = v->src.limit_pos > v->pos ? v->src.limit_pos - v->pos : 0;
        v->abs0 += v->pos;
        v->pos = 0;
    }
    const bool hooked = v->src.hook.inflate != nullptr;
    if (hooked) {   // newlines behind the reader are done with
        size_t k = 0;
        while (k < v->src.dev_nls.size() && (v->src.dev_nls[k] & ~(1ull << 63)) < v->abs0) ++k;
        if (k) v->src.dev_nls.erase(v->src.dev_nls.begin(), v->src.dev_nls.begin() + (ptrdiff_t)k);
    }
    v->line_off.clear();
    v->line_end.clear();
    size_t scan = 0;
    if (v->shard_done) return 0;
    if (v->skip_partial) {
        // the shard begins inside a line the previous rank owns: drop up to and including its newline
        for (;;) {
            size_t nl = v->buf.find('\n', 0);
            if (nl != std::string::npos) {
                scan = nl + 1;
                break;
            }
            if (v->src.eof) {
                scan = v->buf.size();
                break;
            }
            if (!v->src.fill(v->buf, 1u << 16, v->err)) return 1;
        }
        v->skip_partial = false;
    }
    const bool timing = trk_opt("TRK_VCF_TIMING") != nullptr;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = timing ? now() : 0.0;
    double t_fill = 0.0;
    // The newlines of the text are looked for by the inflater pool, a megabyte per task, whenever the scan below runs out
    // of indexed text (a 60 KB record is one memchr over 60 KB: 200 MB per batch of 3355 records at 5000 samples was
    // 5-7 ms on the reader's one thread, next to 9 ms of inflate on 32).  Nothing is kept between calls: the text that
    // stays behind the batch's last line (at most one fill) is indexed again by the next call.
    std::vector<size_t> nls;          // (inflate hook: bit 63 of an entry = the byte before the newline is '\r')
    constexpr size_t CR_BIT = (size_t)1 << 63;
    size_t nls_i = 0, nls_to = scan, hook_i = 0;
    auto index_more = [&]() {
        const size_t a = nls_to, b = v->buf.size();
        nls_to = b;
        if (b <= a) return;
        if (hooked) {    // the text holds line heads only: the newlines are the hook's
            for (; hook_i < v->src.dev_nls.size(); ++hook_i) {
                const uint64_t e = v->src.dev_nls[hook_i];
                const uint64_t off = (e & ~(1ull << 63)) - v->abs0;
                if (off >= b) break;
                if (off >= a) nls.push_back((size_t)off | ((e >> 63) ? CR_BIT : 0));
            }
            return;
        }
        const char* base = v->buf.data();
        constexpr size_t CH = 1u << 20;
        const size_t nch = (b - a + CH - 1) / CH;
        if (nch <= 2 || v->n_threads <= 1) {
            for (const char* q = base + a; (q = static_cast<const char*>(memchr(q, '\n', (size_t)(base + b - q)))) != nullptr; ++q)
                nls.push_back((size_t)(q - base));
            return;
        }
        std::vector<std::vector<size_t>> part(nch);
        std::atomic<size_t> nx{0};
        const std::function<void()> job = [&]() {
            for (;;) {
                const size_t c = nx.fetch_add(1);
                if (c >= nch) break;
                const char* q = base + a + c * CH;
                const char* const qe = base + std::min(b, a + (c + 1) * CH);
                while ((q = static_cast<const char*>(memchr(q, '\n', (size_t)(qe - q)))) != nullptr) {
                    part[c].push_back((size_t)(q - base));
                    ++q;
                }
            }
        };
        v->src.pool.run((int)std::min<size_t>((size_t)v->n_threads, nch), job);
        for (const auto& pc : part) nls.insert(nls.end(), pc.begin(), pc.end());
    };
    auto next_nl = [&](size_t from) -> size_t {
        for (;;) {
            while (nls_i < nls.size() && (nls[nls_i] & ~CR_BIT) < from) ++nls_i;
            if (nls_i < nls.size()) return nls[nls_i];
            if (nls_to >= v->buf.size()) return std::string::npos;
            index_more();
        }
    };
    while ((int)v->line_off.size() < max_records) {
        if (scan >= v->src.limit_pos) {   // the next line starts in the next rank's blocks
            v->shard_done = true;
            break;
        }
        size_t nl = next_nl(scan);
        if (nl == std::string::npos) {
            if (v->src.eof) {
                if (scan < v->buf.size()) {  // last line without a newline
                    v->buf.push_back('\n');
                    if (hooked) {
                        v->src.dev_nls.push_back(v->abs0 + v->buf.size() - 1);
                        v->src.abs_end += 1;
                    }
                    continue;
                }
                break;
            }
            const double tf = timing ? now() : 0.0;
            // (with an inflate hook: runs of ~4500 members -- on the device a member is one wave's serial work of ~8 ms
            // whatever else runs, and the chip holds 4096 of them at a time: tools/inflate_probe.py)
            size_t hook_run = 288u << 20;
            if (const char* e = trk_opt("TRK_VCF_HOOK_RUN_MB")) hook_run = (size_t)std::max(1L, atol(e)) << 20;      // (lab: the run's size)
            if (!v->src.fill(v->buf, hooked ? hook_run : (16u << 20), v->err)) return 1;
            if (timing) t_fill += now() - tf;
            continue;
        }
        const bool cr_flag = (nl & CR_BIT) != 0;
        nl &= ~CR_BIT;
        size_t e = nl;
        if (e > scan && (hooked ? cr_flag : v->buf[e - 1] == '\r')) --e;
        if (e > scan && !(v->sharded && v->buf[scan] == '#')) {  // skip blank lines (and, in a shard, the header)
            v->line_off.push_back((int64_t)scan);
            v->line_end.push_back((int64_t)e);
        }
        scan = nl + 1;
    }
    const int n = (int)v->line_off.size();
    const double t1 = timing ? now() : 0.0;
    v->field_off.assign((size_t)n * 10, 0);
    RecordJob job;
    job.v = v;
    job.text = v->buf.data();
    job.S = S;
    job.P = max_ploidy;
    job.out = out;
    job.line_off = v->line_off.data();
    job.line_end = v->line_end.data();
    job.field_off = v->field_off.data();
    job.samples = !v->skip_samples;
    v->fmt_idx.assign(v->skip_samples ? (size_t)n * (1 + v->planes.size()) : 0, (int8_t)-1);
    job.fmt_idx = v->skip_samples ? v->fmt_idx.data() : nullptr;
    std::atomic<int> next{0};
    auto runner = [&]() {
        for (;;) {
            int i = next.fetch_add(1);
            if (i >= n) break;
            parse_record(job, i);
        }
    };
    int nt = std::max(1, std::min(v->n_threads, n));
    v->src.pool.run(nt, runner);     // (the reader's own pool: this is the thread that fills)
    if (job.error) {
        // nothing is consumed: the same lines are decoded again by the next call (the caller retries a ploidy
        // overflow with a wider genotype tensor, vcfnative.py)
        v->err = job.error == 2 ? "a genotype has more haplotypes than max_ploidy"
                                : "a record has fewer sample columns than the header";
        const int er = job.error_rec.load();
        if (er >= 0 && er < n) {            // name the record: CHROM:POS
            const char* line = v->buf.data() + v->line_off[(size_t)er];
            const char* lend = v->buf.data() + v->line_end[(size_t)er];
            const char* t1p = find_ch(line, lend, '\t');
            const char* t2p = t1p < lend ? find_ch(t1p + 1, lend, '\t') : lend;
            v->err += " (record " + std::string(line, (size_t)(t1p - line)) + ":" +
                      (t1p < lend ? std::string(t1p + 1, (size_t)(t2p - t1p - 1)) : std::string("?")) + ")";
        }
        return 5;
    }
    if (timing)
        fprintf(stderr, "[trk_vcf] batch of %d records: scan %.1f ms (of which inflate/read %.1f), parse %.1f ms, %d threads; "
                        "cumulative fread %.1f ms, buffer growth %.1f ms, inflate %.1f ms\n", n,
                (t1 - t0) * 1e3, t_fill * 1e3, (now() - t1) * 1e3, nt, v->src.t_fread * 1e3, v->src.t_resize * 1e3,
                v->src.t_inflate * 1e3);
    v->pos = scan;
    v->kept_i ^= 1;
    trk_vcf::Kept& kp = v->kept[v->kept_i];
    kp.line_off.swap(v->line_off);
    kp.line_end.swap(v->line_end);
    kp.field_off.swap(v->field_off);
    kp.fmt_idx.swap(v->fmt_idx);
    out->n_records = n;
    out->text = v->buf.data();
    out->line_off = kp.line_off.data();
    out->line_end = kp.line_end.data();
    out->field_off = kp.field_off.data();
    return 0;
}

int trk_vcf_set_inflate_hook(trk_vcf* v, const trk_vcf_inflate_hook* hook) {
    if (!v) return 2;
    if (!hook || !hook->inflate) {
        if (v->src.hook.inflate) {
            v->err = "the inflate hook cannot be taken away again: the text holds line heads only";
            return 2;
        }
        return 0;
    }
    if (!v->src.bgzf || !v->src.fp || v->sharded || v->src.end_coff != UINT64_MAX) {
        v->err = "the inflate hook needs a BGZF file read from its start to its end (no shard, no seek)";
        return 2;
    }
    if (!v->skip_samples) {
        v->err = "the inflate hook needs trk_vcf_skip_samples: only the heads of the lines reach the host";
        return 2;
    }
    // what has been inflated so far (the rest of trk_vcf_open's last fill) is real text: drop what is consumed, index its
    // newlines and the tabs of its unfinished last line here, and hand it to the hook as the start of the stream
    if (v->pos > 0) {
        v->buf.erase(0, v->pos);
        v->pos = 0;
    }
    v->abs0 = 0;
    v->src.dev_nls.clear();
    const char* b = v->buf.data();
    const size_t n = v->buf.size();
    size_t last = 0;
    for (const char* q = b; (q = static_cast<const char*>(memchr(q, '\n', (size_t)(b + n - q)))) != nullptr; ++q) {
        const size_t o = (size_t)(q - b);
        v->src.dev_nls.push_back((uint64_t)o | ((o > 0 && q[-1] == '\r') ? (1ull << 63) : 0));
        last = o + 1;
    }
    int tabs = 0;
    for (size_t i = last; i < n && tabs < 9; ++i) tabs += b[i] == '\t';
    v->src.line_state = tabs;
    v->src.abs_end = n;
    v->src.abs_submit = n;
    v->src.inflight.clear();
    v->src.src_done = false;
    if (hook->seed) {
        const int rc = hook->seed(hook->user, b, n);
        if (rc != 0) {
            v->err = "the inflate hook's seed failed (" + std::to_string(rc) + ")";
            return 2;
        }
    }
    v->src.hook = *hook;
    return 0;
}

uint64_t trk_vcf_text_abs(trk_vcf* v) { return v ? v->abs0 : 0; }

// The reader's two text buffers continue in the caller's memory (pinned pages: the batch's text is uploaded by plain DMA).
// Call between two batches, when the previous batch's text is no longer needed: its bytes move.
int trk_vcf_set_text_buffers(trk_vcf* v, void* a, void* b, size_t cap_each) {
    if (!v || !a || !b || cap_each == 0) return 2;
    if (v->buf.external() || v->prev_text.external()) return 0;
    if (v->buf.size() > cap_each || v->prev_text.size() > cap_each) return 1;
    v->buf.adopt(static_cast<char*>(a), cap_each);
    v->prev_text.adopt(static_cast<char*>(b), cap_each);
    return 0;
}

int trk_vcf_skip_samples(trk_vcf* v, int on) {
    if (!v) return 2;
    v->skip_samples = on != 0;
    return 0;
}

const int8_t* trk_vcf_format_idx(trk_vcf* v, int32_t* stride) {
    if (!v) return nullptr;
    if (stride) *stride = 1 + (int32_t)v->planes.size();
    const trk_vcf::Kept& kp = v->kept[v->kept_i];
    return kp.fmt_idx.empty() ? nullptr : kp.fmt_idx.data();
}

// the half of the last batch that trk_vcf_skip_samples left out: genotypes, phasing, planes, ploidies of every record
int trk_vcf_parse_samples(trk_vcf* v, trk_vcf_batch* b) {
    if (!v || !b) return 2;
    const int n = b->n_records;
    if (n <= 0) return 0;
    trk_vcf::Kept& kp = v->kept[v->kept_i];
    if ((int)kp.line_off.size() < n || b->text == nullptr) {
        v->err = "trk_vcf_parse_samples: not the batch of the last trk_vcf_read_batch";
        return 2;
    }
    RecordJob job;
    job.v = v;
    job.text = b->text;
    job.S = (int)v->samples.size();
    job.P = b->max_ploidy;
    job.out = b;
    job.line_off = kp.line_off.data();
    job.line_end = kp.line_end.data();
    job.field_off = kp.field_off.data();
    std::atomic<int> next{0};
    auto runner = [&]() {
        for (;;) {
            int i = next.fetch_add(1);
            if (i >= n) break;
            parse_record(job, i);
        }
    };
    const int nt = std::max(1, std::min(v->n_threads, n));
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(runner);
    runner();
    for (auto& t : th) t.join();
    if (job.error) {
        v->err = job.error == 2 ? "a genotype has more haplotypes than max_ploidy"
                                : "a record has fewer sample columns than the header";
        return 5;
    }
    return 0;
}

// ---- record serialisation (trk_vcf.h) ------------------------------------------------------
namespace {
struct OutBuf {
    char* p;
    int64_t cap, n;
    inline void put(char c) {
        if (n < cap) p[n] = c;
        ++n;
    }
    inline void put(const char* s, size_t len) {
        if (n + (int64_t)len <= cap) memcpy(p + n, s, len);
        n += (int64_t)len;
    }
    inline void put_int(long v) {
        char tmp[24];
        auto r = std::to_chars(tmp, tmp + sizeof tmp, v);
        put(tmp, (size_t)(r.ptr - tmp));
    }
};

inline void put_utf8(OutBuf& o, uint32_t c) {
    if (c < 0x80) {
        o.put((char)c);
    } else if (c < 0x800) {
        o.put((char)(0xc0 | (c >> 6)));
        o.put((char)(0x80 | (c & 0x3f)));
    } else if (c < 0x10000) {
        o.put((char)(0xe0 | (c >> 12)));
        o.put((char)(0x80 | ((c >> 6) & 0x3f)));
        o.put((char)(0x80 | (c & 0x3f)));
    } else {
        o.put((char)(0xf0 | (c >> 18)));
        o.put((char)(0x80 | ((c >> 12) & 0x3f)));
        o.put((char)(0x80 | ((c >> 6) & 0x3f)));
        o.put((char)(0x80 | (c & 0x3f)));
    }
}
}  // namespace

}  // extern "C"  (helpers below have C++ linkage)

namespace {
// samples [s0, s1) of the record into o
void format_range(int64_t s0, int64_t s1, int32_t n_columns, const trk_vcf_column* cols, OutBuf& o) {
    char tmp[48];
    for (int64_t s = s0; s < s1; ++s) {
        o.put('\t');
        for (int c = 0; c < n_columns; ++c) {
            const trk_vcf_column& f = cols[c];
            if (c) o.put(':');
            const int k = f.ncol;
            switch (f.kind) {
                case TRK_VCF_COL_GT: {
                    const int16_t* g = static_cast<const int16_t*>(f.data) + s * k;
                    const char sep = g[k - 1] ? '|' : '/';
                    int n = 0;
                    for (int j = 0; j < k - 1; ++j) {
                        if (g[j] == -2) continue;
                        if (n++) o.put(sep);
                        if (g[j] == -1) o.put('.');
                        else o.put_int(g[j]);
                    }
                    if (!n) o.put('.');
                    break;
                }
                case TRK_VCF_COL_INT: {
                    const int32_t* v = static_cast<const int32_t*>(f.data) + s * k;
                    int n = 0;
                    for (int j = 0; j < k; ++j) {
                        if (v[j] == INT32_MIN + 1) break;
                        if (n++) o.put(',');
                        if (v[j] == INT32_MIN) o.put('.');
                        else o.put_int(v[j]);
                    }
                    if (!n) o.put('.');
                    break;
                }
                case TRK_VCF_COL_FLOAT: {
                    const float* v = static_cast<const float*>(f.data) + s * k;
                    bool any = false;
                    for (int j = 0; j < k; ++j) any |= !std::isnan(v[j]);
                    if (!any) {
                        o.put('.');
                        break;
                    }
                    for (int j = 0; j < k; ++j) {
                        if (j) o.put(',');
                        if (std::isnan(v[j])) {
                            o.put('.');
                        } else {
                            // printf("%g"): to_chars with a precision is specified to give the same digits
                            auto r = std::to_chars(tmp, tmp + sizeof tmp, (double)v[j], std::chars_format::general, 6);
                            o.put(tmp, (size_t)(r.ptr - tmp));
                        }
                    }
                    break;
                }
                case TRK_VCF_COL_BYTES: {
                    const char* v = static_cast<const char*>(f.data) + s * (int64_t)f.itemsize;
                    size_t len = (size_t)f.itemsize;
                    while (len && v[len - 1] == 0) --len;
                    if (!len) o.put('.');
                    else o.put(v, len);
                    break;
                }
                case TRK_VCF_COL_CALLFILTER: {
                    const trk_vcf_callfilter* cf = static_cast<const trk_vcf_callfilter*>(f.data);
                    const uint32_t m = cf->mask[s];
                    if (m & 0x80000000u) {
                        o.put("NOCALL", 6);
                    } else if (m == 0) {
                        o.put("PASS", 4);
                    } else {
                        int n = 0;
                        for (int b = 0; b < cf->n_filters && b < 31; ++b) {
                            if (!((m >> b) & 1u)) continue;
                            if (n++) o.put(',');
                            o.put(cf->names[b], strlen(cf->names[b]));
                            o.put('_');
                            const double x = cf->values[b] ? cf->values[b][s] : NAN;
                            const int len = snprintf(tmp, sizeof tmp, "%g", x);
                            o.put(tmp, (size_t)len);
                        }
                        if (!n) o.put('.');
                    }
                    break;
                }
                default: {  // TRK_VCF_COL_UCS4
                    const int nch = f.itemsize / 4;
                    const uint32_t* v = reinterpret_cast<const uint32_t*>(static_cast<const char*>(f.data) +
                                                                          s * (int64_t)f.itemsize);
                    int len = nch;
                    while (len && v[len - 1] == 0) --len;
                    if (!len) o.put('.');
                    for (int j = 0; j < len; ++j) put_utf8(o, v[j]);
                    break;
                }
            }
        }
    }
}

// A small fork-join pool for the serialiser: records of thousands of samples are split by sample range.
class FmtPool {
   public:
    static FmtPool& get() {
        static FmtPool* p = new FmtPool;   // never destroyed: its workers outlive static destruction
        return *p;
    }
    // (a forked child inherits the object but not its threads: it formats serially)
    int size() const { return getpid() == pid_ ? (int)workers_.size() + 1 : 1; }
    // run fn(task) for task in [0, n) on the workers and the calling thread
    template <class F>
    void run(int n, F&& fn) {
        std::unique_lock<std::mutex> call_lock(call_mu_);   // one record at a time
        {
            std::lock_guard<std::mutex> g(mu_);
            fn_ = [&](int t) { fn(t); };
            n_tasks_ = n;
            next_ = 0;
            pending_ = n;
            ++gen_;
        }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> g(mu_);
        done_cv_.wait(g, [&] { return pending_ == 0; });
    }

   private:
    FmtPool() {
        int n = (int)std::thread::hardware_concurrency();
        if (const char* e = getenv("TRK_FMT_THREADS")) n = atoi(e);
        if (n > 8) n = 8;
        for (int i = 1; i < n; ++i) workers_.emplace_back([this] { loop(); });
        for (auto& w : workers_) w.detach();
    }
    // tasks are handed out under the mutex (a dozen per record): a worker that is late leaving the previous
    // record's loop sees either the old, exhausted state or the new one whole -- never a mixture
    void work() {
        while (true) {
            std::function<void(int)> f;
            int t;
            {
                std::lock_guard<std::mutex> g(mu_);
                if (next_ >= n_tasks_) break;
                t = next_++;
                f = fn_;
            }
            f(t);
            std::lock_guard<std::mutex> g(mu_);
            if (--pending_ == 0) done_cv_.notify_all();
        }
    }
    void loop() {
        uint64_t seen = 0;
        while (true) {
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [&] { return gen_ != seen; });
                seen = gen_;
            }
            work();
        }
    }
    std::vector<std::thread> workers_;
    pid_t pid_ = getpid();
    std::mutex mu_, call_mu_;
    std::condition_variable cv_, done_cv_;
    std::function<void(int)> fn_;
    int n_tasks_ = 0, pending_ = 0;
    int next_ = 0;
    uint64_t gen_ = 0;
};
}  // namespace

extern "C" {

int64_t trk_vcf_format_samples(int32_t n_samples, int32_t n_columns, const trk_vcf_column* cols, char* out,
                               int64_t cap) {
    if (n_samples < 0 || n_columns < 0 || (n_columns && !cols) || cap < 0 || (cap && !out)) return INT64_MIN;
    for (int c = 0; c < n_columns; ++c)
        if (cols[c].kind < TRK_VCF_COL_GT || cols[c].kind > TRK_VCF_COL_CALLFILTER || cols[c].ncol < 1 ||
            !cols[c].data || ((cols[c].kind >= TRK_VCF_COL_BYTES) && cols[c].itemsize < 0))
            return INT64_MIN;
    // large records: sample ranges formatted in parallel into private buffers, then laid end to end
    const int64_t work = (int64_t)n_samples * n_columns;
    FmtPool* pool = work >= 8192 ? &FmtPool::get() : nullptr;
    if (pool && pool->size() > 1) {
        const int T = (int)std::min<int64_t>(pool->size() * 2, n_samples / 256 + 1);
        std::vector<std::vector<char>> bufs((size_t)T);
        std::vector<int64_t> used((size_t)T, 0);
        pool->run(T, [&](int t) {
            const int64_t s0 = (int64_t)n_samples * t / T, s1 = (int64_t)n_samples * (t + 1) / T;
            std::vector<char>& b = bufs[(size_t)t];
            b.resize((size_t)((s1 - s0) * n_columns * 12 + 64));
            OutBuf o{b.data(), (int64_t)b.size(), 0};
            format_range(s0, s1, n_columns, cols, o);
            if (o.n > o.cap) {   // the guess was short: the count is exact now
                b.resize((size_t)o.n);
                o = OutBuf{b.data(), (int64_t)b.size(), 0};
                format_range(s0, s1, n_columns, cols, o);
            }
            used[(size_t)t] = o.n;
        });
        int64_t total = 0;
        for (int t = 0; t < T; ++t) total += used[(size_t)t];
        if (total > cap) return -total;
        int64_t at = 0;
        for (int t = 0; t < T; ++t) {
            memcpy(out + at, bufs[(size_t)t].data(), (size_t)used[(size_t)t]);
            at += used[(size_t)t];
        }
        return total;
    }
    OutBuf o{out, cap, 0};
    format_range(0, n_samples, n_columns, cols, o);
    return o.n <= cap ? o.n : -o.n;
}

// ---- typed decode of all FORMAT fields of one record (trk_vcf.h) -----------------------------
namespace {
inline int utf8_len(const char* p, const char* e) {   // code points in [p, e)
    int n = 0;
    for (; p < e; ++p) n += ((unsigned char)*p & 0xc0) != 0x80;
    return n;
}
inline void utf8_to_ucs4(const char* p, const char* e, uint32_t* out) {
    while (p < e) {
        const unsigned char c = (unsigned char)*p;
        uint32_t cp;
        int extra;
        if (c < 0x80) { cp = c; extra = 0; }
        else if ((c & 0xe0) == 0xc0) { cp = c & 0x1f; extra = 1; }
        else if ((c & 0xf0) == 0xe0) { cp = c & 0x0f; extra = 2; }
        else { cp = c & 0x07; extra = 3; }
        ++p;
        for (int i = 0; i < extra && p < e; ++i, ++p) cp = (cp << 6) | ((unsigned char)*p & 0x3f);
        *out++ = cp;
    }
}
}  // namespace

int trk_vcf_decode_formats(const char* samples, int64_t len, int32_t n_samples, int32_t n_fields,
                           trk_vcf_decode* fields, int32_t pass) {
    if (!samples || len < 0 || n_samples < 0 || n_fields < 0 || (n_fields && !fields)) return -1;
    const char* p = samples;
    const char* end = samples + len;
    while (end > p && (end[-1] == '\n' || end[-1] == '\r')) --end;
    if (pass == 0)
        for (int f = 0; f < n_fields; ++f) fields[f].ncol = 1;
    int64_t s = 0;
    if (p == end) return n_samples == 0 ? 0 : 1;
    static const char dot[] = ".";
    while (true) {
        if (s >= n_samples) return 1;
        const char* se = static_cast<const char*>(memchr(p, '\t', (size_t)(end - p)));
        if (!se) se = end;
        const char* q = p;
        bool exhausted = false;
        for (int f = 0; f < n_fields; ++f) {
            const char* tb;
            const char* te;
            if (exhausted) {
                tb = dot;
                te = dot + 1;
            } else {
                tb = q;
                te = static_cast<const char*>(memchr(q, ':', (size_t)(se - q)));
                if (!te) {
                    te = se;
                    exhausted = true;
                } else {
                    q = te + 1;
                }
            }
            trk_vcf_decode& d = fields[f];
            if (d.kind == TRK_VCF_COL_UCS4) {
                if (pass == 0) {
                    const int n = utf8_len(tb, te);
                    if (n > d.ncol) d.ncol = n;
                } else {
                    uint32_t* o = static_cast<uint32_t*>(d.out) + s * d.ncol;
                    utf8_to_ucs4(tb, te, o);   // the caller zero-filled the array
                }
            } else if (d.kind == TRK_VCF_COL_INT || d.kind == TRK_VCF_COL_FLOAT) {
                if (pass == 0) {
                    int n = 1;
                    for (const char* c = tb; c < te; ++c) n += *c == ',';
                    if (n > d.ncol) d.ncol = n;
                } else {
                    int j = 0;
                    const char* c = tb;
                    while (true) {
                        const char* ce = static_cast<const char*>(memchr(c, ',', (size_t)(te - c)));
                        if (!ce) ce = te;
                        if (j >= d.ncol) return -1;
                        const bool missing = ce - c == 1 && *c == '.';
                        if (d.kind == TRK_VCF_COL_INT) {
                            int32_t v = INT32_MIN;
                            if (!missing) {
                                auto r = std::from_chars(c, ce, v);
                                if (r.ec != std::errc() || r.ptr != ce) return 2;
                            }
                            static_cast<int32_t*>(d.out)[s * d.ncol + j] = v;
                        } else {
                            float v = NAN;
                            if (!missing) {
                                double x;
                                auto r = std::from_chars(c, ce, x);
                                if (r.ec == std::errc::result_out_of_range && r.ptr == ce) {
                                    x = strtod(std::string(c, ce).c_str(), nullptr);   // inf / denormal as Python's float()
                                } else if (r.ec != std::errc() || r.ptr != ce) {
                                    return 2;
                                }
                                v = (float)x;
                            }
                            static_cast<float*>(d.out)[s * d.ncol + j] = v;
                        }
                        ++j;
                        if (ce == te) break;
                        c = ce + 1;
                    }
                    if (d.kind == TRK_VCF_COL_INT)
                        for (; j < d.ncol; ++j) static_cast<int32_t*>(d.out)[s * d.ncol + j] = INT32_MIN + 1;
                    else
                        for (; j < d.ncol; ++j) static_cast<float*>(d.out)[s * d.ncol + j] = NAN;
                }
            }
        }
        ++s;
        if (se == end) break;
        p = se + 1;
    }
    return s == n_samples ? 0 : 1;
}

}  // extern "C"

// =====================================================================================================
// Batch harmonisation + statSTR rows: the host pipeline of the per-locus hot path without one Python object per
// record (SURVEY.md 8(b) "one call per batch of loci"; the reference streams record -> row, statSTR.py:575-639).
// =====================================================================================================
namespace {

// a[start:stop] of a Python sequence of length n (stop absent when !has_stop) -> [b, e)
inline void py_slice(long n, long start, bool has_stop, long stop, long& b, long& e) {
    b = start < 0 ? std::max(n + start, 0L) : std::min(start, n);
    e = !has_stop ? n : (stop < 0 ? std::max(n + stop, 0L) : std::min(stop, n));
    if (e < b) e = b;
}

// value of `key` in an INFO column "k=v;k2;k3=v3": false when absent
inline bool info_get(const char* s, const char* e, const char* key, size_t klen, const char*& vb, const char*& ve,
                     bool& has_value) {
    const char* p = s;
    while (p < e) {
        const char* t = static_cast<const char*>(memchr(p, ';', (size_t)(e - p)));
        if (!t) t = e;
        if ((size_t)(t - p) >= klen && memcmp(p, key, klen) == 0 && (p + klen == t || p[klen] == '=')) {
            has_value = p + klen < t;
            vb = has_value ? p + klen + 1 : t;
            ve = t;
            return true;
        }
        p = t + 1;
    }
    return false;
}

inline bool parse_long(const char* b, const char* e, long& out) {
    if (b == e) return false;
    auto r = std::from_chars(b, e, out);
    return r.ec == std::errc() && r.ptr == e;
}

struct HzRecord {  // one record's harmonised alleles (views into the line; upper-casing happens on copy)
    std::vector<std::pair<const char*, long>> alleles;  // trimmed [ptr, len)
    double unit = 1.0;                                   // len(motif)
    int64_t pos = 0, end = 0;
    int64_t tr_pos = 0;                                  // TRRecord.pos: INFO START for HipSTR (tr_harmonizer.py:407), else POS
    int32_t hrun = 0;
    int32_t period = INT32_MIN;                          // INFO PERIOD (an integer), INT32_MIN when absent
    bool ok = false, passing = false;
    // ExpansionHunter / PopSTR: alleles given by LENGTH (<STRn>, <n>) are fabricated from the motif (utils.py:566-602);
    // `lengths` then holds the stated lengths (len(fabricated) / len(motif) differs for fractional ones)
    std::vector<std::string> owned;
    std::vector<double> lengths;
};

// utils.FabricateAllele(motif, length): floor(length) copies, then leading bases of the motif while one more base
// stays below `length` copies
inline std::string fabricate_allele(const std::string& motif, double length) {
    std::string fab;
    const double fl = std::floor(length);
    if (fl > 0)
        for (long k = 0; k < (long)fl; ++k) fab += motif;
    size_t i = 0;
    while (((double)(fab.size() + 1)) / (double)motif.size() < length && i < motif.size()) fab += motif[i++];
    return fab;
}

// What _HarmonizeHipSTRRecord / _HarmonizeGangSTRRecord / _HarmonizeAdVNTRRecord + TRRecord.__init__ derive per record
// (reference tr_harmonizer.py:303-408, 693-773): the trimmed alleles and the motif LENGTH (lengths are
// len(allele) / len(motif); the motif's letters do not enter any statistic).  Anything unusual -- a missing mandatory
// INFO field, a symbolic allele, a value that is not an integer -- leaves ok = false: the caller runs that batch
// through the Python harmoniser, which raises the reference's errors.
void harmonize_one(const char* line, const int32_t* fo, int64_t line_len, int vcftype, HzRecord& r) {
    const char* col[9];
    const char* cole[9];
    for (int k = 0; k < 8; ++k) {
        col[k] = line + fo[k];
        cole[k] = line + fo[k + 1] - 1;
    }
    if (fo[8] == 0 && fo[9] == 0) cole[7] = line + line_len;   // no FORMAT column
    if (fo[8] != 0) cole[7] = line + fo[8] - 1;
    r.ok = false;
    long pos;
    if (!parse_long(col[1], cole[1], pos)) return;
    r.pos = pos;
    r.tr_pos = pos;
    const char* ref = col[3];
    const long ref_len = (long)(cole[3] - col[3]);
    const char* flt = col[6];
    const long fl = (long)(cole[6] - col[6]);
    r.passing = (fl == 1 && flt[0] == '.') || (fl == 4 && memcmp(flt, "PASS", 4) == 0);
    {   // utils.GetHomopolymerRun(REF): longest run of one letter, case-insensitive (dumpSTR.py:1304-1306)
        int best = ref_len > 0 ? 1 : 0, run = 1;
        for (long i = 1; i < ref_len; ++i) {
            run = ((ref[i] | 32) == (ref[i - 1] | 32)) ? run + 1 : 1;
            if (run > best) best = run;
        }
        r.hrun = best;
    }
    // alleles: REF then ALT (comma separated, '.' = none)
    std::vector<std::pair<const char*, long>> raw;
    raw.emplace_back(ref, ref_len);
    const char* a = col[4];
    const char* ae = cole[4];
    if (!(ae - a == 1 && a[0] == '.')) {
        while (a <= ae) {
            const char* t = static_cast<const char*>(memchr(a, ',', (size_t)(ae - a)));
            if (!t) t = ae;
            raw.emplace_back(a, (long)(t - a));
            a = t + 1;
            if (t == ae) break;
        }
    }
    const bool by_length = vcftype == TRK_VT_EH || vcftype == TRK_VT_POPSTR;
    // (HipSTR / LongTR: the reference slices and upper-cases a symbolic '<DEL>' like any other allele string --
    // tr_harmonizer.py:336-408 -- so it is an allele here too; the other sequence callers leave such records to Python)
    if (!by_length && vcftype != TRK_VT_HIPSTR)
        for (auto& al : raw)
            for (long i = 0; i < al.second; ++i)
                if (al.first[i] == '<' || al.first[i] == '[' || al.first[i] == ']' || al.first[i] == '*') return;  // symbolic
    const char* info = col[7];
    const char* infoe = cole[7];
    const char *vb, *ve;
    bool hv;
    {   // INFO PERIOD where the record has one (the HRUN locus filter, filters.py:211-213)
        long pv;
        if (info_get(info, infoe, "PERIOD", 6, vb, ve, hv) && hv && parse_long(vb, ve, pv) && pv > INT32_MIN && pv <= INT32_MAX)
            r.period = (int32_t)pv;
    }
    if (by_length) {
        // _HarmonizeEHRecord / _HarmonizePopSTRRecord (tr_harmonizer.py:473-550): the alt alleles are '<STRn>' / '<n>',
        // n a float; EH's reference length is RL / len(RU), PopSTR's the REF string
        std::string motif;
        const char* prefix;
        size_t plen;
        if (vcftype == TRK_VT_EH) {
            const char *xb, *xe;
            if (!info_get(info, infoe, "VARID", 5, xb, xe, hv) || !hv) return;
            if (!info_get(info, infoe, "RU", 2, vb, ve, hv) || !hv || ve == vb) return;
            motif.assign(vb, (size_t)(ve - vb));
            prefix = "<STR";
            plen = 4;
        } else {
            if (!info_get(info, infoe, "Motif", 5, vb, ve, hv) || !hv || ve == vb) return;
            motif.assign(vb, (size_t)(ve - vb));
            prefix = "<";
            plen = 1;
        }
        for (auto& c : motif)
            if (c >= 'a' && c <= 'z') c = (char)(c - 32);
        r.unit = (double)motif.size();
        r.owned.clear();
        r.lengths.clear();
        if (vcftype == TRK_VT_EH) {
            long rl;
            if (!info_get(info, infoe, "RL", 2, vb, ve, hv) || !hv || !parse_long(vb, ve, rl) || rl < 0) return;
            const double ref_units = (double)rl / r.unit;
            r.owned.push_back(fabricate_allele(motif, ref_units));
            r.lengths.push_back(ref_units);
        } else {
            r.owned.emplace_back(ref, (size_t)ref_len);
            r.lengths.push_back((double)ref_len / r.unit);
        }
        for (size_t q = 1; q < raw.size(); ++q) {
            const char* ab = raw[q].first;
            const long al = raw[q].second;
            if (al < (long)plen + 2 || memcmp(ab, prefix, plen) != 0 || ab[al - 1] != '>') return;   // Python: TypeError
            std::string num(ab + plen, (size_t)(al - (long)plen - 1));
            char* endp = nullptr;
            const double nrep = strtod(num.c_str(), &endp);
            if (endp == num.c_str() || *endp != 0 || !(nrep >= 0.0) || !std::isfinite(nrep)) return;
            // (characters Python's float() takes and strtod does not, or the reverse: leave those to Python)
            for (char ch : num)
                if (!((ch >= '0' && ch <= '9') || ch == '.' || ch == 'e' || ch == 'E' || ch == '+' || ch == '-')) return;
            r.owned.push_back(fabricate_allele(motif, nrep));
            r.lengths.push_back(nrep);
        }
        r.alleles.clear();
        for (auto& o : r.owned) r.alleles.emplace_back(o.data(), (long)o.size());
        {   // HRUN is taken on trrecord.ref_allele (dumpSTR.py:1307-1312): for ExpansionHunter the FABRICATED reference
            const std::string& ra = r.owned[0];
            int best = ra.empty() ? 0 : 1, run = 1;
            for (size_t i = 1; i < ra.size(); ++i) {
                run = ((ra[i] | 32) == (ra[i - 1] | 32)) ? run + 1 : 1;
                if (run > best) best = run;
            }
            r.hrun = best;
        }
    } else if (vcftype == TRK_VT_HIPSTR) {
        long start, endv, period;
        if (!info_get(info, infoe, "START", 5, vb, ve, hv) || !hv || !parse_long(vb, ve, start)) return;
        if (!info_get(info, infoe, "END", 3, vb, ve, hv) || !hv || !parse_long(vb, ve, endv)) return;
        if (!info_get(info, infoe, "PERIOD", 6, vb, ve, hv) || !hv || !parse_long(vb, ve, period)) return;
        if (period <= 0) return;
        const long lead = start - pos;
        const long tail = endv - pos + 1 - ref_len;
        r.alleles.clear();
        for (auto& al : raw) {
            long b, e;
            py_slice(al.second, lead, tail != 0, tail, b, e);   // a[lead:stop], stop = None when tail == 0
            r.alleles.emplace_back(al.first + b, e - b);
        }
        r.unit = (double)period;
        r.tr_pos = start;
    } else {
        if (!info_get(info, infoe, "RU", 2, vb, ve, hv) || !hv || ve == vb) return;
        const long rul = (long)(ve - vb);
        const char *xb, *xe;
        const bool has_vid = info_get(info, infoe, "VID", 3, xb, xe, hv);
        const bool has_varid = info_get(info, infoe, "VARID", 5, xb, xe, hv);
        if (vcftype == TRK_VT_GANGSTR && (has_vid || has_varid)) return;
        if (vcftype == TRK_VT_ADVNTR && !has_vid) return;
        r.alleles = raw;
        r.unit = (double)rul;
    }
    r.end = pos + r.alleles[0].second;   // rec.POS + len(trrec.ref_allele), statSTR.py:590
    r.ok = true;
}

// Python's '{:.<p>}'.format(float): '%.<p>g' plus '.0' when the result looks like an integer
inline void put_py_float(std::string& o, double v, int prec) {
    char tmp[64];
    int n;
    // "%.*g": to_chars(general, precision) is printf's %g by definition, without the format parsing and the locale
    auto tr = std::isfinite(v) && prec > 0 && prec < 40 ? std::to_chars(tmp, tmp + sizeof tmp, v, std::chars_format::general, prec)
                                                       : std::to_chars_result{tmp, std::errc::invalid_argument};
    if (tr.ec == std::errc()) n = (int)(tr.ptr - tmp);
    else n = snprintf(tmp, sizeof tmp, "%.*g", prec, v);
    o.append(tmp, (size_t)n);
    bool plain = true;
    for (int i = 0; i < n; ++i)
        if (tmp[i] == '.' || tmp[i] == 'e' || tmp[i] == 'n' || tmp[i] == 'i') plain = false;   // nan / inf keep their text
    if (plain) o.append(".0");
}

// str(numpy.float64): shortest digits that round-trip, positional for 1e-4 <= |x| < 1e16, always with a '.'
inline void put_np_float(std::string& o, double v) {
    char tmp[64];
    const double av = std::fabs(v);
    if (v != v) { o.append("nan"); return; }
    if (std::isinf(v)) { o.append(v < 0 ? "-inf" : "inf"); return; }
    if (av != 0.0 && (av < 1e-4 || av >= 1e16)) {
        auto r = std::to_chars(tmp, tmp + sizeof tmp, v, std::chars_format::scientific);
        std::string s(tmp, r.ptr);           // d.ddde+XX: numpy prints at least two exponent digits, like to_chars
        o.append(s);
        return;
    }
    auto r = std::to_chars(tmp, tmp + sizeof tmp, v, std::chars_format::fixed);
    size_t n = (size_t)(r.ptr - tmp);
    o.append(tmp, n);
    if (!memchr(tmp, '.', n)) o.append(".0");
}

}  // namespace

extern "C" {

int trk_vcf_harmonize(trk_vcf* v, const trk_vcf_batch* b, int vcftype, trk_vcf_harmonized* out) {
    if (!v || !b || !out || vcftype < 0 || vcftype > TRK_VT_POPSTR) return 2;
    v->hz_i ^= 1;
    HzStore& st = v->hz2[v->hz_i];
    const int n = b->n_records;
    std::vector<HzRecord> recs((size_t)n);
    {
        std::atomic<int> next{0};
        auto runner = [&]() {
            for (;;) {
                const int i0 = next.fetch_add(64);
                if (i0 >= n) break;
                for (int i = i0; i < std::min(n, i0 + 64); ++i)
                    harmonize_one(b->text + b->line_off[i], b->field_off + (size_t)i * 10, b->line_end[i] - b->line_off[i],
                                  vcftype, recs[(size_t)i]);
            }
        };
        const int nt = std::max(1, std::min({v->n_threads, 16, (n + 63) / 64}));
        run_on_caller_pool(nt, runner);
    }
    st.allele_off.assign((size_t)n + 1, 0);
    st.pos.assign((size_t)n, 0);
    st.tr_pos.assign((size_t)n, 0);
    st.end.assign((size_t)n, 0);
    st.passing.assign((size_t)n, 0);
    st.status.assign((size_t)n, 0);
    st.n_str_classes.assign((size_t)n, 0);
    st.n_len_classes.assign((size_t)n, 0);
    st.hrun.assign((size_t)n, 0);
    st.period.assign((size_t)n, INT32_MIN);
    int n_python = 0;
    for (int i = 0; i < n; ++i) {
        const HzRecord& r = recs[(size_t)i];
        st.status[(size_t)i] = r.ok ? 0 : 1;
        n_python += !r.ok;
        st.pos[(size_t)i] = r.pos;
        st.tr_pos[(size_t)i] = r.tr_pos;
        st.end[(size_t)i] = r.end;
        st.passing[(size_t)i] = r.passing;
        st.hrun[(size_t)i] = r.hrun;
        st.period[(size_t)i] = r.period;
        const size_t A = r.ok ? r.alleles.size() : 1;
        if (A > 65535) { st.status[(size_t)i] = 1; ++n_python; }
        st.allele_off[(size_t)i + 1] = st.allele_off[(size_t)i] + (int32_t)(A > 65535 ? 1 : A);
    }
    const size_t sumA = (size_t)st.allele_off[(size_t)n];
    st.len_class.assign(sumA, 0);
    st.str_class.assign(sumA, 0);
    st.len_class_value.assign(sumA, 0.0);
    st.allele_len.assign(sumA, 0.0);
    st.key_off.assign(sumA + 1, 0);
    st.keys.clear();
    // The classes of a record's alleles are its own business: chunks of 64 records by the same threads, every chunk with
    // the sorted distinct sequences of its records in a string of its own (key_off relative to it); the strings are
    // joined in order afterwards.  (This loop was serial: 3 of the call's 5 ms on a batch of 3355 HipSTR records.)
    constexpr int HZ_CHUNK = 64;
    const int n_chunks = (n + HZ_CHUNK - 1) / HZ_CHUNK;
    std::vector<std::string> chunk_keys((size_t)n_chunks);
    {
        std::atomic<int> next{0};
        auto runner = [&]() {
            std::vector<int> order;
            std::vector<std::string> up;
            for (;;) {
                const int c = next.fetch_add(1);
                if (c >= n_chunks) break;
                std::string& keys = chunk_keys[(size_t)c];
                for (int i = c * HZ_CHUNK; i < std::min(n, (c + 1) * HZ_CHUNK); ++i) {
                    const size_t o = (size_t)st.allele_off[(size_t)i];
                    const size_t A = (size_t)st.allele_off[(size_t)i + 1] - o;
                    if (st.status[(size_t)i]) {
                        for (size_t q = 0; q < A; ++q) st.key_off[o + q] = (int64_t)keys.size();
                        continue;
                    }
                    const HzRecord& r = recs[(size_t)i];
                    up.resize(A);
                    for (size_t q = 0; q < A; ++q) {
                        up[q].assign(r.alleles[q].first, (size_t)r.alleles[q].second);
                        for (auto& ch : up[q])
                            if (ch >= 'a' && ch <= 'z') ch = (char)(ch - 32);
                        st.allele_len[o + q] = r.lengths.empty() ? (double)r.alleles[q].second / r.unit   // len(allele) / len(motif)
                                                                 : r.lengths[q];                          // the stated length
                    }
                    // sequence classes: dense rank in sorted order of the distinct sequences (python str order == byte order)
                    order.resize(A);
                    for (size_t q = 0; q < A; ++q) order[q] = (int)q;
                    std::sort(order.begin(), order.end(), [&](int x, int y) { return up[(size_t)x] < up[(size_t)y]; });
                    int rank = -1;
                    for (size_t k = 0; k < A; ++k) {
                        if (k == 0 || up[(size_t)order[k]] != up[(size_t)order[k - 1]]) {
                            ++rank;
                            st.key_off[o + (size_t)rank] = (int64_t)keys.size();
                            keys.append(up[(size_t)order[k]]);
                        }
                        st.str_class[o + (size_t)order[k]] = (uint16_t)rank;
                    }
                    st.n_str_classes[(size_t)i] = rank + 1;
                    for (size_t q = (size_t)rank + 1; q < A; ++q) st.key_off[o + q] = (int64_t)keys.size();
                    // length classes: dense rank of the distinct lengths, ascending
                    std::sort(order.begin(), order.end(),
                              [&](int x, int y) { return st.allele_len[o + (size_t)x] < st.allele_len[o + (size_t)y]; });
                    rank = -1;
                    for (size_t k = 0; k < A; ++k) {
                        const double lv = st.allele_len[o + (size_t)order[k]];
                        if (k == 0 || lv != st.allele_len[o + (size_t)order[k - 1]]) {
                            ++rank;
                            st.len_class_value[o + (size_t)rank] = lv;
                        }
                        st.len_class[o + (size_t)order[k]] = (uint16_t)rank;
                    }
                    st.n_len_classes[(size_t)i] = rank + 1;
                }
            }
        };
        const int nt = std::max(1, std::min({v->n_threads, 16, n_chunks}));
        run_on_caller_pool(nt, runner);
    }
    {   // join the chunks' strings; the offsets become absolute
        size_t total = 0;
        for (const auto& k : chunk_keys) total += k.size();
        st.keys.reserve(total);
        for (int c = 0; c < n_chunks; ++c) {
            const int64_t base = (int64_t)st.keys.size();
            const size_t o0 = (size_t)st.allele_off[(size_t)c * HZ_CHUNK];
            const size_t o1 = (size_t)st.allele_off[(size_t)std::min(n, (c + 1) * HZ_CHUNK)];
            if (base)
                for (size_t q = o0; q < o1; ++q) st.key_off[q] += base;
            st.keys.append(chunk_keys[(size_t)c]);
        }
    }
    st.key_off[sumA] = (int64_t)st.keys.size();
    out->n_records = n;
    out->n_python = n_python;
    out->n_alleles_total = (int64_t)sumA;
    out->allele_off = st.allele_off.data();
    out->len_class = st.len_class.data();
    out->str_class = st.str_class.data();
    out->len_class_value = st.len_class_value.data();
    out->allele_len = st.allele_len.data();
    out->pos = st.pos.data();
    out->end = st.end.data();
    out->passing = st.passing.data();
    out->status = st.status.data();
    out->keys = st.keys.data();
    out->key_off = st.key_off.data();
    out->n_str_classes = st.n_str_classes.data();
    out->n_len_classes = st.n_len_classes.data();
    out->hrun = st.hrun.data();
    out->period = st.period.data();
    out->tr_pos = st.tr_pos.data();
    return 0;
}

int64_t trk_vcf_statstr_rows(const trk_vcf_batch* b, const trk_vcf_harmonized* h, const trk_vcf_statstr* in,
                             const uint8_t* skip, char* out, int64_t cap, int32_t* err_locus, int32_t* err_kind) {
    if (!b || !h || !in || !in->locus_int || !in->locus_f64 || !in->allele_count) return INT64_MIN;
    const int n = h->n_records, G = in->n_groups;
    const int64_t sumA = h->n_alleles_total;
    if (err_locus) *err_locus = -1;
    if (err_kind) *err_kind = 0;
    // columns of trk_stats_out (include/trk.h)
    enum { LI_N_CALLED = 0, LI_N_BAD = 5, LI_HWE_LEN = 6, LI_HWE_STR = 7, LI_NALL_LEN = 9, LI_NALL_STR = 10, LI_COLS = 12 };
    enum { LF_THRESH = 0, LF_MEAN = 1, LF_MODE = 2, LF_VAR = 3, LF_HET_LEN = 4, LF_HET_STR = 5, LF_ENT_LEN = 6,
           LF_ENT_STR = 7, LF_HWEP_LEN = 8, LF_HWEP_STR = 9, LF_COLS = 12 };
    const bool ul = in->use_length != 0;
    const int prec = in->precision;
    // rows are independent: formatted by a few threads into per-chunk strings, concatenated in order
    const int chunk = 64;
    const int n_chunks = (n + chunk - 1) / chunk;
    std::vector<std::string> parts((size_t)n_chunks);
    std::atomic<int> next{0};
    std::atomic<int> bad_locus{INT32_MAX};
    std::atomic<int> bad_kind{0};
    auto fmt_float = [&](std::string& o, double v) {
        o.push_back('\t');
        if (v != v) o.append("nan");
        else put_py_float(o, v, prec);
    };
    auto runner = [&]() {
        std::vector<int64_t> cc;
        char tmp[64];
        for (;;) {
            const int c = next.fetch_add(1);
            if (c >= n_chunks) break;
            std::string& o = parts[(size_t)c];
            o.reserve((size_t)chunk * 160);
            for (int l = c * chunk; l < std::min(n, (c + 1) * chunk); ++l) {
                if (skip && skip[l]) continue;
                const int64_t off = h->allele_off[l], A = h->allele_off[l + 1] - off;
                const char* line = b->text + b->line_off[l];
                const int32_t* fo = b->field_off + (size_t)l * 10;
                o.append(line + fo[0], (size_t)(fo[1] - 1 - fo[0]));   // str(rec.CHROM)
                o.push_back('\t');
                auto r = std::to_chars(tmp, tmp + sizeof tmp, (long long)h->pos[l]);
                o.append(tmp, (size_t)(r.ptr - tmp));
                o.push_back('\t');
                r = std::to_chars(tmp, tmp + sizeof tmp, (long long)h->end[l]);
                o.append(tmp, (size_t)(r.ptr - tmp));
                auto I = [&](int g, int col) { return in->locus_int[((int64_t)g * n + l) * LI_COLS + col]; };
                auto F = [&](int g, int col) { return in->locus_f64[((int64_t)g * n + l) * LF_COLS + col]; };
                for (int g = 0; g < G; ++g)
                    if (I(g, LI_N_BAD)) {
                        int cur = bad_locus.load();
                        while (l < cur && !bad_locus.compare_exchange_weak(cur, l)) {}
                        if (l <= bad_locus.load()) bad_kind = 3;
                    }
                auto afreq = [&](int g, bool count) {
                    // statSTR.py:158-172: 'key:value' of every class with a non-zero count, in sorted key order
                    const int ncls = ul ? h->n_len_classes[l] : h->n_str_classes[l];
                    const uint16_t* cls = (ul ? h->len_class : h->str_class) + off;
                    cc.assign((size_t)ncls, 0);
                    const int32_t* cnt = in->allele_count + (int64_t)g * sumA + off;
                    int64_t total = 0;
                    for (int64_t q = 0; q < A; ++q) {
                        cc[cls[q]] += cnt[q];
                        total += cnt[q];
                    }
                    o.push_back('\t');
                    bool any = false;
                    for (int k = 0; k < ncls; ++k) {
                        if (!cc[(size_t)k]) continue;
                        if (any) o.push_back(',');
                        any = true;
                        if (ul) put_np_float(o, h->len_class_value[off + k]);
                        else o.append(h->keys + h->key_off[off + k], (size_t)(h->key_off[off + k + 1] - h->key_off[off + k]));
                        o.push_back(':');
                        // ('%.3f' and '%d' of the reference's format strings: to_chars(fixed, 3) is printf's %.3f)
                        std::to_chars_result m = count ? std::to_chars(tmp, tmp + sizeof tmp, (long long)cc[(size_t)k])
                                                       : std::to_chars(tmp, tmp + sizeof tmp, (double)cc[(size_t)k] / (double)total,
                                                                       std::chars_format::fixed, 3);
                        o.append(tmp, (size_t)(m.ptr - tmp));
                    }
                    if (!any) o.push_back('.');
                };
                if (in->flags & TRK_SS_THRESH) for (int g = 0; g < G; ++g) fmt_float(o, F(g, LF_THRESH));
                if (in->flags & TRK_SS_AFREQ) for (int g = 0; g < G; ++g) afreq(g, false);
                if (in->flags & TRK_SS_ACOUNT) for (int g = 0; g < G; ++g) afreq(g, true);
                if (in->flags & TRK_SS_NALLELES)
                    for (int g = 0; g < G; ++g) {
                        o.push_back('\t');
                        r = std::to_chars(tmp, tmp + sizeof tmp, (int)I(g, ul ? LI_NALL_LEN : LI_NALL_STR));
                        o.append(tmp, (size_t)(r.ptr - tmp));
                    }
                if (in->flags & TRK_SS_HWEP)
                    for (int g = 0; g < G; ++g) {
                        const int stt = I(g, ul ? LI_HWE_LEN : LI_HWE_STR);
                        if (stt == 2 || stt == 3) {   // the reference raises here (ValueError / IndexError)
                            int cur = bad_locus.load();
                            while (l < cur && !bad_locus.compare_exchange_weak(cur, l)) {}
                            if (l <= bad_locus.load()) bad_kind = stt == 2 ? 1 : 2;
                        }
                        fmt_float(o, F(g, ul ? LF_HWEP_LEN : LF_HWEP_STR));
                    }
                if (in->flags & TRK_SS_HET) for (int g = 0; g < G; ++g) fmt_float(o, F(g, ul ? LF_HET_LEN : LF_HET_STR));
                if (in->flags & TRK_SS_ENTROPY) for (int g = 0; g < G; ++g) fmt_float(o, F(g, ul ? LF_ENT_LEN : LF_ENT_STR));
                if (in->flags & TRK_SS_MEAN) for (int g = 0; g < G; ++g) fmt_float(o, F(g, LF_MEAN));
                if (in->flags & TRK_SS_MODE) for (int g = 0; g < G; ++g) fmt_float(o, F(g, LF_MODE));
                if (in->flags & TRK_SS_VAR) for (int g = 0; g < G; ++g) fmt_float(o, F(g, LF_VAR));
                if (in->flags & TRK_SS_NUMCALLED)
                    for (int g = 0; g < G; ++g) {
                        o.push_back('\t');
                        r = std::to_chars(tmp, tmp + sizeof tmp, (int)I(g, LI_N_CALLED));
                        o.append(tmp, (size_t)(r.ptr - tmp));
                    }
                o.push_back('\n');
            }
        }
    };
    // (a row is ~25 us of number formatting: 17 000 rows on eight threads were 67 ms of statSTR's 0.34 s per GB of text)
    const int nt = std::max(1, std::min(default_threads(32), n_chunks));
    run_on_caller_pool(nt, runner);
    if (bad_locus.load() != INT32_MAX) {
        if (err_locus) *err_locus = bad_locus.load();
        if (err_kind) *err_kind = bad_kind.load();
    }
    int64_t total = 0;
    for (auto& p : parts) total += (int64_t)p.size();
    if (total > cap || !out) return -total;
    int64_t w = 0;
    for (auto& p : parts) {
        memcpy(out + w, p.data(), p.size());
        w += (int64_t)p.size();
    }
    return total;
}

}  // extern "C"

// =====================================================================================================
// dumpSTR records of a whole batch: what the reference does per record in ApplyCallFilters (dumpSTR.py:613-774: the
// FORMAT/FILTER column, genotypes and every other FORMAT field of filtered calls nulled) and cyvcf2.Writer.write_record
// (dumpSTR.py:1338) -- decode every FORMAT field of the record to its typed array, null the filtered samples,
// serialise -- for all records of a batch on host threads, one record per task.  The caller supplies the nine leading
// columns of every output line (FILTER and INFO rewritten, FORMAT with ':FILTER' appended).
// =====================================================================================================
extern "C" {

}  // extern "C"

namespace {

// ---- dumpSTR's record heads, natively (round 4) ---------------------------------------------------------------
// What trtools_amd/dumpSTR/dumpSTR.py builds per record in Python: the eight leading columns with FILTER replaced
// and INFO rewritten (vcfio.rewrite_info / _rewrite_info_general: every value re-serialised the way htslib writes it
// back -- Integer from the parsed int, Float from float32 by '%g', a Flag as its bare key, String as it is), the
// updates HRUN / HET / HWEP / AC / REFAC in place or appended in that order (dumpSTR.py:1304-1336), FORMAT + ':FILTER'.
// Returns false where the Python code must build the head (a key twice, a number Python's int() / float() takes and
// this parser does not).
inline bool info_int_token(const char* p, const char* e, std::string& o) {
    // Python's int(): here only [+-]?digits (<= 18); anything else is left to Python
    if (e - p == 1 && *p == '.') { o.push_back('.'); return true; }
    const char* q = p;
    bool neg = false;
    if (q < e && (*q == '+' || *q == '-')) { neg = *q == '-'; ++q; }
    if (q == e || e - q > 18) return false;
    long long v = 0;
    for (const char* c = q; c < e; ++c) {
        if (*c < '0' || *c > '9') return false;
        v = v * 10 + (*c - '0');
    }
    char tmp[24];
    auto r = std::to_chars(tmp, tmp + sizeof tmp, neg ? -v : v);
    o.append(tmp, (size_t)(r.ptr - tmp));
    return true;
}
inline bool info_float_token(const char* p, const char* e, std::string& o) {
    if (e - p == 1 && *p == '.') { o.push_back('.'); return true; }
    if (p == e) return false;
    for (const char* c = p; c < e; ++c)
        if (!((*c >= '0' && *c <= '9') || *c == '.' || *c == 'e' || *c == 'E' || *c == '+' || *c == '-')) return false;
    double d = 0;
    const char* q = (*p == '+') ? p + 1 : p;      // from_chars takes no leading '+', Python's float() does
    auto r = std::from_chars(q, e, d);
    if (r.ptr != e) return false;
    if (r.ec == std::errc::result_out_of_range) d = strtod(std::string(p, e).c_str(), nullptr);
    else if (r.ec != std::errc()) return false;
    const float f = (float)d;                      // float(np.float32(x))
    if (!std::isfinite(f)) return false;
    char tmp[48];
    const int len = snprintf(tmp, sizeof tmp, "%g", (double)f);
    o.append(tmp, (size_t)len);
    return true;
}
struct InfoUpdates {
    int32_t hrun;
    bool have_stats;
    double het, hwep;
    const int32_t* ac;     // counts by allele index: ac[0] the reference's
    int n_alt;
};
inline void put_update(std::string& o, int which, const InfoUpdates& u) {
    char tmp[48];
    switch (which) {
        case 0: { auto r = std::to_chars(tmp, tmp + sizeof tmp, u.hrun); o.append("HRUN=").append(tmp, (size_t)(r.ptr - tmp)); break; }
        case 1:
        case 2: {
            o.append(which == 1 ? "HET=" : "HWEP=");
            if (!u.have_stats) { o.append("-1"); break; }
            const int len = snprintf(tmp, sizeof tmp, "%g", which == 1 ? u.het : u.hwep);
            o.append(tmp, (size_t)len);
            break;
        }
        case 3: {
            o.append("AC=");
            if (u.n_alt == 0) { o.push_back('0'); break; }
            for (int j = 1; j <= u.n_alt; ++j) {
                if (j > 1) o.push_back(',');
                auto r = std::to_chars(tmp, tmp + sizeof tmp, u.have_stats ? u.ac[j] : 0);
                o.append(tmp, (size_t)(r.ptr - tmp));
            }
            break;
        }
        default: {
            auto r = std::to_chars(tmp, tmp + sizeof tmp, u.have_stats ? u.ac[0] : 0);
            o.append("REFAC=").append(tmp, (size_t)(r.ptr - tmp));
        }
    }
}
bool rewrite_info_native(const char* p, const char* e, const trk_vcf_dumpstr2* x, const InfoUpdates& u, std::string& o) {
    static const char* const UPD[5] = {"HRUN", "HET", "HWEP", "AC", "REFAC"};
    static const size_t UPDL[5] = {4, 3, 4, 2, 5};
    bool done[5] = {false, false, false, false, false};
    const size_t o0 = o.size();
    bool first = true;
    // keys seen so far (a key twice: Python's dict keeps the last value for both places)
    const char* seen_b[64];
    size_t seen_l[64];
    int n_seen = 0;
    if (!(e - p == 1 && *p == '.')) {
        while (p < e) {
            const char* te = find_ch(p, e, ';');
            if (te > p) {
                const char* eq = find_ch(p, te, '=');
                const size_t kl = (size_t)(eq - p);
                for (int i = 0; i < n_seen; ++i)
                    if (seen_l[i] == kl && memcmp(seen_b[i], p, kl) == 0) return false;
                if (n_seen == 64) return false;
                seen_b[n_seen] = p;
                seen_l[n_seen++] = kl;
                if (!first) o.push_back(';');
                first = false;
                int up = -1;
                for (int i = 0; i < 5; ++i)
                    if (UPDL[i] == kl && memcmp(UPD[i], p, kl) == 0) up = i;
                if (up >= 0) {
                    put_update(o, up, u);
                    done[up] = true;
                } else if (eq == te) {
                    o.append(p, kl);                 // a bare key, whatever its declared type
                } else {
                    int kind = 0;                    // String when the header does not declare the key
                    for (int t = 0; t < x->n_info_keys; ++t)
                        if (strlen(x->info_keys[t]) == kl && memcmp(x->info_keys[t], p, kl) == 0) { kind = x->info_kinds[t]; break; }
                    if (kind == 3) {
                        o.append(p, kl);             // a Flag with a value is written back as the flag
                    } else if (kind == 1 || kind == 2) {
                        o.append(p, kl + 1);
                        const char* v = eq + 1;
                        while (true) {
                            const char* ve = find_ch(v, te, ',');
                            if (!(kind == 1 ? info_int_token(v, ve, o) : info_float_token(v, ve, o))) return false;
                            if (ve == te) break;
                            o.push_back(',');
                            v = ve + 1;
                        }
                    } else {
                        o.append(p, (size_t)(te - p));
                    }
                }
            }
            if (te == e) break;
            p = te + 1;
        }
    }
    for (int i = 0; i < 5; ++i)
        if (!done[i]) {
            if (!first) o.push_back(';');
            first = false;
            put_update(o, i, u);
        }
    if (o.size() == o0) o.push_back('.');
    return true;
}

// ---- the sample columns of an output record without decoding them (round 4) ----------------------------------------
// trk_vcf_decode_formats -> nulling -> format_range turns every FORMAT value into a typed array element and back into
// text (44 ns per value on 32 threads, 0.33 s per GB).  Almost every value comes back as the bytes it was read from:
// an integer without leading zeros or sign, a decimal of at most six significant digits without trailing zeros
// between 1e-4 and 1e6 (float32 holds it, '%g' prints it back), a string, a '.'.  This pass walks the sample columns
// once, copies such tokens as they are, re-serialises the others one by one (the same parse and '%g' the decoder and
// the formatter use), writes the filtered calls as the reference nulls them (dumpSTR.py:721-746: every allele '.',
// unphased, every other field '.') and appends the FILTER value.  It declines (returns false: the record takes the
// decode / format path) whatever it cannot prove equal to that path's output: mixed phase separators, non-ASCII
// strings, vector fields whose length differs between samples, numbers from_chars does not take whole.
inline bool canon_uint(const char* p, const char* e) {      // digits without a leading zero, at most nine
    const ptrdiff_t n = e - p;
    if (n < 1 || n > 9) return false;
    if (*p == '0') return n == 1;
    for (; p < e; ++p)
        if (*p < '0' || *p > '9') return false;
    return true;
}
inline bool canon_int(const char* p, const char* e) {
    if (p < e && *p == '-') return e - p > 1 && p[1] != '0' && canon_uint(p + 1, e);
    return canon_uint(p, e);
}
// a decimal '%g' of its float32 value prints back unchanged: -?(0|[1-9]d*)(.d*[1-9])?, <= 6 significant digits,
// 1e-4 <= |x| < 1e6 or an integer below 1e6 (0 included)
inline bool canon_float(const char* p, const char* e) {
    if (p < e && *p == '-') ++p;
    if (p == e) return false;
    const char* ip = p;
    while (p < e && *p >= '0' && *p <= '9') ++p;
    const ptrdiff_t ni = p - ip;
    if (ni < 1 || (ni > 1 && *ip == '0') || ni > 6) return false;
    if (p == e) return true;                                // an integer of at most six digits
    if (*p != '.') return false;
    const char* fp = ++p;
    while (p < e && *p >= '0' && *p <= '9') ++p;
    const ptrdiff_t nf = p - fp;
    if (p != e || nf < 1 || e[-1] == '0') return false;
    if (*ip != '0') return ni + nf <= 6;
    // 0.000ddd: significant digits start at the first non-zero; at most three zeros behind the point (>= 1e-4)
    const char* z = fp;
    while (z < e && *z == '0') ++z;
    return (z - fp) <= 3 && (e - z) <= 6;
}
// "%g" of x: small integers by hand, other finite values through to_chars (general, 6 = printf's %g by definition,
// without the format parsing and the locale of snprintf), non-finite values through snprintf itself
inline int fmt_g6(char* tmp, size_t cap, double x) {
    if (x == (double)(int32_t)x && x > -1e6 && x < 1e6 && !(x == 0.0 && std::signbit(x))) {
        auto r = std::to_chars(tmp, tmp + cap, (int32_t)x);
        return (int)(r.ptr - tmp);
    }
    if (std::isfinite(x)) {
        auto r = std::to_chars(tmp, tmp + cap, x, std::chars_format::general, 6);
        if (r.ec == std::errc()) return (int)(r.ptr - tmp);
    }
    return snprintf(tmp, cap, "%g", x);
}
inline void put_callfilter(OutBuf& o, uint32_t m, int n_filters, const char* const* names, const double* const* values,
                           int64_t s) {
    char tmp[48];
    if (m & 0x80000000u) {
        o.put("NOCALL", 6);
    } else if (m == 0) {
        o.put("PASS", 4);
    } else {
        int n = 0;
        for (int b = 0; b < n_filters && b < 31; ++b) {
            if (!((m >> b) & 1u)) continue;
            if (n++) o.put(',');
            o.put(names[b], strlen(names[b]));
            o.put('_');
            const double x = values[b] ? values[b][s] : NAN;
            const int len = fmt_g6(tmp, sizeof tmp, x);
            o.put(tmp, (size_t)len);
        }
        if (!n) o.put('.');
    }
}
// The span transducer's first tier (round 4): records whose FORMAT fields are the genotype, scalar Integer / Float
// fields and strings, every sample's token followed by a tab (the last one by the line's newline).  A call that is kept is checked
// in ONE forward scan of its bytes (the canonical forms of canon_uint / canon_int / canon_float, inlined; no memchr
// per token and per field, no bounds-checked put per piece) and copied whole; a filtered call is one prepared string.
// Anything else about a record -- vectors, a token with more fields than keys, a number that is not canonical -- returns
// false before anything is kept and fast_samples below takes the record.  `end` must point at a readable byte (the
// newline).  53 -> ~20 ns per call on the 1 GB probe (profiles/r04_notes.md section 13).
bool fast_samples_scalar(const char* smp, const char* end, int S, int pl, int nf, const int* kinds, const uint32_t* m32,
                         const uint8_t* filtered, int n_filters, const char* const* names, const double* const* values,
                         OutBuf& o) {
    if (nf < 1 || nf > 16 || pl < 1) return false;
    for (int f = 0; f < nf; ++f)
        if (kinds[f] != -1 && kinds[f] != TRK_VCF_COL_INT && kinds[f] != TRK_VCF_COL_FLOAT && kinds[f] != TRK_VCF_COL_UCS4)
            return false;
    // a filtered call: alleles missing and unphased, every other field missing
    char nulltok[64];
    int nl = 0;
    for (int f = 0; f < nf; ++f) {
        if (f) nulltok[nl++] = ':';
        if (kinds[f] < 0) {
            if (nl + 2 * pl + 2 > (int)sizeof nulltok) return false;
            for (int j = 0; j < pl; ++j) {
                if (j) nulltok[nl++] = '/';
                nulltok[nl++] = '.';
            }
        } else {
            nulltok[nl++] = '.';
        }
    }
    char* const w0 = o.p + o.n;
    char* w = w0;
    const char* c = smp;
    char tmp[48];
#define TRK_DIGIT(ch) ((unsigned)((ch) - '0') <= 9u)
    for (int s = 0; s < S; ++s) {
        const char* const tok = c;
        const bool flt = filtered[s] != 0;
        int pad = 0;            // fields a kept call does not hold (HipSTR writes '.' for a sample without a call): '.' each
        for (int f = 0; f < nf; ++f) {
            const int kind = kinds[f];
            if (flt) {
                // only the shape matters: a scalar (no comma), at most nf fields
                while (*c != ':' && *c != '\t' && *c != ',' && *c != '\n' && *c != '\r' && *c != 0) ++c;
                if (*c == ',') return false;
                if (c == tok && f == 0) return false;             // an empty token
            } else if (kind < 0) {
                int na = 0;
                char sep = 0;
                for (;;) {
                    if (*c == '.') {
                        ++c;
                    } else if (*c == '0') {
                        ++c;
                        if (TRK_DIGIT(*c)) return false;
                    } else if (TRK_DIGIT(*c)) {
                        const char* a = c;
                        do ++c; while (TRK_DIGIT(*c));
                        if (c - a > 9) return false;
                    } else {
                        return false;
                    }
                    ++na;
                    if (*c != '/' && *c != '|') break;
                    if (sep && *c != sep) return false;
                    sep = *c++;
                }
                if (na > pl) return false;
            } else if (kind == TRK_VCF_COL_INT) {
                if (*c == '.') {
                    ++c;
                } else {
                    if (*c == '-') {
                        ++c;
                        if (*c == '0') return false;
                    }
                    if (*c == '0') {
                        ++c;
                        if (TRK_DIGIT(*c)) return false;
                    } else if (TRK_DIGIT(*c)) {
                        const char* a = c;
                        do ++c; while (TRK_DIGIT(*c));
                        if (c - a > 9) return false;
                    } else {
                        return false;
                    }
                }
            } else if (kind == TRK_VCF_COL_UCS4) {   // a string: ASCII, not empty, copied as it stands
                const char* a = c;
                while (*c != ':' && *c != '\t' && *c != '\n' && *c != '\r') {
                    if ((unsigned char)*c >= 0x80 || *c == 0) return false;
                    ++c;
                }
                if (c == a) return false;
            } else {   // Float: '.', or -?(0|[1-9]d*)(.d*[1-9])? with at most six significant digits, 1e-4 <= |x| < 1e6
                if (*c == '.' && !TRK_DIGIT(c[1])) {
                    ++c;
                } else {
                    if (*c == '-') ++c;
                    const char* ip = c;
                    while (TRK_DIGIT(*c)) ++c;
                    const ptrdiff_t ni = c - ip;
                    if (ni < 1 || (ni > 1 && *ip == '0') || ni > 6) return false;
                    if (*c == '.') {
                        const char* fp = ++c;
                        while (TRK_DIGIT(*c)) ++c;
                        const ptrdiff_t nfr = c - fp;
                        if (nfr < 1 || c[-1] == '0') return false;
                        if (*ip != '0') {
                            if (ni + nfr > 6) return false;
                        } else {
                            const char* z = fp;
                            while (*z == '0') ++z;
                            if (z - fp > 3 || c - z > 6) return false;
                        }
                    }
                }
            }
            if (f + 1 < nf) {
                if (*c == ':') { ++c; continue; }
                if (*c == '\t' || c == end) {                     // fewer fields than keys
                    pad = flt ? 0 : nf - 1 - f;
                    break;
                }
                return false;
            }
        }
        // the token ends here: a tab, or the line's end for the last sample
        if (s + 1 < S) {
            if (*c != '\t') return false;
        } else if (c != end) {
            return false;
        }
        *w++ = '\t';
        const uint32_t m = m32[s];
        if (flt) {
            memcpy(w, nulltok, (size_t)nl);
            w += nl;
            *w++ = ':';
            int n = 0;
            for (int b = 0; b < n_filters && b < 31; ++b) {
                if (!((m >> b) & 1u)) continue;
                if (n++) *w++ = ',';
                const size_t ln = strlen(names[b]);
                memcpy(w, names[b], ln);
                w += ln;
                *w++ = '_';
                const int len = fmt_g6(tmp, sizeof tmp, values[b] ? values[b][s] : NAN);
                memcpy(w, tmp, (size_t)len);
                w += len;
            }
            if (!n) *w++ = '.';
        } else {
            const size_t tl = (size_t)(c - tok);
            memcpy(w, tok, tl);
            w += tl;
            for (int k = 0; k < pad; ++k) {
                *w++ = ':';
                *w++ = '.';
            }
            if (m & 0x80000000u) {
                memcpy(w, ":NOCALL", 7);
                w += 7;
            } else {
                memcpy(w, ":PASS", 5);
                w += 5;
            }
        }
        ++c;   // past the tab (or the newline: not read again)
    }
#undef TRK_DIGIT
    o.n += (int64_t)(w - w0);
    return true;
}

// kinds[f]: -1 GT, -2 the record's own FILTER field (replaced in place), TRK_VCF_COL_INT / _FLOAT / _UCS4.
// m32 / filtered: the record's mask row.
bool fast_samples(const char* smp, const char* end, int S, int pl, int nf, const int* kinds, const uint32_t* m32,
                  const uint8_t* filtered, int n_filters, const char* const* names, const double* const* values,
                  OutBuf& o) {
    while (end > smp && (end[-1] == '\n' || end[-1] == '\r')) --end;
    int fcount[64], imax[64], iused[64];
    if (nf > 64) return false;
    for (int f = 0; f < nf; ++f) {
        fcount[f] = 0;          // values per token of a float vector field (0: not seen yet)
        imax[f] = 1;            // integer field: the widest token so far = the columns of the decoder's array
        iused[f] = INT32_MAX;   // ... and the narrowest width a filtered call of the record was written with
    }
    const char* p = smp;
    char tmp[48];
    bool has_filter = false;
    for (int s = 0; s < S; ++s) {
        if (p > end) return false;
        const char* se = find_ch(p, end, '\t');
        o.put('\t');
        const bool flt = filtered[s] != 0;
        const char* q = p;
        bool exhausted = q == se && false;
        for (int f = 0; f < nf; ++f) {
            if (f) o.put(':');
            const char* tb = q;
            const char* te = se;
            if (exhausted) {
                tb = te = nullptr;
            } else {
                te = find_ch(q, se, ':');
                if (te == se) exhausted = true;
                else q = te + 1;
            }
            const int kind = kinds[f];
            if (kind == -2) {                              // the record's own FORMAT/FILTER field: replaced in place
                put_callfilter(o, m32[s], n_filters, names, values, s);
                has_filter = true;
                continue;
            }
            if (flt) {
                // the reference nulls the call: every allele missing and unphased, every other field missing; what the
                // token held only matters for the shape of a vector field (checked on the calls that are kept)
                if (kind < 0) {
                    for (int j = 0; j < pl; ++j) {
                        if (j) o.put('/');
                        o.put('.');
                    }
                    if (pl == 0) o.put('.');
                } else if (kind == TRK_VCF_COL_INT) {
                    // one '.' per column of the record's widest integer vector (the decoder's array width; this
                    // call's own token counts): written with the width known so far, checked at the end
                    int cnt = 1;
                    if (tb)
                        for (const char* c = tb; c < te; ++c) cnt += *c == ',';
                    if (cnt > imax[f]) imax[f] = cnt;
                    for (int j = 0; j < imax[f]; ++j) {
                        if (j) o.put(',');
                        o.put('.');
                    }
                    if (imax[f] < iused[f]) iused[f] = imax[f];
                } else {
                    o.put('.');
                }
                continue;
            }
            if (!tb || (te - tb == 1 && *tb == '.')) {     // missing (or a column the sample does not have)
                if (tb == nullptr && kind < 0) return false;
                o.put('.');
                continue;
            }
            if (te == tb) return false;                    // an empty token: the decoder's business
            if (kind < 0) {
                // alleles: '.' or canonical integers, one separator throughout, at most pl of them
                char sep = 0;
                int na = 0;
                const char* a = tb;
                while (true) {
                    const char* ae = a;
                    while (ae < te && *ae != '/' && *ae != '|') ++ae;
                    if (!((ae - a == 1 && *a == '.') || canon_uint(a, ae))) return false;
                    ++na;
                    if (ae == te) break;
                    if (sep && *ae != sep) return false;
                    sep = *ae;
                    a = ae + 1;
                }
                if (na > pl) return false;
                o.put(tb, (size_t)(te - tb));
            } else if (kind == TRK_VCF_COL_INT) {
                const char* v = tb;
                int nv = 0;
                while (true) {
                    const char* ve = find_ch(v, te, ',');
                    if (nv++) o.put(',');
                    if (ve - v == 1 && *v == '.') {
                        o.put('.');
                    } else if (canon_int(v, ve)) {
                        o.put(v, (size_t)(ve - v));
                    } else {
                        int32_t x;
                        auto r = std::from_chars(v, ve, x);
                        if (r.ec != std::errc() || r.ptr != ve || x == INT32_MIN || x == INT32_MIN + 1) return false;
                        o.put_int(x);
                    }
                    if (ve == te) break;
                    v = ve + 1;
                }
                if (nv > imax[f]) imax[f] = nv;
            } else if (kind == TRK_VCF_COL_FLOAT) {
                const char* v = tb;
                int nv = 0;
                bool all_missing = true;
                const int64_t mark = o.n;
                while (true) {
                    const char* ve = find_ch(v, te, ',');
                    if (nv++) o.put(',');
                    if (ve - v == 1 && *v == '.') {
                        o.put('.');
                    } else if (canon_float(v, ve)) {
                        o.put(v, (size_t)(ve - v));
                        all_missing = false;
                    } else {
                        double x;
                        auto r = std::from_chars(v, ve, x);
                        if (r.ec != std::errc() || r.ptr != ve) return false;
                        const float fl = (float)x;
                        if (std::isnan(fl)) {
                            o.put('.');
                        } else {
                            auto w = std::to_chars(tmp, tmp + sizeof tmp, (double)fl, std::chars_format::general, 6);
                            o.put(tmp, (size_t)(w.ptr - tmp));
                            all_missing = false;
                        }
                    }
                    if (ve == te) break;
                    v = ve + 1;
                }
                if (nv > 1) {
                    // a vector: every sample must carry the same number of values (a shorter one is padded with
                    // missing values by the decoder and printed with them); all missing prints one '.'
                    if (fcount[f] == 0) fcount[f] = nv;
                    else if (fcount[f] != nv) return false;
                    if (all_missing) { o.n = mark; o.put('.'); }
                } else if (fcount[f] > 1) {
                    return false;
                } else if (fcount[f] == 0) {
                    fcount[f] = 1;
                }
            } else {
                for (const char* c = tb; c < te; ++c)
                    if ((unsigned char)*c >= 0x80 || *c == 0) return false;
                o.put(tb, (size_t)(te - tb));
            }
        }
        if (!exhausted) return false;                       // more tokens than FORMAT keys
        if (!has_filter) {
            o.put(':');
            put_callfilter(o, m32[s], n_filters, names, values, s);
        }
        if (se == end) {
            if (s != S - 1) return false;
            p = end + 1;
        } else {
            p = se + 1;
        }
    }
    if (p <= end) return false;                             // more columns than samples
    // a filtered call written before the record's widest integer vector was met has too few '.': the decode path's job
    for (int f = 0; f < nf; ++f)
        if (iused[f] != INT32_MAX && iused[f] != imax[f]) return false;
    return true;
}

// output chunks of the record writer, kept between calls (at most 1 GB of them)
struct FmtChunk {
    char* p = nullptr;
    size_t cap = 0;
};
struct FmtChunkPool {
    std::mutex m;
    std::vector<FmtChunk> free_list;
    size_t bytes = 0;
    FmtChunk take(size_t need) {
        {
            std::lock_guard<std::mutex> g(m);
            for (size_t i = 0; i < free_list.size(); ++i)
                if (free_list[i].cap >= need) {
                    FmtChunk c = free_list[i];
                    free_list.erase(free_list.begin() + (long)i);
                    bytes -= c.cap;
                    return c;
                }
        }
        FmtChunk c;
        c.cap = std::max<size_t>(need, (size_t)8 << 20);
        c.p = static_cast<char*>(malloc(c.cap));
        if (!c.p) throw std::bad_alloc();
        return c;
    }
    void give(const FmtChunk& c) {
        std::lock_guard<std::mutex> g(m);
        if (bytes + c.cap <= ((size_t)1 << 30)) {
            free_list.push_back(c);
            bytes += c.cap;
        } else {
            free(c.p);
        }
    }
};
FmtChunkPool& fmt_chunks() {
    static FmtChunkPool* p = new FmtChunkPool;
    return *p;
}

// records written without decoding / through the decode path / with a caller-built head, since the process started
std::atomic<int64_t> g_fmt_fast{0}, g_fmt_slow{0}, g_fmt_py_heads{0}, g_fmt_scalar{0}, g_fmt_device{0};   // (scalar / device: of the fast ones, the scalar tier's / the device's)

int64_t dumpstr_impl(const trk_vcf_batch* b, const trk_vcf_dumpstr* in, const trk_vcf_dumpstr2* ext, char* out,
                     int64_t cap, int32_t* err_record) {
    if (!b || !in || (!in->heads && !ext) || !in->gt || !in->locus_ploidy || (!in->mask8 && !in->mask32)) return INT64_MIN;
    const int n = b->n_records, S = in->n_samples, P = in->ploidy;
    if (err_record) *err_record = -1;
    // a record is formatted straight into a chunk of its thread (head, sample columns, newline); rptr / rlen say where
    // it lies.  Chunks come from a process-wide pool and go back to it: no allocation, no page fault per batch
    std::vector<const char*> rptr((size_t)n, nullptr);
    std::vector<size_t> rlen((size_t)n, 0);
    // a record whose sample columns the device wrote: the head sits in a chunk (rptr / rlen), the columns stay where the
    // download put them (gptr / glen) until the gather copies both, and the newline, into the output block
    std::vector<const char*> gptr((size_t)n, nullptr);
    std::vector<size_t> glen((size_t)n, 0);
    std::vector<uint8_t> on_device((size_t)n, 0);     // the device holds the record's sample columns
    std::vector<FmtChunk> used_chunks;
    std::mutex used_mu;
    std::atomic<int> next{0};
    std::atomic<int> bad{INT32_MAX};
    std::atomic<int> need_heads{0};
    const bool fast_ok = ext && ext->fast_path && !(trk_opt("TRK_FMT_FAST") && atoi(trk_opt("TRK_FMT_FAST")) == 0);
    const bool scalar_tier = !(trk_opt("TRK_FMT_SCALAR") && atoi(trk_opt("TRK_FMT_SCALAR")) == 0);   // (0: the general transducer only)
    auto fail = [&](int l) {
        int cur = bad.load();
        while (l < cur && !bad.compare_exchange_weak(cur, l)) {}
    };
    auto runner = [&]() {
        std::vector<trk_vcf_decode> dec;
        std::vector<std::vector<char>> store;
        std::vector<trk_vcf_column> cols;
        std::vector<int16_t> gtrow;
        std::vector<uint32_t> m32((size_t)S);
        std::vector<uint8_t> filtered((size_t)S);
        std::vector<std::vector<double>> vals((size_t)in->n_filters);
        std::vector<const double*> vptr((size_t)std::max(in->n_filters, 1));
        std::vector<const char*> names((size_t)std::max(in->n_filters, 1));
        for (int k = 0; k < in->n_filters; ++k) names[(size_t)k] = in->filters[k].name;
        // the numbers behind '<filter name>_<value>' of the filters that fired somewhere in record l
        auto fire_values = [&](int l, uint32_t any_bits) {
        for (int k = 0; k < in->n_filters; ++k) {
            vptr[(size_t)k] = nullptr;
            if (!((any_bits >> k) & 1u)) continue;
            const trk_vcf_cf_value& fv = in->filters[k];
            std::vector<double>& v = vals[(size_t)k];
            v.resize((size_t)S);
            auto elem = [&](const void* plane, int dtype, int ncol, int col, int s) -> double {
                const size_t i = ((size_t)l * S + s) * ncol + col;
                return dtype == 1 ? (double)static_cast<const float*>(plane)[i] : (double)static_cast<const int32_t*>(plane)[i];
            };
            if (fv.kind == 2) {
                // sum of two columns of one plane: GangSTR's QEXP[1] + QEXP[2] (a float32 sum, as numpy adds two
                // float32 arrays) and RC[1] + RC[3] (integers)
                for (int s = 0; s < S; ++s) {
                    const size_t i = ((size_t)l * S + s) * fv.ncol_a;
                    if (fv.dtype_a == 1) {
                        const float* pa = static_cast<const float*>(fv.plane_a) + i;
                        const volatile float f = pa[fv.col_a] + pa[fv.col_a2];
                        v[(size_t)s] = (double)f;
                    } else {
                        const int32_t* pa = static_cast<const int32_t*>(fv.plane_a) + i;
                        v[(size_t)s] = (double)((int64_t)pa[fv.col_a] + (int64_t)pa[fv.col_a2]);
                    }
                }
            } else if (fv.kind == 3) {
                // GangSTR's bad confidence interval: the maximum-likelihood copy number (plane a, one column per
                // haplotype) of the FIRST haplotype whose interval (plane b: lo, hi per haplotype) excludes it
                for (int s = 0; s < S; ++s) {
                    const int32_t* ml = static_cast<const int32_t*>(fv.plane_a) + ((size_t)l * S + s) * fv.ncol_a;
                    const int32_t* ci = static_cast<const int32_t*>(fv.plane_b) + ((size_t)l * S + s) * fv.ncol_b;
                    double x = NAN;
                    for (int j = 0; j < fv.ncol_a && 2 * j + 1 < fv.ncol_b; ++j)
                        if (ml[j] < ci[2 * j] || ci[2 * j + 1] < ml[j]) {
                            x = (double)ml[j];
                            break;
                        }
                    v[(size_t)s] = x;
                }
            } else if (fv.kind == 4) {
                // PopSTR's require-support (filters.py:858-867): the read support (plane a: AD, one column per
                // allele) of the LAST haplotype whose allele has fewer than col_a reads
                for (int s = 0; s < S; ++s) {
                    const int32_t* ad = static_cast<const int32_t*>(fv.plane_a) + ((size_t)l * S + s) * fv.ncol_a;
                    const int16_t* g = in->gt + ((size_t)l * S + s) * in->ploidy;
                    double x = NAN;
                    for (int j = 0; j < in->ploidy; ++j) {
                        int a = g[j];
                        if (a < 0) a += fv.ncol_a;
                        if (a < 0 || a >= fv.ncol_a) continue;
                        if ((double)ad[a] < (double)fv.col_a) x = (double)ad[a];
                    }
                    v[(size_t)s] = x;
                }
            } else {
                for (int s = 0; s < S; ++s) {
                    const double a = elem(fv.plane_a, fv.dtype_a, fv.ncol_a, fv.col_a, s);
                    v[(size_t)s] = fv.kind == 1 ? a / elem(fv.plane_b, fv.dtype_b, fv.ncol_b, fv.col_b, s) : a;
                }
            }
            vptr[(size_t)k] = v.data();
        }
        };
        std::string head_store;
        std::vector<int> kinds;
        FmtChunk cur;
        size_t cur_n = 0;
        auto room = [&](size_t need) -> char* {
            if (!cur.p || cur_n + need > cur.cap) {
                cur = fmt_chunks().take(need);
                cur_n = 0;
                std::lock_guard<std::mutex> g(used_mu);
                used_chunks.push_back(cur);
            }
            return cur.p + cur_n;
        };
        for (;;) {
            const int l = next.fetch_add(1);
            if (l >= n) break;
            const char* line = b->text + b->line_off[l];
            const int32_t* fo = b->field_off + (size_t)l * 10;
            const int64_t line_len = b->line_end[l] - b->line_off[l];
            const char* head = in->heads ? in->heads[l] : nullptr;
            size_t hl = 0;
            if (ext) {
                if (ext->keep && !ext->keep[l]) continue;     // record dropped by the caller (--drop-filtered)
                if (!head) {
                    // the nine leading columns: 0-5 as read, FILTER, INFO rewritten, FORMAT + ':FILTER'
                    if (fo[9] <= fo[8] || fo[9] >= line_len) { fail(l); continue; }
                    head_store.clear();
                    head_store.append(line, (size_t)fo[6]);
                    const char* ft = ext->filter_text ? ext->filter_text[l] : nullptr;
                    if (ft) head_store.append(ft);
                    else head_store.append(line + fo[6], (size_t)(fo[7] - 1 - fo[6]));
                    head_store.push_back('\t');
                    InfoUpdates u;
                    u.hrun = ext->hrun[l];
                    u.have_stats = ext->have_stats[l] != 0;
                    u.het = ext->het[l];
                    u.hwep = ext->hwep[l];
                    u.ac = ext->allele_count + ext->allele_off[l];
                    u.n_alt = ext->allele_off[l + 1] - ext->allele_off[l] - 1;
                    if (!rewrite_info_native(line + fo[7], line + fo[8] - 1, ext, u, head_store)) {
                        if (ext->need_head) ext->need_head[l] = 1;
                        need_heads.fetch_add(1);
                        continue;
                    }
                    head_store.push_back('\t');
                    head_store.append(line + fo[8], (size_t)(fo[9] - 1 - fo[8]));
                    {   // FORMAT + ':FILTER' -- unless the record carries the key already (a second dumpSTR round)
                        bool has = false;
                        const char* fb = line + fo[8];
                        const char* fe2 = line + fo[9] - 1;
                        while (fb <= fe2) {
                            const char* c = find_ch(fb, fe2, ':');
                            if (c - fb == 6 && memcmp(fb, "FILTER", 6) == 0) has = true;
                            if (c >= fe2) break;
                            fb = c + 1;
                        }
                        if (!has) head_store.append(":FILTER");
                    }
                    head = head_store.data();
                    hl = head_store.size();
                } else {
                    hl = strlen(head);
                    g_fmt_py_heads.fetch_add(1, std::memory_order_relaxed);
                }
            } else {
                if (!head) continue;     // record dropped by the caller (--drop-filtered)
                hl = strlen(head);
            }
            if (fo[9] <= fo[8] || fo[9] >= line_len) { fail(l); continue; }
            // FORMAT keys of the record -> decode kinds from the header table
            const char* f = line + fo[8];
            const char* fe = line + fo[9] - 1;
            dec.clear();
            int gt_idx = -1, filter_idx = -1;
            bool ok = true;
            for (const char* a = f; a <= fe;) {
                const char* c = find_ch(a, fe, ':');
                const size_t kl = (size_t)(c - a);
                int kind = TRK_VCF_COL_UCS4;
                if (kl == 2 && a[0] == 'G' && a[1] == 'T') {
                    gt_idx = (int)dec.size();
                    kind = -1;
                } else if (kl == 6 && memcmp(a, "FILTER", 6) == 0) {
                    // a second dumpSTR round: the field's values are replaced where they stand (Variant.set_format of
                    // an existing key, dumpSTR.py:648-683), nothing is appended
                    if (filter_idx >= 0 || !ext) ok = false;
                    filter_idx = (int)dec.size();
                } else {
                    for (int t = 0; t < in->n_format_keys; ++t)
                        if (strlen(in->format_keys[t]) == kl && memcmp(in->format_keys[t], a, kl) == 0) {
                            kind = in->format_kinds[t];
                            break;
                        }
                }
                dec.push_back({kind, 0, nullptr});
                if (c >= fe) break;
                a = c + 1;
            }
            if (!ok || gt_idx < 0) { fail(l); continue; }
            const int nf = (int)dec.size();
            const char* smp = line + fo[9];
            const int64_t smp_len = line_len - fo[9];
            const int pl = in->locus_ploidy[l];
            if (fast_ok && (ext->dev_regions || ext->dev_emit) && ext->dev_flags && !ext->dev_flags[l] && filter_idx < 0) {
                // the device wrote this record's sample columns (trk_format_samples): head + those bytes + newline
                const size_t rl = ext->dev_region_len[l];
                char* dst = room(hl);
                memcpy(dst, head, hl);
                rptr[(size_t)l] = dst;
                rlen[(size_t)l] = hl;
                // (whole-record emit: the columns are put into `out` by dev_emit, gptr stays NULL)
                gptr[(size_t)l] = ext->dev_emit ? nullptr : ext->dev_regions + ext->dev_region_off[l];
                glen[(size_t)l] = rl + 1;        // (+ the newline, written by the gather)
                on_device[(size_t)l] = 1;
                cur_n += hl;
                g_fmt_fast.fetch_add(1, std::memory_order_relaxed);
                g_fmt_device.fetch_add(1, std::memory_order_relaxed);
                continue;
            }
            // the mask of this record as 32 bits; filtered = some filter fired on a called sample (dumpSTR.py:715-717)
            uint32_t any_bits = 0;
            for (int s = 0; s < S; ++s) {
                uint32_t m;
                if (in->mask8) {
                    const uint8_t x = in->mask8[(size_t)l * S + s];
                    m = (uint32_t)(x & 0x7f) | ((x & 0x80) ? 0x80000000u : 0u);
                } else {
                    m = in->mask32[(size_t)l * S + s];
                }
                m32[(size_t)s] = m;
                filtered[(size_t)s] = (m & 0x7fffffffu) != 0 && !(m & 0x80000000u);
                any_bits |= m & 0x7fffffffu;
            }
            fire_values(l, any_bits);
            if (fast_ok) {
                // the span transducer first (fast_samples): no typed arrays at all
                kinds.resize((size_t)nf);
                for (int i = 0; i < nf; ++i) kinds[(size_t)i] = i == filter_idx ? -2 : dec[(size_t)i].kind;
                int64_t cfw = 8;
                for (int k = 0; k < in->n_filters; ++k) cfw += (int64_t)strlen(names[(size_t)k]) + 26;
                const int64_t need = smp_len * 2 + (int64_t)S * (cfw + 4 * (pl + 1) + 2 * nf) + 64;
                char* dst = room(hl + (size_t)need + 1);
                OutBuf ob{dst + hl, need, 0};
                // (the scalar tier reads the byte behind the last token: text[line_end[l]] is the line's '\n' or '\r' in the
                // reader's buffer -- read_batch appends one to a last line without -- see trk_vcf.h)
                const char* send = smp + smp_len;
                while (send > smp && (send[-1] == '\n' || send[-1] == '\r')) --send;
                const bool scalar_ok = (*send == '\n' || *send == '\r') && filter_idx < 0 && scalar_tier &&
                                       fast_samples_scalar(smp, send, S, pl, nf, kinds.data(), m32.data(), filtered.data(),
                                                           in->n_filters, names.data(), vptr.data(), ob);
                if (!scalar_ok) ob.n = 0;
                if ((scalar_ok || fast_samples(smp, smp + smp_len, S, pl, nf, kinds.data(), m32.data(), filtered.data(), in->n_filters,
                                               names.data(), vptr.data(), ob)) && ob.n <= need) {
                    memcpy(dst, head, hl);
                    dst[hl + (size_t)ob.n] = '\n';
                    rptr[(size_t)l] = dst;
                    rlen[(size_t)l] = hl + (size_t)ob.n + 1;
                    cur_n += rlen[(size_t)l];
                    g_fmt_fast.fetch_add(1, std::memory_order_relaxed);
                    if (scalar_ok) g_fmt_scalar.fetch_add(1, std::memory_order_relaxed);
                    continue;
                }
            }
            g_fmt_slow.fetch_add(1, std::memory_order_relaxed);
            if (trk_vcf_decode_formats(smp, smp_len, S, nf, dec.data(), 0) != 0) { fail(l); continue; }
            store.resize((size_t)nf);
            for (int i = 0; i < nf; ++i) {
                if (dec[(size_t)i].kind < 0) continue;
                const size_t bytes = (size_t)S * (size_t)std::max(dec[(size_t)i].ncol, 1) * 4;
                if (store[(size_t)i].size() < bytes) store[(size_t)i].resize(bytes);
                if (dec[(size_t)i].kind == TRK_VCF_COL_UCS4) memset(store[(size_t)i].data(), 0, bytes);   // NUL padded
                dec[(size_t)i].out = store[(size_t)i].data();
            }
            if (trk_vcf_decode_formats(smp, smp_len, S, nf, dec.data(), 1) != 0) { fail(l); continue; }
            // genotypes: [S, pl + 1] with the phase column; a filtered call is all '.' and unphased (:721-727)
            gtrow.assign((size_t)S * (pl + 1), 0);
            for (int s = 0; s < S; ++s) {
                int16_t* g = &gtrow[(size_t)s * (pl + 1)];
                const int16_t* src = in->gt + ((size_t)l * S + s) * P;
                if (filtered[(size_t)s]) {
                    for (int j = 0; j < pl; ++j) g[j] = -1;
                    g[pl] = 0;
                } else {
                    for (int j = 0; j < pl; ++j) g[j] = src[j];
                    g[pl] = in->phased ? in->phased[(size_t)l * S + s] : 0;
                }
            }
            // every other field of a filtered call is missing (:730-746)
            for (int i = 0; i < nf; ++i) {
                const trk_vcf_decode& d = dec[(size_t)i];
                if (d.kind < 0) continue;
                for (int s = 0; s < S; ++s) {
                    if (!filtered[(size_t)s]) continue;
                    if (d.kind == TRK_VCF_COL_INT) {
                        int32_t* o = static_cast<int32_t*>(d.out) + (size_t)s * d.ncol;
                        for (int j = 0; j < d.ncol; ++j) o[j] = INT_MISSING;
                    } else if (d.kind == TRK_VCF_COL_FLOAT) {
                        float* o = static_cast<float*>(d.out) + (size_t)s * d.ncol;
                        for (int j = 0; j < d.ncol; ++j) o[j] = NAN;
                    } else {
                        uint32_t* o = static_cast<uint32_t*>(d.out) + (size_t)s * d.ncol;
                        o[0] = '.';
                        for (int j = 1; j < d.ncol; ++j) o[j] = 0;
                    }
                }
            }
            trk_vcf_callfilter cf{m32.data(), in->n_filters, 0, names.data(), vptr.data()};
            cols.clear();
            int64_t need = 0;
            for (int i = 0; i < nf; ++i) {
                const trk_vcf_decode& d = dec[(size_t)i];
                if (i == filter_idx) {
                    cols.push_back({TRK_VCF_COL_CALLFILTER, 1, 0, 0, &cf});
                } else if (d.kind < 0) {
                    cols.push_back({TRK_VCF_COL_GT, pl + 1, 0, 0, gtrow.data()});
                    need += (int64_t)S * (7 * (pl + 1) + 1);
                } else if (d.kind == TRK_VCF_COL_UCS4) {
                    cols.push_back({TRK_VCF_COL_UCS4, 1, d.ncol * 4, 0, d.out});
                    need += (int64_t)S * (d.ncol * 4 + 2);
                } else {
                    cols.push_back({d.kind, d.ncol, 0, 0, d.out});
                    need += (int64_t)S * (17 * d.ncol + 1);
                }
            }
            if (filter_idx < 0) cols.push_back({TRK_VCF_COL_CALLFILTER, 1, 0, 0, &cf});
            int64_t cfw = 8;
            for (int k = 0; k < in->n_filters; ++k) cfw += (int64_t)strlen(names[(size_t)k]) + 26;
            need += (int64_t)S * cfw;
            // formatted into this thread's chunk behind the head (sized by a generous bound, never value-initialised)
            // (format_range directly: trk_vcf_format_samples would hand the record to its sample-range thread pool,
            // and thirty-two callers queueing on that one pool serialise -- here the records are the parallel axis)
            char* dst = room(hl + (size_t)need + 1);
            OutBuf ob{dst + hl, need, 0};
            format_range(0, S, (int)cols.size(), cols.data(), ob);
            const int64_t w = ob.n;
            if (w > need) { fail(l); continue; }
            memcpy(dst, head, hl);
            dst[hl + (size_t)w] = '\n';
            rptr[(size_t)l] = dst;
            rlen[(size_t)l] = hl + (size_t)w + 1;
            cur_n += rlen[(size_t)l];
        }
    };
    struct GiveBack {      // the chunks return to the pool however the call ends
        std::vector<FmtChunk>& v;
        ~GiveBack() { for (auto& c : v) fmt_chunks().give(c); }
    } give_back{used_chunks};
    int want = in->n_threads > 0 ? in->n_threads : default_threads(32);
    if (in->n_threads <= 0)
        if (const char* e = getenv("TRK_FMT_THREADS")) want = std::max(1, atoi(e));    // formatter threads (default 32)
    const int nt = std::max(1, std::min({want, 128, n}));
    const bool timing = trk_opt("TRK_FMT_TIMING") != nullptr;
    const auto tf0 = std::chrono::steady_clock::now();
    run_on_caller_pool(nt, runner);
    const auto tf1 = std::chrono::steady_clock::now();
    if (bad.load() != INT32_MAX) {
        if (err_record) *err_record = bad.load();
        return INT64_MIN + 1;
    }
    if (need_heads.load() > 0) return INT64_MIN + 2;   // ext->need_head says which records want their head from the caller
    int64_t total = 0;
    for (size_t i = 0; i < (size_t)n; ++i) total += (int64_t)(rlen[i] + glen[i]);
    if (!out || total > cap) return -total;
    const bool emit = ext && ext->dev_emit;
    if (ext && !emit && ext->dev_regions && ext->dev_wait && ext->dev_wait(ext->dev_wait_arg) != 0) {   // the columns' download
        if (err_record) *err_record = -1;
        return INT64_MIN + 1;
    }
    {   // the lines land at their prefix offsets, copied by the same number of threads
        std::vector<int64_t> at((size_t)n + 1, 0);
        for (int i = 0; i < n; ++i) at[(size_t)i + 1] = at[(size_t)i] + (int64_t)(rlen[(size_t)i] + glen[(size_t)i]);
        if (emit) {
            // whole-record emit: the device's columns come straight to their places (the callee may scribble over the rest
            // of out[0, total): heads and host-written records go in after it)
            std::vector<int64_t> rec_off((size_t)n, -1);
            for (int i = 0; i < n; ++i)
                if (on_device[(size_t)i]) rec_off[(size_t)i] = at[(size_t)i] + (int64_t)rlen[(size_t)i];
            if (ext->dev_emit(ext->dev_emit_arg, rec_off.data(), total, out) != 0) {
                if (err_record) *err_record = -1;
                return INT64_MIN + 1;
            }
        }
        std::atomic<int> nx{0};
        auto copier = [&]() {
            for (;;) {
                const int i0 = nx.fetch_add(16);
                if (i0 >= n) break;
                for (int i = i0; i < std::min(n, i0 + 16); ++i)
                    if (rlen[(size_t)i]) memcpy(out + at[(size_t)i], rptr[(size_t)i], rlen[(size_t)i]);
                for (int i = i0; i < std::min(n, i0 + 16); ++i)
                    if (glen[(size_t)i]) {
                        char* d = out + at[(size_t)i] + rlen[(size_t)i];
                        if (gptr[(size_t)i]) memcpy(d, gptr[(size_t)i], glen[(size_t)i] - 1);     // (NULL: dev_emit put them there)
                        d[glen[(size_t)i] - 1] = '\n';
                    }
            }
        };
        run_on_caller_pool(nt, copier);
    }
    if (timing)
        fprintf(stderr, "[trk_vcf] %d records written: format %.1f ms, gather %.1f ms, %d threads, %.0f MB; since the process "
                        "started %ld records by the span writer (%ld its scalar tier), %ld decoded\n", n,
                std::chrono::duration<double, std::milli>(tf1 - tf0).count(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tf1).count(), nt, total * 1e-6,
                (long)g_fmt_fast.load(), (long)g_fmt_scalar.load(), (long)g_fmt_slow.load());
    return total;
}

}  // namespace

extern "C" {

int64_t trk_vcf_dumpstr_lines(const trk_vcf_batch* b, const trk_vcf_dumpstr* in, char* out, int64_t cap,
                              int32_t* err_record) {
    if (!in || !in->heads) return INT64_MIN;
    return dumpstr_impl(b, in, nullptr, out, cap, err_record);
}

void trk_vcf_dumpstr_stats(int64_t* fast, int64_t* decoded, int64_t* caller_heads) {
    if (fast) *fast = g_fmt_fast.load();
    if (decoded) *decoded = g_fmt_slow.load();
    if (caller_heads) *caller_heads = g_fmt_py_heads.load();
}

int trk_vcf_format_kinds(const trk_vcf_batch* b, const trk_vcf_dumpstr* in, uint8_t* kinds16, uint8_t* n_fields) {
    if (!b || !in || !kinds16 || !n_fields) return 2;
    const int n = b->n_records;
    for (int l = 0; l < n; ++l) {
        uint8_t* k16 = kinds16 + (size_t)l * 16;
        memset(k16, 0, 16);
        n_fields[l] = 0;
        const char* line = b->text + b->line_off[l];
        const int32_t* fo = b->field_off + (size_t)l * 10;
        const int64_t line_len = b->line_end[l] - b->line_off[l];
        if (fo[9] <= fo[8] || fo[9] >= line_len) continue;
        const char* f = line + fo[8];
        const char* fe = line + fo[9] - 1;
        int nf = 0;
        bool ok = true, have_gt = false;
        for (const char* a = f; a <= fe;) {
            const char* c = find_ch(a, fe, ':');
            const size_t kl = (size_t)(c - a);
            int kind = 4;
            if (kl == 2 && a[0] == 'G' && a[1] == 'T') {
                kind = 1;
                if (have_gt) ok = false;
                have_gt = true;
            } else if (kl == 6 && memcmp(a, "FILTER", 6) == 0) {
                ok = false;
            } else {
                for (int t = 0; t < in->n_format_keys; ++t)
                    if (strlen(in->format_keys[t]) == kl && memcmp(in->format_keys[t], a, kl) == 0) {
                        const int fk = in->format_kinds[t];
                        kind = fk == TRK_VCF_COL_INT ? 2 : fk == TRK_VCF_COL_FLOAT ? 3 : fk == TRK_VCF_COL_UCS4 ? 4 : 0;
                        break;
                    }
            }
            if (kind == 0 || nf >= 16) { ok = false; break; }
            k16[nf++] = (uint8_t)kind;
            if (c >= fe) break;
            a = c + 1;
        }
        if (ok && have_gt) n_fields[l] = (uint8_t)nf;
    }
    return 0;
}

int64_t trk_vcf_dumpstr_records(const trk_vcf_batch* b, const trk_vcf_dumpstr2* in, char* out, int64_t cap,
                                int32_t* err_record) {
    if (!in || !in->hrun || !in->have_stats || !in->het || !in->hwep || !in->allele_count || !in->allele_off ||
        (in->n_info_keys > 0 && (!in->info_keys || !in->info_kinds)))
        return INT64_MIN;
    return dumpstr_impl(b, &in->base, in, out, cap, err_record);
}

}  // extern "C"

// ---------------------------------------------------------------------------------------
// BGZF output (round 6; include/trk_vcf.h): dumpSTR --zip.  The reference writes its VCF and runs `bgzip -f` over it
// (dumpSTR.py:1241-1245, 1347-1352); the Python writer of rounds 1-5 (trtools_amd/bgzf.py: zlib members on a thread pool of
// the interpreter) made 45-100 MB/s of a 1.5 GB output -- 15-30 s behind a command line that takes 0.2 s without --zip.
// Here the members of a block of text are compressed on the caller-side worker pool, libdeflate where the image has it.
// ---------------------------------------------------------------------------------------
namespace {
constexpr size_t BGZF_TEXT = 0xff00;                 // bytes of text per member (bgzip's own choice)
constexpr size_t BGZF_SLOT = 65536 + 256;            // a member never needs more: stored form = text + 5 + 26
constexpr unsigned char BGZF_EOF[28] = {0x1f, 0x8b, 0x08, 0x04, 0, 0, 0, 0, 0, 0xff, 0x06, 0, 0x42, 0x43, 0x02, 0,
                                        0x1b, 0, 0x03, 0, 0, 0, 0, 0, 0, 0, 0, 0};

// one member into slot[0 .. BGZF_SLOT); returns its length, 0: failed
size_t bgzf_member(const unsigned char* text, size_t n, int level, void* ld_comp, unsigned char* slot) {
    const Deflater& d = deflater();
    unsigned char* payload = slot + 18;
    const size_t room = BGZF_SLOT - 18 - 8;
    size_t clen = 0;
    uint32_t crc = 0;
    if (ld_comp) {
        clen = d.c_compress(ld_comp, text, n, payload, room);
        crc = d.c_crc32(0, text, n);
    } else if (level > 0) {
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return 0;
        zs.next_in = const_cast<unsigned char*>(text);
        zs.avail_in = (uInt)n;
        zs.next_out = payload;
        zs.avail_out = (uInt)room;
        const int rc = deflate(&zs, Z_FINISH);
        clen = rc == Z_STREAM_END ? (size_t)zs.total_out : 0;
        deflateEnd(&zs);
        crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), text, (uInt)n);
    }
    if (clen == 0) {
        // stored (level 0, or text the compressor could not fit): one final stored block -- 5 bytes + the text
        if (n > 0xffff || n + 5 > room) return 0;
        payload[0] = 1;
        payload[1] = (unsigned char)(n & 0xff);
        payload[2] = (unsigned char)(n >> 8);
        payload[3] = (unsigned char)(~n & 0xff);
        payload[4] = (unsigned char)((~n >> 8) & 0xff);
        if (n) memcpy(payload + 5, text, n);
        clen = n + 5;
        if (!ld_comp || level == 0) crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), text, (uInt)n);
    }
    const size_t total = clen + 26;
    if (total > 65536) return 0;
    const unsigned char head[18] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0,
                                    (unsigned char)((total - 1) & 0xff), (unsigned char)((total - 1) >> 8)};
    memcpy(slot, head, 18);
    unsigned char* tail = payload + clen;
    for (int k = 0; k < 4; ++k) tail[k] = (unsigned char)(crc >> (8 * k));
    for (int k = 0; k < 4; ++k) tail[4 + k] = (unsigned char)((uint32_t)n >> (8 * k));
    return total;
}
}  // namespace

// the CRC-32 of every member of a text (trk_deflate_bgzf, trk_api.hip: the device makes the DEFLATE streams, the text's
// checksums are computed here while it does)
extern "C" __attribute__((visibility("hidden"))) void trk_member_crc32(const void* text, size_t n, size_t member, uint32_t* crc) {
    const size_t nb = (n + member - 1) / member;
    if (!nb) return;
    const unsigned char* t = static_cast<const unsigned char*>(text);
    const Deflater& d = deflater();
    int want = default_threads(32);
    if (const char* e = getenv("TRK_FMT_THREADS")) want = std::max(1, atoi(e));
    const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::min(want, 128), (nb + 15) / 16));
    std::atomic<size_t> next{0};
    const std::function<void()> job = [&]() {
        for (;;) {
            const size_t b0 = next.fetch_add(16);
            if (b0 >= nb) break;
            for (size_t b = b0; b < std::min(nb, b0 + 16); ++b) {
                const size_t at = b * member, m = std::min(member, n - at);
                crc[b] = d.c_crc32 ? d.c_crc32(0, t + at, m) : (uint32_t)crc32(crc32(0L, Z_NULL, 0), t + at, (uInt)m);
            }
        }
    };
    run_on_caller_pool(nt, job);
}

extern "C" {

static bool plain_digits(const char* p, const char* e, int64_t* v) {
    if (p == e || e - p > 18) return false;
    int64_t x = 0;
    for (; p < e; ++p) {
        if (*p < '0' || *p > '9') return false;
        x = x * 10 + (*p - '0');
    }
    *v = x;
    return true;
}

int64_t trk_text_record_places(const void* text, size_t n, const int64_t* nl, size_t n_nl, int64_t* out, size_t cap) {
    const char* t = static_cast<const char*>(text);
    int64_t k = 0;
    size_t pos = 0;
    const char* prev = nullptr;
    size_t prev_len = 0;
    for (size_t i = 0; i <= n_nl && pos < n; ++i) {
        const size_t e = i < n_nl ? (size_t)nl[i] : n;          // the line is [pos, e)
        if (e > pos && t[pos] != '#') {
            if ((size_t)k < cap) {
                int64_t* row = out + (size_t)k * 8;
                // the first eight columns: col[c] begins at b[c] and ends at b[c + 1] - 1
                const char* b[9];
                int nc = 0;
                const char* q = t + pos;
                const char* const le = t + e;
                b[0] = q;
                while (nc < 8) {
                    const char* tab = static_cast<const char*>(memchr(q, '\t', (size_t)(le - q)));
                    ++nc;
                    if (!tab) {
                        b[nc] = le + 1;
                        break;
                    }
                    b[nc] = tab + 1;
                    q = tab + 1;
                }
                const size_t clen = (size_t)(b[1] - 1 - b[0]);
                row[0] = (int64_t)pos;
                row[1] = (int64_t)std::min(e + 1, n);
                row[2] = row[3] = -1;
                row[4] = (int64_t)(b[0] - t);
                row[5] = (int64_t)clen;
                row[6] = !(prev && prev_len == clen && memcmp(prev, b[0], clen) == 0);
                prev = b[0];
                prev_len = clen;
                int64_t p1 = 0;
                bool odd = nc < 4 || !plain_digits(b[1], b[2] - 1, &p1);
                if (!odd) {
                    for (const char* r = b[3]; r < b[4] - 1; ++r) odd |= (unsigned char)*r >= 0x80;
                }
                if (!odd) {
                    const int64_t beg = p1 - 1;
                    int64_t end = beg + (int64_t)(b[4] - 1 - b[3]);
                    if (nc >= 8) {                 // INFO: the first item that begins with END=
                        const char* it = b[7];
                        const char* const ie = b[8] - 1;
                        while (it <= ie) {
                            const char* semi = static_cast<const char*>(memchr(it, ';', (size_t)(ie - it)));
                            const char* const item_end = semi ? semi : ie;
                            if (item_end - it >= 4 && memcmp(it, "END=", 4) == 0) {
                                int64_t ev = 0;
                                if (plain_digits(it + 4, item_end, &ev)) {
                                    if (ev > beg) end = ev;
                                } else {
                                    odd = true;    // (whatever python's int() makes of it: the caller's)
                                }
                                break;
                            }
                            if (!semi) break;
                            it = semi + 1;
                        }
                    }
                    row[2] = beg;
                    row[3] = std::max(end, beg + 1);
                }
                row[7] = odd;
                if (odd) row[2] = row[3] = -1;
            }
            ++k;
        }
        pos = e + 1;
    }
    return k;
}

int64_t trk_bgzf_member_offsets(const void* data, size_t n, uint64_t* out, size_t cap) {
    const uint8_t* d = static_cast<const uint8_t*>(data);
    size_t pos = 0;
    int64_t k = 0;
    while (pos < n) {
        if (n - pos < 18 || d[pos] != 0x1f || d[pos + 1] != 0x8b || d[pos + 12] != 'B' || d[pos + 13] != 'C') return -1;
        if ((size_t)k < cap) out[k] = pos;
        ++k;
        pos += ((size_t)d[pos + 16] | ((size_t)d[pos + 17] << 8)) + 1;
    }
    return pos == n ? k : -1;
}

int64_t trk_text_newlines(const void* text, size_t n, int64_t* out, size_t cap) {
    if (!text || !n) return 0;
    const char* base = static_cast<const char*>(text);
    constexpr size_t CH = (size_t)1 << 20;
    const size_t nch = (n + CH - 1) / CH;
    std::vector<std::vector<int64_t>> part(nch);
    std::atomic<size_t> nx{0};
    const std::function<void()> job = [&]() {
        for (;;) {
            const size_t c = nx.fetch_add(1);
            if (c >= nch) break;
            const char* q = base + c * CH;
            const char* const qe = base + std::min(n, (c + 1) * CH);
            while ((q = static_cast<const char*>(memchr(q, '\n', (size_t)(qe - q)))) != nullptr) {
                part[c].push_back((int64_t)(q - base));
                ++q;
            }
        }
    };
    run_on_caller_pool((int)std::min<size_t>((size_t)default_threads(32), nch), job);
    int64_t total = 0;
    for (const auto& pc : part) {
        for (int64_t v : pc) {
            if ((size_t)total < cap && out) out[total] = v;
            ++total;
        }
    }
    return total;
}

size_t trk_bgzf_bound(size_t n) { return ((n + BGZF_TEXT - 1) / BGZF_TEXT + 1) * BGZF_SLOT; }

size_t trk_bgzf_eof(void* out28) {
    memcpy(out28, BGZF_EOF, sizeof BGZF_EOF);
    return sizeof BGZF_EOF;
}

int trk_bgzf_compress(const void* data, size_t n, int level, int n_threads, void* out, size_t out_cap, size_t* out_bytes) {
    if (out_bytes) *out_bytes = 0;
    if ((!data && n) || !out || !out_bytes || level < 0 || level > 9) return 2;
    if (out_cap < trk_bgzf_bound(n)) return 1;
    const size_t nb = (n + BGZF_TEXT - 1) / BGZF_TEXT;
    if (nb == 0) return 0;
    const unsigned char* text = static_cast<const unsigned char*>(data);
    unsigned char* slots = static_cast<unsigned char*>(out);
    std::vector<uint32_t> len(nb, 0);
    const Deflater& d = deflater();
    const bool use_ld = d.c_compress != nullptr && level > 0 && trk_opt("TRK_BGZF_ZLIB") == nullptr;
    int want = n_threads > 0 ? n_threads : default_threads(32);
    if (n_threads <= 0)
        if (const char* e = getenv("TRK_FMT_THREADS")) want = std::max(1, atoi(e));
    const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::min(want, 128), nb));
    std::atomic<size_t> next{0};
    std::atomic<int> bad{0};
    const std::function<void()> job = [&]() {
        void* comp = use_ld ? d.c_alloc(level) : nullptr;     // (a compressor per thread and call: 1-2 us against milliseconds of work)
        if (use_ld && !comp) {
            bad.store(1);
            return;
        }
        for (;;) {
            const size_t b = next.fetch_add(1);
            if (b >= nb) break;
            const size_t at = b * BGZF_TEXT, m = std::min(BGZF_TEXT, n - at);
            const size_t got = bgzf_member(text + at, m, level, comp, slots + b * BGZF_SLOT);
            if (!got) bad.store(1);
            len[b] = (uint32_t)got;
        }
        if (comp) d.c_release(comp);
    };
    run_on_caller_pool(nt, job);
    if (bad.load()) return 2;
    // the members back to back (member 0 is in place; the others move down)
    size_t at = len[0];
    for (size_t b = 1; b < nb; ++b) {
        memmove(slots + at, slots + b * BGZF_SLOT, len[b]);
        at += len[b];
    }
    *out_bytes = at;
    return 0;
}

}  // extern "C"
