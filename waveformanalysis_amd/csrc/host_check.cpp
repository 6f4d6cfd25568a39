// Sanitizer driver for the host-only C++ of libwfa_hip.so (wfa_host.hpp), built by `make SANITIZE=1 host_check` with
// -fsanitize=address,undefined and run by tests/test_host_sanitized.py on the CPU box (the pool's GPU boxes offer no
// GPU AddressSanitizer; the reference has no sanitizer run at all: SURVEY section 5).  The device is a stand-in: an
// in-order queue of asynchronous copies executed by one thread, with a completion stamp per staging buffer -- what the
// HIP stream + events are to the real ring.
//   host_check v1725 <blob file>     header walk over the whole stream and over every truncation of it
//   host_check ring <bytes> <stage>  staged copy of a pseudo-random buffer, compared byte for byte
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <functional>
#include <mutex>
#include <vector>

#include "wfa_host.hpp"

namespace {

struct AsyncQueue {  // one worker, tasks in order (a HIP stream)
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    bool done = false;
    std::thread worker;
    AsyncQueue() : worker([this] { run(); }) {}
    ~AsyncQueue() {
        { std::lock_guard<std::mutex> l(m); done = true; }
        cv.notify_all();
        worker.join();
    }
    void push(std::function<void()> f) {
        { std::lock_guard<std::mutex> l(m); q.push_back(std::move(f)); }
        cv.notify_all();
    }
    void run() {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> l(m);
                cv.wait(l, [this] { return done || !q.empty(); });
                if (q.empty()) return;
                f = std::move(q.front());
                q.pop_front();
            }
            f();
        }
    }
};

int check_v1725(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); return 2; }
    std::vector<uint8_t> blob;
    uint8_t tmp[65536];
    size_t got;
    while ((got = fread(tmp, 1, sizeof(tmp), f)) > 0) blob.insert(blob.end(), tmp, tmp + got);
    fclose(f);
    char err[160] = "";
    auto walk = [&](int64_t n_bytes, bool print) {
        // exactly-sized copy of the prefix: a read behind it is a heap-buffer-overflow for the sanitizer
        std::vector<uint8_t> buf(blob.begin(), blob.begin() + n_bytes);
        int64_t n = 0;
        int rc = wfa::host::v1725_index(buf.data(), n_bytes, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &n, err, sizeof(err));
        if (rc) return (int64_t)-1;
        std::vector<int16_t> ch(n);
        std::vector<int64_t> ts(n), off(n);
        std::vector<uint8_t> tr(n);
        std::vector<uint16_t> bl(n);
        std::vector<int32_t> ns(n);
        int64_t n2 = 0;
        rc = wfa::host::v1725_index(buf.data(), n_bytes, n, ch.data(), ts.data(), tr.data(), bl.data(), off.data(), ns.data(), &n2, err, sizeof(err));
        if (rc || n2 != n) return (int64_t)-2;
        uint64_t d_ts = 0, d_ch = 0, d_bl = 0, samples = 0, d_first = 0;
        for (int64_t k = 0; k < n; ++k) {
            d_ts += (uint64_t)ts[k] * (uint64_t)(k + 1); d_ch += (uint64_t)ch[k]; d_bl += bl[k]; samples += (uint64_t)ns[k];
            if (ns[k] > 0) d_first += (uint64_t)(buf[off[k]] | (buf[off[k] + 1] << 8));  // first sample of every wave: inside the buffer
            if (off[k] + 2 * (int64_t)ns[k] > n_bytes) return (int64_t)-3;
        }
        if (print)
            printf("{\"waves\": %lld, \"samples\": %llu, \"ts_digest\": %llu, \"channel_sum\": %llu, \"baseline_sum\": %llu, \"first_sample_sum\": %llu}\n",
                   (long long)n, (unsigned long long)samples, (unsigned long long)d_ts, (unsigned long long)d_ch,
                   (unsigned long long)d_bl, (unsigned long long)d_first);
        return n;
    };
    const int64_t full = walk((int64_t)blob.size(), true);
    if (full < 0) { fprintf(stderr, "walk failed: %s\n", err); return 1; }
    // every truncation of the first 4 KiB and of the last 4 KiB, and a stride through the middle: never more waves than
    // the whole stream, never a read outside the prefix
    int64_t prev = 0;
    const int64_t n = (int64_t)blob.size();
    for (int64_t cut = 0; cut <= n; cut += (cut < 4096 || cut > n - 4096) ? 1 : 997) {
        const int64_t w = walk(cut, false);
        if (w < 0 || w > full || w < prev) { fprintf(stderr, "truncation at %lld: %lld waves (prev %lld, full %lld)\n", (long long)cut, (long long)w, (long long)prev, (long long)full); return 1; }
        prev = w;
    }
    // a channel block that claims fewer than 3 words is an error with a message, not a crash
    if (blob.size() >= 28) {
        std::vector<uint8_t> bad(blob.begin(), blob.begin() + 28);
        bad[4] |= 1;  // channel 0 present
        bad[16] = 2; bad[17] = 0; bad[18] &= 0xc0;
        int64_t nn = 0;
        if (wfa::host::v1725_index(bad.data(), 28, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &nn, err, sizeof(err)) != -1 ||
            !strstr(err, "< 3 words")) { fprintf(stderr, "malformed block not reported\n"); return 1; }
    }
    return 0;
}

int check_ring(size_t bytes, size_t stage_bytes) {
    std::vector<uint8_t> src(bytes), dst(bytes, 0);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < bytes; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; src[i] = (uint8_t)s; }
    std::vector<uint8_t> st0(stage_bytes), st1(stage_bytes);
    void* stage[2] = {st0.data(), st1.data()};
    std::atomic<uint64_t> queued[2] = {{0}, {0}}, finished[2] = {{0}, {0}};
    std::mutex m;
    std::condition_variable cv;
    {
        AsyncQueue stream;
        const int rc = wfa::host::staged_copy(
            dst.data(), src.data(), bytes, stage, stage_bytes,
            [&](void* d, const void* staged, size_t n, int b) {
                const uint64_t ticket = ++queued[b];
                stream.push([&, d, staged, n, b, ticket] {
                    memcpy(d, staged, n);  // reads the staging buffer LATER: a ring that refills it too early corrupts dst
                    { std::lock_guard<std::mutex> l(m); finished[b] = ticket; }
                    cv.notify_all();
                });
                return 0;
            },
            [&](int b) {
                std::unique_lock<std::mutex> l(m);
                cv.wait(l, [&] { return finished[b].load() == queued[b].load(); });
                return 0;
            },
            /*serial_below=*/stage_bytes / 2);  // small enough for the copy threads to take part
        if (rc) return 1;
    }  // the queue drains here (hipStreamSynchronize)
    if (bytes && memcmp(src.data(), dst.data(), bytes) != 0) { fprintf(stderr, "staged copy differs from its source\n"); return 1; }
    printf("{\"bytes\": %zu, \"stage_bytes\": %zu, \"chunks\": %zu}\n", bytes, stage_bytes, (bytes + stage_bytes - 1) / stage_bytes);
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc == 3 && !strcmp(argv[1], "v1725")) return check_v1725(argv[2]);
    if (argc == 4 && !strcmp(argv[1], "ring")) return check_ring((size_t)atoll(argv[2]), (size_t)atoll(argv[3]));
    fprintf(stderr, "usage: host_check v1725 <blob> | ring <bytes> <stage bytes>\n");
    return 2;
}
