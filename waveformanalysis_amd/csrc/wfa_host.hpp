// Host-only pieces of libwfa_hip.so as plain C++17 (no HIP): the code here is what the library runs on the CPU side of
// the path, and what `make SANITIZE=1 host_check` builds with AddressSanitizer + UndefinedBehaviorSanitizer behind a
// stand-in for the device (csrc/host_check.cpp; SURVEY section 5: the reference has no sanitizer run of its own).
//   * v1725_index: header walk of a CAEN V1725 DAW_DEMO binary stream (reference utils/formats/v1725.py:66-114);
//   * staged_copy: host -> device through two pinned staging buffers, a few host threads filling one while the other is
//     on the wire (the ring of wfa_upload_pool_u16 / wfa_upload_pool_f32).
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>

namespace wfa {
namespace host {

// Returns 0, or -1 with a message in err (the stream is malformed: a channel block shorter than its own header).
// capacity = rows the output columns hold (0: count only); *n_waves = waves in the stream either way.
inline int v1725_index(const uint8_t* buf, int64_t n_bytes, int64_t capacity, int16_t* channel, int64_t* timestamp,
                       uint8_t* trunc, uint16_t* baseline, int64_t* payload_offset, int32_t* n_samples, int64_t* n_waves,
                       char* err, size_t err_len) {
    int64_t pos = 0, k = 0;
    bool stop = false;
    while (!stop && n_bytes - pos >= 16) {  // a short event header ends the stream
        const uint8_t* eh = buf + pos;
        pos += 16;
        const unsigned mask = (unsigned)eh[4] | ((unsigned)eh[11] << 8);
        for (int ch = 0; ch < 16 && !stop; ++ch) {
            if (!((mask >> ch) & 1u)) continue;
            if (n_bytes - pos < 12) { pos = n_bytes; stop = true; break; }  // short channel header
            const uint8_t* h = buf + pos;
            pos += 12;
            const int64_t ch_size = ((int64_t)h[0] | ((int64_t)h[1] << 8) | ((int64_t)h[2] << 16)) & 0x3fffff;
            if (ch_size < 3) {
                if (err && err_len)
                    snprintf(err, err_len, "V1725 channel size %lld < 3 words at byte %lld", (long long)ch_size, (long long)(pos - 12));
                return -1;
            }
            const int64_t sig_bytes = (ch_size - 3) << 2;
            if (n_bytes - pos < sig_bytes) { pos = n_bytes; stop = true; break; }  // short waveform
            if (k < capacity) {
                int64_t ts = 0;
                for (int b = 5; b >= 0; --b) ts = (ts << 8) | h[4 + b];
                channel[k] = (int16_t)ch;
                timestamp[k] = ts;
                trunc[k] = (uint8_t)((h[3] >> 6) & 1u);
                baseline[k] = (uint16_t)(h[10] | (h[11] << 8));
                payload_offset[k] = pos;
                n_samples[k] = (int32_t)(sig_bytes >> 1);
            }
            ++k;
            pos += sig_bytes;
        }
    }
    *n_waves = k;
    return 0;
}

constexpr int kStageThreads = 4;

inline void parallel_memcpy(void* dst, const void* src, size_t bytes, size_t serial_below = (8u << 20)) {
    if (bytes < serial_below) { memcpy(dst, src, bytes); return; }
    std::thread th[kStageThreads - 1];
    const size_t part = (bytes / kStageThreads + 4095) & ~(size_t)4095;
    for (int t = 1; t < kStageThreads; ++t) {
        const size_t o = (size_t)t * part;
        if (o >= bytes) break;
        const size_t n = o + part < bytes ? part : bytes - o;
        th[t - 1] = std::thread([=] { memcpy((char*)dst + o, (const char*)src + o, n); });
    }
    memcpy(dst, src, part < bytes ? part : bytes);
    for (auto& x : th) if (x.joinable()) x.join();
}

// src -> dst in chunks of stage_bytes through stage[0] / stage[1].  `copy(dst, staged, n, b)` queues the device copy out of
// staging buffer b and marks its completion; `wait(b)` blocks until the last copy queued out of buffer b has finished.
// Both return 0 or an error code, which ends the transfer.  The caller waits for the whole queue afterwards.
template <class Copy, class Wait>
inline int staged_copy(void* dst, const void* src, size_t bytes, void* const stage[2], size_t stage_bytes, Copy copy, Wait wait,
                       size_t serial_below = (8u << 20)) {
    bool used[2] = {false, false};
    int b = 0;
    for (size_t off = 0; off < bytes; off += stage_bytes, b ^= 1) {
        const size_t n = bytes - off < stage_bytes ? bytes - off : stage_bytes;
        if (used[b])
            if (int rc = wait(b)) return rc;  // the copy out of this buffer has finished
        parallel_memcpy(stage[b], (const char*)src + off, n, serial_below);
        if (int rc = copy((char*)dst + off, stage[b], n, b)) return rc;
        used[b] = true;
    }
    return 0;
}

}  // namespace host
}  // namespace wfa
