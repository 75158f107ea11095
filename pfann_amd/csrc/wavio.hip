// Host-side WAV reader of the drop-in CLIs: the decode workers of the reference's builder
// (DataLoader(num_workers=4) over MusicDataset, builder.py:66; datautil/audio.py:130-149 reads WAV files through the
// `wave` module and accepts 16-bit PCM only) as native threads that read the samples of a whole launch group straight
// into ONE pinned slab, so that the group reaches the GPU by a single copy.  No device code in this file.
#include <fcntl.h>
#include <stdint.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

#include "../../include/pfann_amd.h"

namespace {

inline uint32_t rd32(const unsigned char *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint16_t rd16(const unsigned char *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

bool read_full(int fd, void *dst, size_t n, off_t pos) {
    char *d = static_cast<char *>(dst);
    while (n) {
        const ssize_t g = pread(fd, d, n, pos);
        if (g <= 0) return false;
        d += g; pos += g; n -= (size_t)g;
    }
    return true;
}

// Walks the RIFF chunks like the `wave` module does (wave.py: first "fmt " then "data"; other chunks skipped, odd sizes
// padded).  n_frames = whole frames that are really in the file (a truncated data chunk yields what is there).
void probe_one(const char *path, pfann_wav_info *o) {
    o->n_frames = 0; o->data_pos = 0; o->n_ch = 0; o->sample_rate = 0; o->status = PFANN_WAV_EOPEN; o->reserved = 0;
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return;
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return; }
    const int64_t fsize = st.st_size;
    unsigned char h[40];
    o->status = PFANN_WAV_EFORMAT;
    if (fsize < 12 || !read_full(fd, h, 12, 0) || memcmp(h, "RIFF", 4) != 0 || memcmp(h + 8, "WAVE", 4) != 0) { close(fd); return; }
    int64_t pos = 12;
    bool have_fmt = false;
    while (pos + 8 <= fsize) {
        if (!read_full(fd, h, 8, pos)) break;
        const uint32_t size = rd32(h + 4);
        pos += 8;
        if (memcmp(h, "fmt ", 4) == 0) {
            const size_t take = std::min<size_t>(size, 40);
            if (size < 16 || pos + (int64_t)take > fsize || !read_full(fd, h, take, pos)) break;
            uint16_t tag = rd16(h);
            o->n_ch = rd16(h + 2);
            o->sample_rate = (int32_t)rd32(h + 4);
            const uint16_t bits = rd16(h + 14);
            if (tag == 0xFFFE && take >= 26) tag = rd16(h + 24);          // WAVE_FORMAT_EXTENSIBLE: the sub-format's tag
            if (tag != 1 || o->n_ch < 1) { o->status = PFANN_WAV_ECODEC; close(fd); return; }
            if ((bits + 7) / 8 != 2) { o->status = PFANN_WAV_EWIDTH; close(fd); return; }
            have_fmt = true;
        } else if (memcmp(h, "data", 4) == 0) {
            if (!have_fmt) break;
            const int64_t avail = std::min<int64_t>(size, fsize - pos);
            o->n_frames = avail / (2 * (int64_t)o->n_ch);
            o->data_pos = pos;
            o->status = 0;
            close(fd);
            return;
        }
        pos += (int64_t)size + (size & 1);
    }
    close(fd);
}

void read_one(const char *path, pfann_wav_info *o, int16_t *dst) {
    if (o->status != 0 || o->n_frames <= 0) return;
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) { o->status = PFANN_WAV_EOPEN; return; }
    if (!read_full(fd, dst, (size_t)o->n_frames * o->n_ch * sizeof(int16_t), (off_t)o->data_pos)) o->status = PFANN_WAV_EREAD;
    close(fd);
}

template <class F>
void parallel_for(int n, int n_threads, F f) {
    n_threads = std::max(1, std::min(n_threads, n));
    if (n_threads == 1) { for (int i = 0; i < n; ++i) f(i); return; }
    std::atomic<int> next(0);
    std::vector<std::thread> th;
    th.reserve(n_threads);
    for (int t = 0; t < n_threads; ++t)
        th.emplace_back([&] { for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) f(i); });
    for (auto &t : th) t.join();
}

}  // namespace

extern "C" {

int pfann_wav_probe(const char *const *paths, int n, int n_threads, pfann_wav_info *info) {
    if (n < 0 || (n > 0 && (!paths || !info))) return -1;
    parallel_for(n, n_threads, [&](int i) { probe_one(paths[i], &info[i]); });
    return 0;
}

int pfann_wav_read(const char *const *paths, int n, int n_threads, pfann_wav_info *info, const int64_t *dst_off,
                   int16_t *dst, int64_t dst_cap) {
    if (n < 0 || (n > 0 && (!paths || !info || !dst_off || !dst))) return -1;
    for (int i = 0; i < n; ++i)
        if (info[i].status == 0 && (dst_off[i] < 0 || dst_off[i] + info[i].n_frames * info[i].n_ch > dst_cap)) return -2;
    parallel_for(n, n_threads, [&](int i) { read_one(paths[i], &info[i], dst + dst_off[i]); });
    return 0;
}

}  // extern "C"
