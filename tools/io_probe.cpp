// Lease probe (host side only): how fast does an 8 GB image leave the process on this box?  Writes SIZE bytes from an
// in-memory buffer to PATH in several ways and prints GB/s (page-cache writes, the .rl_bwt writer's situation).
//   g++ -O2 -pthread -o /tmp/io_probe tools/io_probe.cpp && /tmp/io_probe /tmp/io_probe.out 8
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void pwrite_all(int fd, const char *p, size_t n, off_t off) {
    while (n) {
        ssize_t r = pwrite(fd, p, n, off);
        if (r <= 0) { perror("pwrite"); exit(1); }
        p += r; n -= (size_t)r; off += r;
    }
}

int main(int argc, char **argv) {
    const std::string path = argc > 1 ? argv[1] : "/tmp/io_probe.out";
    const size_t gb = argc > 2 ? (size_t)atoi(argv[2]) : 8;
    const size_t size = gb << 30;
    char *buf = (char *)mmap(nullptr, size, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (buf == MAP_FAILED) { perror("mmap"); return 1; }
    for (size_t i = 0; i < size; i += 4096) buf[i] = (char)(i >> 12);
    auto run = [&](const char *name, int threads, bool fallocate_first, int flags, size_t piece) {
        unlink(path.c_str());
        sync();
        const double t0 = now();
        int fd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC | flags, 0644);
        if (fd < 0) { printf("%-60s open failed\n", name); return; }
        if (fallocate_first) { if (posix_fallocate(fd, 0, (off_t)size) != 0) printf("(fallocate failed) "); }
        else if (ftruncate(fd, (off_t)size) != 0) perror("ftruncate");
        const double t1 = now();
        std::vector<std::thread> th;
        // interleaved pieces: thread k writes pieces k, k + threads, ... (what a chunked double-buffered writer does per chunk)
        for (int k = 0; k < threads; k++)
            th.emplace_back([&, k] {
                for (size_t off = (size_t)k * piece; off < size; off += (size_t)threads * piece)
                    pwrite_all(fd, buf + off, std::min(piece, size - off), (off_t)off);
            });
        for (auto &x : th) x.join();
        const double t2 = now();
        close(fd);
        const double t3 = now();
        printf("%-60s %6.2f GB/s   (open+size %.3f s, write %.3f s, close %.3f s)\n", name, size / 1e9 / (t3 - t0), t1 - t0, t2 - t1, t3 - t2);
        fflush(stdout);
    };
    printf("file: %s, %zu GiB\n", path.c_str(), gb);
    run("1 thread, 64 MiB pieces", 1, false, 0, (size_t)64 << 20);
    run("4 threads, 16 MiB pieces", 4, false, 0, (size_t)16 << 20);
    run("16 threads, 4 MiB pieces", 16, false, 0, (size_t)4 << 20);
    run("16 threads, 64 MiB pieces (disjoint far-apart ranges)", 16, false, 0, (size_t)64 << 20);
    run("16 threads, 4 MiB pieces, posix_fallocate first", 16, true, 0, (size_t)4 << 20);
    run("16 threads, 4 MiB pieces, O_DIRECT", 16, false, O_DIRECT, (size_t)4 << 20);
    run("64 threads, 1 MiB pieces", 64, false, 0, (size_t)1 << 20);
    for (int threads : {1, 8, 16, 32}) for (int prealloc = 0; prealloc < 2; prealloc++) {
        // the file mapped MAP_SHARED and filled by memcpy from `threads` threads (no inode lock per write: page faults instead)
        unlink(path.c_str());
        sync();
        const double t0 = now();
        int fd = open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0644);
        if (prealloc) { if (posix_fallocate(fd, 0, (off_t)size) != 0) printf("(fallocate failed) "); }
        else if (ftruncate(fd, (off_t)size) != 0) perror("ftruncate");
        char *m = (char *)mmap(nullptr, size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (m == MAP_FAILED) { perror("mmap file"); close(fd); continue; }
        const double t1 = now();
        std::vector<std::thread> th;
        const size_t piece = (size_t)4 << 20;
        for (int k = 0; k < threads; k++)
            th.emplace_back([&, k] { for (size_t off = (size_t)k * piece; off < size; off += (size_t)threads * piece) memcpy(m + off, buf + off, std::min(piece, size - off)); });
        for (auto &x : th) x.join();
        const double t2 = now();
        munmap(m, size);
        close(fd);
        const double t3 = now();
        printf("mmap MAP_SHARED + memcpy, %2d threads, 4 MiB pieces%-12s %6.2f GB/s   (open+size+map %.3f s, copy %.3f s, unmap+close %.3f s)\n", threads,
               prealloc ? ", fallocate" : "", size / 1e9 / (t3 - t0), t1 - t0, t2 - t1, t3 - t2);
        fflush(stdout);
    }
    {   // N files instead of one (not a drop-in, but it tells whether the inode is the bottleneck)
        const int nf = 8;
        const double t0 = now();
        std::vector<std::thread> th;
        for (int k = 0; k < nf; k++)
            th.emplace_back([&, k] {
                const std::string p = path + "." + std::to_string(k);
                unlink(p.c_str());
                int fd = open(p.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
                const size_t a = size / nf * k, b = k == nf - 1 ? size : size / nf * (k + 1);
                for (size_t off = a; off < b; off += (size_t)16 << 20) pwrite_all(fd, buf + off, std::min((size_t)16 << 20, b - off), (off_t)(off - a));
                close(fd);
            });
        for (auto &x : th) x.join();
        printf("%-60s %6.2f GB/s\n", "8 files, one thread each", size / 1e9 / (now() - t0));
        for (int k = 0; k < nf; k++) unlink((path + "." + std::to_string(k)).c_str());
    }
    {   // reading it back the way the loader does (file in the page cache -> memory), 16 threads
        run("(rewrite for the read test) 16 threads, 4 MiB", 16, false, 0, (size_t)4 << 20);
        int fd = open(path.c_str(), O_RDONLY);
        for (int threads : {1, 8, 16, 32, 64}) {
            const double t0 = now();
            std::vector<std::thread> th;
            for (int k = 0; k < threads; k++)
                th.emplace_back([&, k] {
                    const size_t piece = (size_t)4 << 20;
                    for (size_t off = (size_t)k * piece; off < size; off += (size_t)threads * piece) {
                        size_t n = std::min(piece, size - off), done = 0;
                        while (done < n) { ssize_t r = pread(fd, buf + off + done, n - done, (off_t)(off + done)); if (r <= 0) break; done += (size_t)r; }
                    }
                });
            for (auto &x : th) x.join();
            printf("pread from the page cache, %2d threads, 4 MiB pieces            %6.2f GB/s\n", threads, size / 1e9 / (now() - t0));
        }
        close(fd);
    }
    unlink(path.c_str());
    system("df -T /tmp | tail -2; nproc");
    return 0;
}
