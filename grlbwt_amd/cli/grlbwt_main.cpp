// grlbwt -- command line of the MI355X-native BCR-BWT engine.
//
// Drop-in for the reference CLI (ddiazdom/grlBWT main.cpp:43-154): same flags, same
// output naming (main.cpp:112-113), same stdout labels, same exit behaviour
// (0 ok, 1 ill-formed input, 105 validation error, 106 missing TEXT), over the
// C-ABI of include/grlbwt_hip.h.  Own argument parser (CLI11 is third-party).
// -t/-f/-b/-T are accepted and validated as in the reference; they are tuning knobs of
// the reference's CPU tables/tmp files and never change the output.
#include <fcntl.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <atomic>

#include <chrono>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "grlbwt_hip.h"

struct arguments {
    std::string input_file, output_file, tmp_dir = "/tmp";
    size_t n_threads = 1;
    int b_f_r = 1;
    float hbuff_frac = 0.15f;
    bool ver = false;
    int alph_bytes = 1;
    int device = 0;
    bool rev_comp = false;                  // -R (main.cpp:17,75): also the DNA reverse complements, FASTA/Q inputs
    bool fastx = false;                     // --fastx: TEXT holds FASTA/Q records (the reference's own conversion, main.cpp:117-136, is switched off upstream)
    int gpus = 1;                           // --gpus N: one process per GPU, record shards, RCCL (collection-level mode)
    std::string version = "v1.0.1 alpha";   // main.cpp:20 (reference version string)
};

static void usage(const char *prog) {
    std::cout << "Repetition-aware BWT construction (MI355X / HIP engine)\n"
              << "Usage: " << prog << " [OPTIONS] TEXT\n\n"
              << "Positionals:\n"
              << "  TEXT                   Input file in one-string-per-line format\n\n"
              << "Options:\n"
              << "  -h,--help              Print this help message and exit\n"
              << "  -o,--output-file       Output file\n"
              << "  -a,--alphabet          Number of bytes for the alphabet (def. 1)\n"
              << "  -t,--threads           Maximum number of working threads\n"
              << "  -f,--hbuff             Hashing step will use at most INPUT_SIZE*f bytes. O means no limit (def. 0.5)\n"
              << "  -b,--run-len-bytes     Max. number of bytes to encode the run lengths in the recursive BWTs (def. 1)\n"
              << "  -T,--tmp               Temporary folder (def. /tmp/grl.bwt.xxxx)\n"
              << "  -v,--version           Print the software version and exit\n"
              << "  --fastx                TEXT is a FASTA/Q file (optionally gzip): its records become the strings.  Without this\n"
              << "                         flag TEXT is always taken as one-string-per-line cells, as the reference does\n"
              << "  -R,--rev-comp          Also consider the DNA reverse complements of the strings in TEXT (implies --fastx)\n"
              << "  -g,--gpu               HIP device ordinal (def. 0; with --gpus: the first of N consecutive devices)\n"
              << "  --gpus                 Number of GPUs: the collection is sharded by record, one process per GPU,\n"
              << "                         exchanges over RCCL; the output does not depend on it (def. 1)\n";
}
[[noreturn]] static void fail(int code, const std::string &msg) {
    std::cerr << msg << "\nRun with --help for more information.\n";
    std::exit(code);
}
static bool is_dir(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode); }
static bool is_file(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode); }

// the reference's report_time (external/cdt/include/utils.h:109-126): same wording, same padding argument
static void report_ms(double seconds, int pad) {
    long long total = (long long)(seconds * 1000.0 + 0.5);
    long long h = total / 3600000, m = (total / 60000) % 60, s = (total / 1000) % 60, ms = total % 1000;
    std::printf("%*s", pad, "");
    if (h > 0) std::printf("Elapsed time (hh:mm:ss.ms): %02lld:%02lld:%02lld.%lld\n", h, m, s, ms);
    else if (m > 0) std::printf("Elapsed time (mm:ss.ms): %02lld:%02lld.%lld\n", m, s, ms);
    else if (s > 0) std::printf("Elapsed time (ss.ms): %02lld.%lld\n", s, ms);
    else std::printf("Elapsed time (ms): %lld\n", ms);
    std::fflush(stdout);
}
static void report_time(std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b, int pad) {
    report_ms(std::chrono::duration<double>(b - a).count(), pad);
}

// ---- --gpus N: the collection-level mode from the command line -------------------------------------------------------
// The parent cuts the file into N record shards (byte ranges ending on a separator) and starts one process per GPU BEFORE
// anything touches the HIP runtime; rank 0 makes the RCCL id and publishes it through a shared page; every rank loads its
// shard, joins the communicator (grlbwt_rccl_comm_create) and runs grlbwt_dist_build; rank 0 reports.  The image stays in parts
// on the ranks that induced them (GRLBWT_COMM_KEEP_PARTS): every rank writes its part at its offset of <output>.tmp~<parent pid>
// (grlbwt_result_write_part), the parent renames the file over the output once every rank has ended well, and removes it otherwise.
struct SharedPage {
    std::atomic<int> written[64];           // 1 = this rank's part is in the file, < 0 = it failed
    std::atomic<int> id_ready;
    char id[GRLBWT_RCCL_ID_BYTES];
    std::atomic<int> loaded[64];            // 0 = not yet, 1 = shard loaded, < 0 = the error code of the load
    std::atomic<int> aborted;               // set by the parent when a rank died: the others stop waiting on this page
};
static bool read_cell(int fd, uint64_t idx, int w, uint64_t *out) {
    unsigned char b[8] = {0};
    if (pread(fd, b, (size_t)w, (off_t)(idx * (uint64_t)w)) != (ssize_t)w) return false;
    uint64_t v = 0;
    for (int i = 0; i < w; i++) v |= (uint64_t)b[i] << (8 * i);
    *out = v;
    return true;
}
// first cell index > from whose predecessor is the separator (= start of the next record), or n
static uint64_t next_record_start(int fd, uint64_t from, uint64_t n, int w, uint64_t sep) {
    std::vector<unsigned char> buf((size_t)1 << 20);
    uint64_t i = from > 0 ? from - 1 : 0;
    while (i < n) {
        const uint64_t cells = std::min<uint64_t>(buf.size() / (size_t)w, n - i);
        const ssize_t got = pread(fd, buf.data(), (size_t)(cells * (uint64_t)w), (off_t)(i * (uint64_t)w));
        if (got <= 0) return n;
        for (uint64_t k = 0; k < (uint64_t)got / (uint64_t)w; k++) {
            uint64_t v = 0;
            for (int b = 0; b < w; b++) v |= (uint64_t)buf[k * w + b] << (8 * b);
            if (v == sep) return i + k + 1;
        }
        i += (uint64_t)got / (uint64_t)w;
    }
    return n;
}
static int rank_main(const arguments &args, int rank, int size, uint64_t off_bytes, uint64_t n_bytes, bool idx64, SharedPage *sh, const std::string &tmp_out) {
    const bool root = rank == 0;
    const auto t_start = std::chrono::steady_clock::now();
    grlbwt_ctx *ctx = nullptr;
    // (GRLBWT_CLI_SAME_DEVICE=1: every rank on --gpu's device, for boxes with fewer GPUs than ranks -- if the RCCL build allows it)
    const int dev = std::getenv("GRLBWT_CLI_SAME_DEVICE") ? args.device : args.device + rank;
    // the index width follows the COLLECTION, not the shard: every rank takes the same one (the ranks exchange idx_t arrays)
    int rc = grlbwt_ctx_create(dev, GRLBWT_FLAG_CLASSIC_POOL | (idx64 ? GRLBWT_FLAG_FORCE_IDX64 : 0u), &ctx);   // (classic pool: buffers are handed to RCCL)
    if (rc == GRLBWT_OK) rc = grlbwt_text_load_file_range(ctx, args.input_file.c_str(), off_bytes, n_bytes, args.alph_bytes);
    // every rank learns whether every shard loaded before the first collective (nobody may be left waiting in one)
    sh->loaded[rank].store(rc == GRLBWT_OK ? 1 : (rc < 0 ? rc : -1));
    int worst = 1;
    for (int g = 0; g < size; g++) {
        int v;
        while ((v = sh->loaded[g].load()) == 0) { if (sh->aborted.load()) return 2; usleep(200); }
        if (v < 0 && worst == 1) worst = v;
    }
    if (worst < 0) {
        if (root) {
            if (worst == GRLBWT_EILLFORMED) std::cout << "Error: the file is ill formed" << std::endl;
            else std::cerr << "grlbwt: a rank could not load its shard (" << grlbwt_strerror(worst) << ")" << std::endl;
        }
        if (ctx) grlbwt_ctx_destroy(ctx);
        return worst == GRLBWT_EILLFORMED ? 1 : (worst == GRLBWT_EDEVICE ? 3 : 2);
    }
    const auto t_loaded = std::chrono::steady_clock::now();
    auto die = [&](int code, const char *what) -> int {
        std::cerr << "grlbwt[rank " << rank << "]: " << what << ": " << grlbwt_last_error(ctx) << " (" << grlbwt_strerror(code) << ")" << std::endl;
        return code == GRLBWT_EILLFORMED ? 1 : 2;      // (no destroy: peers may be inside a collective; the process ends)
    };
    if (root) {
        rc = grlbwt_rccl_unique_id(sh->id);
        sh->id_ready.store(rc == GRLBWT_OK ? 1 : -1);
    }
    int ready;
    while ((ready = sh->id_ready.load()) == 0) { if (sh->aborted.load()) return 2; usleep(200); }
    if (ready < 0) { if (root) std::cerr << "grlbwt: RCCL is not available" << std::endl; return 3; }
    grlbwt_comm comm;
    std::memset(&comm, 0, sizeof comm);
    rc = grlbwt_rccl_comm_create(ctx, sh->id, rank, size, &comm);
    if (rc != GRLBWT_OK) return die(rc, "joining the communicator");
    if (root) std::cout << "Parsing the text and inferring the BWT on " << size << " GPUs (record shards, RCCL)" << std::endl;
    comm.flags |= GRLBWT_COMM_KEEP_PARTS;
    rc = grlbwt_dist_build(ctx, &comm);
    if (rc != GRLBWT_OK) {
        if (rc == GRLBWT_EILLFORMED && root) std::cout << "Error: the file is ill formed" << std::endl;
        return die(rc, "collection-level build");
    }
    const auto t_built = std::chrono::steady_clock::now();
    int code = 0;
    rc = grlbwt_result_write_part(ctx, tmp_out.c_str());
    sh->written[rank].store(rc == GRLBWT_OK ? 1 : -1);
    if (rc != GRLBWT_OK) code = die(rc, "writing the output");
    if (root) {
        // the reference's per-round / per-level statistics (exact_par_phase.cpp:484-488, exact_ind_phase.cpp:372-378)
        grlbwt_stats st;
        grlbwt_get_stats(ctx, &st);
        std::cout << "Stats: " << std::endl;
        std::cout << "  Smallest symbol               : " << st.min_sym << std::endl;
        std::cout << "  Greatest symbol               : " << st.max_sym << std::endl;
        std::cout << "  Number of symbols in the file : " << st.n_syms << std::endl;
        std::cout << "  Number of strings             : " << st.n_strings << std::endl;
        std::cout << "Parsing the text:    " << std::endl;
        int rounds = 0;
        grlbwt_round_info ri;
        while (grlbwt_round_info_get(ctx, rounds, &ri) == GRLBWT_OK) {
            std::cout << "  Parsing round " << ++rounds << std::endl;
            std::cout << "    Stats:" << std::endl;
            std::cout << "      Parsing phrases:                  " << ri.n_phrases << std::endl;
            std::cout << "      Number of symbols in the phrases: " << ri.dict_syms << std::endl;
            std::cout << "      Number of unsolved BWT blocks:    " << ri.n_metasyms << std::endl;
            std::cout << "      Parse size:                       " << ri.parse_size << std::endl;
        }
        std::cout << "Inferring the BWT" << std::endl;
        grlbwt_counters c;
        grlbwt_get_counters(ctx, &c);
        std::cout << "  Stage seconds of rank 0: dictionary of LMS phrases " << c.t_classify + c.t_hash << ", sorting + preliminary BWT " << c.t_dict_sort
                  << ", compressing " << c.t_dict_groups << ", parse " << c.t_emit << ", induced symbols " << c.t_ind_expand << ", induction "
                  << c.t_ind_split << ", assembling " << c.t_ind_assemble << std::endl;
        bool all_written = true;
        for (int g = 0; g < size && all_written; g++) {
            int v;
            while ((v = sh->written[g].load()) == 0) { if (sh->aborted.load()) { v = -1; break; } usleep(200); }
            if (v < 0) all_written = false;
        }
        if (!all_written) { if (code == 0) code = 2; }
        else {
            const auto t_written = std::chrono::steady_clock::now();
            // (the parts sit in the temporary file: the parent says "stored" once it has moved the file into place)
            auto sec = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
            const double tr = sec(t_start, t_loaded), tb = sec(t_loaded, t_built), tw = sec(t_built, t_written), tt = sec(t_start, t_written);
            std::printf("grlbwt-timing: read+upload %.3f s, build %.3f s, write %.3f s, total %.3f s, %.1f MB/s (input bytes / total), %d GPUs\n", tr, tb, tw, tt,
                        (double)st.n_syms * args.alph_bytes / 1e6 / (tt > 0 ? tt : 1e-9), size);
        }
    }
    grlbwt_rccl_comm_destroy(&comm);
    grlbwt_ctx_destroy(ctx);
    return code;
}
static int run_multi_gpu(const arguments &args) {
    const int N = args.gpus, w = args.alph_bytes;
    int fd = open(args.input_file.c_str(), O_RDONLY);
    struct stat st;
    if (fd < 0 || fstat(fd, &st) != 0) fail(105, "TEXT: File does not exist: " + args.input_file);
    const uint64_t bytes = (uint64_t)st.st_size;
    if (bytes == 0 || bytes % (uint64_t)w) { std::cout << "Error: the file is ill formed" << std::endl; return 1; }
    const uint64_t n = bytes / (uint64_t)w;
    uint64_t sep = 0;
    if (!read_cell(fd, n - 1, w, &sep)) fail(105, "TEXT: cannot read " + args.input_file);
    // record shards: cut g at the first record start at or after cell g*n/N (the separator is the file's last cell)
    std::vector<uint64_t> cut(N + 1, 0);
    cut[N] = n;
    for (int g = 1; g < N; g++) cut[g] = std::max(cut[g - 1], next_record_start(fd, (uint64_t)g * (n / (uint64_t)N), n, w, sep));
    close(fd);
    for (int g = 0; g < N; g++)
        if (cut[g + 1] <= cut[g]) fail(105, "--gpus: the collection has too few strings for " + std::to_string(N) + " record shards");
    SharedPage *sh = (SharedPage *)mmap(nullptr, sizeof(SharedPage), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    if (sh == MAP_FAILED) fail(2, "grlbwt: cannot map the rendezvous page");
    new (sh) SharedPage();
    sh->id_ready.store(0);
    sh->aborted.store(0);
    for (int g = 0; g < 64; g++) { sh->loaded[g].store(0); sh->written[g].store(0); }
    const std::string tmp_out = args.output_file + ".tmp~" + std::to_string((long)getpid());
    unlink(tmp_out.c_str());
    const bool idx64 = n >= 0xFFFFFF00ull;       // (include/grlbwt_hip.h: 64-bit positions from 2^32 - 256 cells on)
    std::cout << std::flush;
    std::vector<pid_t> kids;
    for (int g = 0; g < N; g++) {
        pid_t pid = fork();                  // (this process has not touched the GPU: the children initialise HIP themselves)
        if (pid < 0) fail(2, "grlbwt: fork failed");
        if (pid == 0) {
            int rc = rank_main(args, g, N, cut[g] * (uint64_t)w, (cut[g + 1] - cut[g]) * (uint64_t)w, idx64, sh, tmp_out);
            std::cout << std::flush;
            _exit(rc);
        }
        kids.push_back(pid);
    }
    // Reap in completion order.  A rank that ends abnormally (non-zero exit, a signal) may leave its peers inside a
    // collective that will never complete: the page tells the ones still at the rendezvous, the others are killed.
    // The exit code reported is the FIRST non-zero one (later ones are usually consequences: a peer killed here ends with 2).
    int first_bad = 0;
    size_t left = kids.size();
    bool killed = false;
    auto reap = [&](pid_t pid, int status) {
        bool mine = false;
        for (pid_t &k : kids) if (k == pid) { k = -1; mine = true; }
        if (!mine) return false;
        left--;
        const int code = WIFEXITED(status) ? WEXITSTATUS(status) : 2;
        if (code != 0 && first_bad == 0) first_bad = code;
        return code != 0;
    };
    while (left > 0) {
        int status = 0;
        const pid_t pid = waitpid(-1, &status, 0);
        if (pid < 0) {
            if (errno == EINTR) continue;        // (a signal delivered to the parent: the ranks are still running)
            break;                               // ECHILD: nothing left to wait for
        }
        if (reap(pid, status) && !killed) {
            sh->aborted.store(1);
            // ranks that fail together (an agreed error) print their message, tear their arenas down and leave by themselves:
            // give them up to 2 s (polling), then kill what is still inside a collective that will never complete
            for (int waited_ms = 0; left > 0 && waited_ms < 2000; ) {
                const pid_t q = waitpid(-1, &status, WNOHANG);
                if (q > 0) { reap(q, status); continue; }
                if (q < 0 && errno != EINTR) break;
                usleep(20000);
                waited_ms += 20;
            }
            for (pid_t k : kids) if (k > 0) kill(k, SIGKILL);
            killed = true;
        }
    }
    int worst = first_bad;
    // the ranks wrote their parts into the temporary: it becomes the output only when every one of them ended well
    if (worst == 0 && rename(tmp_out.c_str(), args.output_file.c_str()) != 0) {
        std::cerr << "grlbwt: cannot move the output into place: " << args.output_file << std::endl;
        worst = 2;
    }
    if (worst != 0) unlink(tmp_out.c_str());
    else std::cout << "The resulting BCR BWT was stored in " << args.output_file << std::endl;   // grl_bwt.hpp:78 -- true from here on
    return worst;
}

int main(int argc, char **argv) {
    arguments args;
    bool have_text = false;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto need = [&](const char *name) -> std::string {
            if (i + 1 >= argc) fail(114, std::string(name) + ": 1 required");
            return argv[++i];
        };
        if (a == "-h" || a == "--help") { usage(argv[0]); return 0; }
        else if (a == "-v" || a == "--version") args.ver = true;
        else if (a == "-o" || a == "--output-file") args.output_file = need("--output-file");
        else if (a == "-a" || a == "--alphabet") {
            std::string v = need("--alphabet");
            if (!(v == "1" || v == "2" || v == "4" || v == "8"))
                fail(105, "--alphabet: " + v + " is not a valid number of bytes for a native integer type");
            args.alph_bytes = std::atoi(v.c_str());
        } else if (a == "-t" || a == "--threads") args.n_threads = std::strtoul(need("--threads").c_str(), nullptr, 10);
        else if (a == "-f" || a == "--hbuff") {
            args.hbuff_frac = std::strtof(need("--hbuff").c_str(), nullptr);
            if (args.hbuff_frac < 0.f || args.hbuff_frac > 1.f) fail(105, "--hbuff: Value not in range 0 to 1");
        } else if (a == "-b" || a == "--run-len-bytes") {
            args.b_f_r = std::atoi(need("--run-len-bytes").c_str());
            if (args.b_f_r < 0 || args.b_f_r > 5) fail(105, "--run-len-bytes: Value not in range 0 to 5");
        } else if (a == "-T" || a == "--tmp") {
            args.tmp_dir = need("--tmp");
            if (!is_dir(args.tmp_dir)) fail(105, "--tmp: Directory does not exist: " + args.tmp_dir);
        } else if (a == "-g" || a == "--gpu") args.device = std::atoi(need("--gpu").c_str());
        else if (a == "-R" || a == "--rev-comp") { args.rev_comp = true; args.fastx = true; }
        else if (a == "--fastx") args.fastx = true;
        else if (a == "--plain") args.fastx = false;                              // (the default: kept so that scripts can say it)
        else if (a == "--gpus") {
            args.gpus = std::atoi(need("--gpus").c_str());
            if (args.gpus < 1 || args.gpus > 64) fail(105, "--gpus: Value not in range 1 to 64");
        }
        else if (!a.empty() && a[0] == '-' && a.size() > 1) fail(109, "The following argument was not expected: " + a);
        else if (!have_text) { args.input_file = a; have_text = true; }
        else fail(109, "The following argument was not expected: " + a);
    }
    if (!have_text) fail(106, "TEXT is required");                                  // main.cpp:53 ->required()
    if (!is_file(args.input_file)) fail(105, "TEXT: File does not exist: " + args.input_file);
    if (args.ver) { std::cout << args.version << std::endl; return 0; }              // main.cpp:106-109

    std::cout << "Input file:       " << args.input_file << std::endl;               // main.cpp:111
    if (args.output_file.empty()) args.output_file = std::filesystem::path(args.input_file).filename();
    args.output_file = std::filesystem::path(args.output_file).replace_extension(".rl_bwt");   // main.cpp:112-113
    std::cout << (args.alph_bytes > 1 ? "Alphabet type:    integer" : "Alphabet type:    byte") << std::endl;
    std::cout << "Temporary folder: (none: all levels stay resident in HBM)" << std::endl;
    std::cout << "BWT type:         BCR exact" << std::endl;
    // FASTA/Q conversion is OPT-IN (--fastx / -R).  The reference has its is_fastx branch commented out (main.cpp:117-136) and
    // runs collection_stats on the raw file whatever it starts with, so a one-string-per-line collection whose first byte
    // happens to be '>' or '@' (quoted mail, handles) must give the BWT of exactly those bytes here too.
    // is_fastx (external/cdt/lib/utils.cpp:13-30): first (decompressed) byte '>' or '@'
    int fastx = 0, is_gz = 0;
    int looks_fastx = 0;
    if (grlbwt_fastx_probe(args.input_file.c_str(), &looks_fastx, &is_gz) != GRLBWT_OK) looks_fastx = 0;
    if (args.fastx) {
        if (!looks_fastx) fail(105, "--fastx: TEXT is not in FASTA/Q format (its first byte is neither '>' nor '@')");
        fastx = 1;
        if (args.alph_bytes != 1) fail(105, "--alphabet: a FASTA/Q input has a byte alphabet");
        if (args.gpus > 1) fail(105, "--gpus: FASTA/Q inputs are converted on one GPU; convert first or use one GPU");
        std::cout << "The input is in FASTA/Q format" << (is_gz ? " (gzip)" : "") << std::endl;          // main.cpp:120
    } else if (looks_fastx && args.alph_bytes == 1)
        std::cerr << "grlbwt: note: TEXT starts like a FASTA/Q file but is read as one-string-per-line cells (the reference does the same); "
                     "pass --fastx to convert its records" << std::endl;
    // GRLBWT_CLI_FORCE_RCCL=1: the collection-level path also with one GPU (that is all a single-GPU box can test)
    if (args.gpus > 1 || std::getenv("GRLBWT_CLI_FORCE_RCCL")) return run_multi_gpu(args);

    const auto t_start = std::chrono::steady_clock::now();
    grlbwt_ctx *ctx = nullptr;
    int rc = grlbwt_ctx_create(args.device, 0, &ctx);
    if (rc != GRLBWT_OK) {
        std::cerr << "grlbwt: no usable HIP device (" << grlbwt_strerror(rc) << "); this build has no CPU path" << std::endl;
        return 3;
    }
    auto die = [&](int code) -> int {
        if (code == GRLBWT_EILLFORMED) { std::cout << "Error: the file is ill formed" << std::endl; grlbwt_ctx_destroy(ctx); return 1; }
        std::cerr << "grlbwt: " << grlbwt_last_error(ctx) << " (" << grlbwt_strerror(code) << ")" << std::endl;
        grlbwt_ctx_destroy(ctx);
        return 2;
    };

    // the file goes to HBM through pinned staging buffers, chunk k+1 read while chunk k is copied; the symbol
    // statistics of collection_stats are taken on the device behind the copies (the reference streams the file
    // through i_file_stream twice: once for the statistics, once for the first parsing round)
    std::cout << "Reading the file" << std::endl;                                      // exact_par_phase.cpp:288
    if (fastx) {
        // main.cpp:118-124 (switched off upstream): FASTA/Q records -> one string per line, here on the device and in HBM
        // instead of a temporary plain file
        uint64_t n_strings = 0;
        rc = grlbwt_text_load_fastx(ctx, args.input_file.c_str(), args.rev_comp ? GRLBWT_FASTX_REVCOMP : 0, &n_strings);
        if (rc == GRLBWT_ENOTDNA) { std::cerr << grlbwt_last_error(ctx) << std::endl; grlbwt_ctx_destroy(ctx); return 1; }   // fastx_handler.cpp:30-33
    } else rc = grlbwt_text_load_file(ctx, args.input_file.c_str(), args.alph_bytes);
    if (rc != GRLBWT_OK) return die(rc);
    const auto t_loaded = std::chrono::steady_clock::now();
    grlbwt_stats st;
    grlbwt_get_stats(ctx, &st);
    std::cout << "Stats: " << std::endl;                                               // exact_par_phase.cpp:290-294
    std::cout << "  Smallest symbol               : " << st.min_sym << std::endl;
    std::cout << "  Greatest symbol               : " << st.max_sym << std::endl;
    std::cout << "  Number of symbols in the file : " << st.n_syms << std::endl;
    std::cout << "  Number of strings             : " << st.n_strings << std::endl;

    // Stage labels of par_round (exact_par_phase.cpp:380,410,111,124,428,452) with the device time of the kernels
    // that replace each stage (stage clocks of the engine, grlbwt_get_counters); the stages of one round run back
    // to back on the GPU, so the labels are printed once the round is done.
    grlbwt_counters c0, c1;
    std::memset(&c0, 0, sizeof c0);
    std::cout << "Parsing the text:    " << std::endl;                                 // exact_par_phase.cpp:301
    int done = 0, iter = 1;
    while (!done) {
        std::cout << "  Parsing round " << iter++ << std::endl;                        // :327,340
        auto t0 = std::chrono::steady_clock::now();
        grlbwt_round_info ri;
        rc = grlbwt_parse_round(ctx, &ri, &done);
        if (rc != GRLBWT_OK) return die(rc);
        grlbwt_get_counters(ctx, &c1);
        std::cout << "    Computing the dictionary of LMS phrases" << std::flush;     // :380  (LMS breaks + phrase hashing)
        report_ms((c1.t_classify - c0.t_classify) + (c1.t_hash - c0.t_hash), 22);
        std::cout << "    Compacting the dictionary" << std::flush;                   // :410  (+ :111 sorting, pre-BWT)
        report_ms(0.0, 36);
        std::cout << "    Sorting the dictionary and constructing the preliminary BWT" << std::flush;   // :111
        report_ms(c1.t_dict_sort - c0.t_dict_sort, 2);
        std::cout << "    Compressing the dictionary" << std::flush;                  // :124  (groups, grammar)
        report_ms(c1.t_dict_groups - c0.t_dict_groups, 35);
        std::cout << "    Assigning metasymbols to the LMS phrases" << std::flush;    // :428
        report_ms(0.0, 21);
        std::cout << "    Creating the parse of the text" << std::flush;              // :452
        report_ms(c1.t_emit - c0.t_emit, 31);
        c0 = c1;
        std::cout << "    Stats:" << std::endl;                                        // :484-488
        std::cout << "      Parsing phrases:                  " << ri.n_phrases << std::endl;
        std::cout << "      Number of symbols in the phrases: " << ri.dict_syms << std::endl;
        std::cout << "      Number of unsolved BWT blocks:    " << ri.n_metasyms << std::endl;
        std::cout << "      Parse size:                       " << ri.parse_size << std::endl;
        report_time(t0, std::chrono::steady_clock::now(), 4);                          // :332,356
    }

    std::cout << "Inferring the BWT" << std::endl;                                     // exact_ind_phase.cpp:679
    std::cout << "  Computing the deepest recursive BWT" << std::endl;                 // :605
    rc = grlbwt_induce_first(ctx);
    if (rc != GRLBWT_OK) return die(rc);
    int level = iter - 1;
    while (level > 0) {
        std::cout << "  Inducing the BWT for parse " << level << std::endl;            // :683
        auto t0 = std::chrono::steady_clock::now();
        grlbwt_level_info li;
        rc = grlbwt_induce_level(ctx, &level, &li);
        if (rc != GRLBWT_OK) return die(rc);
        grlbwt_get_counters(ctx, &c1);
        std::cout << "    Computing the number of induced symbols" << std::flush;     // :121
        report_ms(c1.t_ind_expand - c0.t_ind_expand, 9);
        std::cout << "    Performing the induction from the previous BWT" << std::flush;   // :141
        report_ms(c1.t_ind_split - c0.t_ind_split, 2);
        std::cout << "    Assembling the new BWT" << std::flush;                      // :270
        report_ms(c1.t_ind_assemble - c0.t_ind_assemble, 26);
        c0 = c1;
        std::cout << "    Stats:       " << std::endl;                                 // :372-378
        std::cout << "      BWT size (n):                        " << li.n << std::endl;
        std::cout << "      Number of runs (r):                  " << li.n_runs << std::endl;
        std::cout << "      n/r:                                 " << double(li.n) / double(li.n_runs ? li.n_runs : 1) << std::endl;
        if (level == 0) {
            std::cout << "      Bytes per run symbol:                " << st.sb << std::endl;
            std::cout << "      Bytes per run length:                " << st.fb << std::endl;
        }
        std::cout << "      Bytes per induced run length:        " << args.b_f_r << " (fixed by CLI)" << std::endl;
        std::cout << "        Induced runs with length overflow: 0" << std::endl;
        report_time(t0, std::chrono::steady_clock::now(), 4);                          // :691
    }
    const auto t_built = std::chrono::steady_clock::now();
    rc = grlbwt_result_write_file(ctx, args.output_file.c_str());
    if (rc != GRLBWT_OK) return die(rc);
    const auto t_written = std::chrono::steady_clock::now();
    std::cout << "The resulting BCR BWT was stored in " << args.output_file << std::endl;   // grl_bwt.hpp:78
    // one machine-readable line for harnesses: wall seconds of the three legs of the run and the input rate
    {
        auto sec = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
        const double tr = sec(t_start, t_loaded), tb = sec(t_loaded, t_built), tw = sec(t_built, t_written), tt = sec(t_start, t_written);
        std::printf("grlbwt-timing: read+upload %.3f s, build %.3f s, write %.3f s, total %.3f s, %.1f MB/s (input bytes / total)\n", tr, tb, tw, tt,
                    (double)st.n_syms * args.alph_bytes / 1e6 / (tt > 0 ? tt : 1e-9));
    }
    // The output is complete and closed: leave without tearing down the context (returning ~70 GB of device mappings page by page
    // and the runtime's own exit handlers cost the 10 GB run 0.05-0.1 s of wall time; the driver reclaims a process's device
    // memory when it ends either way).  GRLBWT_CLI_TEARDOWN=1 keeps the orderly way (leak checkers).
    std::cout << std::flush;
    std::fflush(stdout);
    std::fflush(stderr);
    if (!std::getenv("GRLBWT_CLI_TEARDOWN")) _exit(0);
    grlbwt_ctx_destroy(ctx);
    return 0;
}
