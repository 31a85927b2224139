// grlbwt -- command line of the MI355X-native BCR-BWT engine.
//
// Drop-in for the reference CLI (ddiazdom/grlBWT main.cpp:43-154): same flags, same
// output naming (main.cpp:112-113), same stdout labels, same exit behaviour
// (0 ok, 1 ill-formed input, 105 validation error, 106 missing TEXT), over the
// C-ABI of include/grlbwt_hip.h.  Own argument parser (CLI11 is third-party).
// -t/-f/-b/-T are accepted and validated as in the reference; they are tuning knobs of
// the reference's CPU tables/tmp files and never change the output.
#include <sys/stat.h>

#include <chrono>
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
              << "  -g,--gpu               HIP device ordinal (def. 0)\n";
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
    rc = grlbwt_text_load_file(ctx, args.input_file.c_str(), args.alph_bytes);
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
    grlbwt_ctx_destroy(ctx);
    return 0;
}
