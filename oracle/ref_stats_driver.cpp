// ref_stats_driver.cpp -- TEST INFRASTRUCTURE.  A main() around the pieces of the reference's BUILDER path that compile
// without SDSL-lite: collection_stats<T> (external/cdt/lib/utils.cpp:100-189 -- row a1 of SURVEY 8a, the inputs of the
// final header), sym_width (external/cdt/lib/cdt_common.cpp:7-10) with INT_CEIL (external/cdt/include/macros.h:8) as
// exact_ind_phase.cpp:274-276 combines them at level 0 (row a17), and bwt_buff_writer::push_back / inc_freq_last /
// close (include/bwt_io.h:448-550 -- row a16) driven with the emit idiom of pass C (exact_ind_phase.cpp:338-343,
// 352-357: "same symbol as the last run -> inc_freq_last, else push_back").  oracle/Makefile (target `ref`) compiles THIS
// file with the reference's sources where they lie; nothing of the reference is copied.
//   ref_stats IN W            -> "n_strings N" "longest_string L" "min_sym" "max_sym" "n_syms" "max_sym_freq" "sb" "fb"
//   ref_stats IN W RUNS OUT   -> also writes the (sym len) pairs of RUNS (text, one pair per line, not necessarily
//                                maximal runs) to OUT through the reference's writer with the header widths above
#include <algorithm>
#include <cstdint>
#include <fstream>
#include <iostream>
#include <limits>
#include <string>

#include "bwt_io.h"
#include "cdt_common.hpp"
#include "macros.h"
#include "utils.h"

int main(int argc, char **argv) {
    if (argc != 3 && argc != 5) { std::cerr << "usage: ref_stats IN W [RUNS OUT]" << std::endl; return 2; }
    std::string in = argv[1];
    const int w = std::stoi(argv[2]);
    str_collection c;
    if (w == 1) c = collection_stats<uint8_t>(in);
    else if (w == 2) c = collection_stats<uint16_t>(in);
    else if (w == 4) c = collection_stats<uint32_t>(in);
    else if (w == 8) c = collection_stats<uint64_t>(in);
    else return 2;
    // level 0 of infer_lvl_bwt (exact_ind_phase.cpp:274-276): dict.alphabet = max_sym + 1 (exact_par_phase.hpp:108) + 3
    // (exact_par_phase.cpp:19), dict.prev_alphabet = 0, dict.max_sym_freq = the collection's (exact_par_phase.cpp:315,413)
    const size_t alphabet = c.max_sym + 1 + 3, prev_alphabet = 0;
    const size_t sb = INT_CEIL(sym_width(std::max(alphabet, prev_alphabet)), 8);
    const size_t fb = INT_CEIL(sym_width(c.max_sym_freq), 8);
    std::cout << "n_strings " << c.n_strings << "\nlongest_string " << c.longest_string << "\nmin_sym " << c.min_sym << "\nmax_sym "
              << c.max_sym << "\nn_syms " << c.n_syms << "\nmax_sym_freq " << c.max_sym_freq << "\nsb " << sb << "\nfb " << fb << std::endl;
    if (argc == 5) {
        std::ifstream runs(argv[3]);
        bwt_buff_writer out(argv[4], std::ios::out, (uint8_t)sb, (uint8_t)fb);
        size_t sym, len;
        while (runs >> sym >> len) {
            if (out.size() > 0 && out.last_sym() == sym) out.inc_freq_last(len);
            else out.push_back(sym, len);
        }
        out.close();
        std::cout << "runs " << out.size() << std::endl;
    }
    return 0;
}
