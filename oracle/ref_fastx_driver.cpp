// ref_fastx_driver.cpp -- TEST INFRASTRUCTURE.  A main() around the reference's own FASTA/Q converter: the reference
// switched the call off in its main.cpp (:118-136, "this option is not implemented yet") but still ships the code.
// oracle/Makefile (target `ref`) compiles THIS file together with the reference's sources where they lie
// (external/bioparsers/lib/fastx_handler.cpp, dna_string.cpp, external/cdt/lib/utils.cpp; zlib) into
// oracle/_ref/fastx2plain.  Usage: fastx2plain IN OUT RC(0|1)   -> prints "is_fastx <0|1>" and "n_strings <n>".
#include <iostream>
#include <string>

#include "fastx_handler.h"
#include "utils.h"

int main(int argc, char **argv) {
    if (argc != 4) { std::cerr << "usage: fastx2plain IN OUT RC" << std::endl; return 2; }
    const std::string in = argv[1], out = argv[2];
    const bool rc = std::string(argv[3]) == "1";
    std::cout << "is_fastx " << (is_fastx(in) ? 1 : 0) << std::endl;
    str_collection c = fastx2plain_format(in, out, rc, '\n');
    std::cout << "n_strings " << c.n_strings << std::endl;
    return 0;
}
