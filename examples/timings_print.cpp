// timings_print.cpp — prints g4s::Timings::print / reg_print for stage times given on the command line (milliseconds: create spmm convert
// order export_csr destroy total, then total_flop). tests/test_cpp_host.py diffs the bytes against the layout of mm/src/Timings.cpp:36-65.
#include <cstdlib>
#include "g4s/csr.hpp"

int main(int argc, char **argv)
{
    if (argc != 9) return 2;
    g4s::Timings t;
    t.create = std::atof(argv[1]); t.spmm = std::atof(argv[2]); t.convert = std::atof(argv[3]); t.order = std::atof(argv[4]);
    t.export_csr = std::atof(argv[5]); t.destroy = std::atof(argv[6]); t.total = std::atof(argv[7]);
    t.print(std::atof(argv[8]));
    t.reg_print(std::atof(argv[8]));
    return 0;
}
