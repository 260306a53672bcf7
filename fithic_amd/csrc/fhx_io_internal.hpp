// fhx_io_internal.hpp - pieces of the host reader shared with the device-side ingest (not part of the C ABI)
#pragma once
#include <cstdint>
#include <string>
#include <vector>

// an inflated file (fhx_host_inflate): the text as pieces in file order, one per inflating thread
struct fhx_text {
    std::string path;
    std::vector<std::string> pieces;
    int64_t bytes = 0;
    double seconds[2] = {0, 0};                 // file read, inflate
    std::string error;
};

namespace fhx {

// path -> the inflated text as pieces in file order (see fhx_io.cpp); returns an FHX_* code and, on failure, the message
int io_inflate_file(const char* path, int n_threads, std::vector<std::string>& pieces, std::string& error, double* seconds);

}  // namespace fhx
