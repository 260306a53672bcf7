// fhx_io_internal.hpp - pieces of the host reader shared with the device-side ingest (not part of the C ABI)
#pragma once
#include <string>
#include <vector>

namespace fhx {

// path -> the inflated text as pieces in file order (see fhx_io.cpp); returns an FHX_* code and, on failure, the message
int io_inflate_file(const char* path, int n_threads, std::vector<std::string>& pieces, std::string& error, double* seconds);

}  // namespace fhx
