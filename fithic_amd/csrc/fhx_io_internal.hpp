// fhx_io_internal.hpp - pieces of the host reader shared with the device-side ingest (not part of the C ABI)
#pragma once
#include <sys/mman.h>

#include <cstdint>
#include <string>
#include <utility>
#include <vector>

namespace fhx {

// One piece of an inflated text.  A piece of known size is an anonymous mapping that asks for transparent huge pages: the
// inflating threads touch 4.6 GB for the first time, and with 4 KB pages those 1.1 M page faults (16 threads on one address
// space) took as long as the inflate itself - 0.40 s against 0.02 s with 2 MB pages on the MI355X box - and munmap 0.5 s
// against 0.23 s.  A piece of unknown size (one plain gzip stream) grows as a std::string.
class TextPiece {
  public:
    TextPiece() = default;
    TextPiece(const TextPiece&) = delete;
    TextPiece& operator=(const TextPiece&) = delete;
    TextPiece(TextPiece&& o) noexcept { *this = std::move(o); }
    TextPiece& operator=(TextPiece&& o) noexcept {
        if (this != &o) {
            release();
            map_ = o.map_;
            n_ = o.n_;
            grown = std::move(o.grown);
            o.map_ = nullptr;
            o.n_ = 0;
        }
        return *this;
    }
    ~TextPiece() { release(); }
    bool allocate(size_t n) {
        release();
        if (n == 0) return true;
        void* m = ::mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (m == MAP_FAILED) return false;
        (void)::madvise(m, n, MADV_HUGEPAGE);
        map_ = (char*)m;
        n_ = n;
        return true;
    }
    void release() {
        if (map_) ::munmap(map_, n_);
        map_ = nullptr;
        n_ = 0;
        std::string().swap(grown);
    }
    char* data() { return map_ ? map_ : &grown[0]; }
    const char* data() const { return map_ ? map_ : grown.data(); }
    size_t size() const { return map_ ? n_ : grown.size(); }
    std::string grown;

  private:
    char* map_ = nullptr;
    size_t n_ = 0;
};

// a file's bytes: a read-only mapping where the file can be mapped, a buffer otherwise
struct FileBytes {
    const unsigned char* p = nullptr;
    size_t n = 0;
    void* map = nullptr;
    std::vector<unsigned char> owned;
    const unsigned char* data() const { return p; }
    size_t size() const { return n; }
    FileBytes() = default;
    FileBytes(const FileBytes&) = delete;
    FileBytes& operator=(const FileBytes&) = delete;
    ~FileBytes();
};
// FHX_OK, or the code and message for a file that cannot be opened / read / is not gzip
int io_read_file(const char* path, FileBytes& out, std::string& error);

// position and compressed size of a gzip member in the file, uncompressed size (ISIZE)
struct GzMember {
    size_t off, size, isize;
};
// the members of a file whose members all carry their compressed size ("FH" of this library's writers, "BC" of BGZF); false:
// some member has no size field (plain gzip) or the chain does not end exactly at the end of the file
bool io_scan_members(const unsigned char* d, size_t n, std::vector<GzMember>& out);

// One plain gzip stream inflated on all cores (fhx_gunzip.cpp: block starts found by their headers, chunks decoded with the
// unknown window as 16-bit symbols, windows resolved down the chain, CRC-32 and ISIZE checked).  false + why: not done, use zlib.
bool io_parallel_gunzip(const unsigned char* gz, size_t n, int n_threads, std::vector<TextPiece>& pieces, std::string& why);

// path -> the inflated text as pieces in file order (see fhx_io.cpp); returns an FHX_* code and, on failure, the message
int io_inflate_file(const char* path, int n_threads, std::vector<TextPiece>& pieces, std::string& error, double* seconds);

}  // namespace fhx

// an inflated file (fhx_host_inflate): the text as pieces in file order, one per inflating thread
struct fhx_text {
    std::string path;
    std::vector<fhx::TextPiece> pieces;
    int64_t bytes = 0;
    double seconds[2] = {0, 0};                 // file read, inflate
    std::string error;
};
