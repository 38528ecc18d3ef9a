// pfbwt-f_amd/host/include/file_wrappers.hpp -- the container template arguments of PrefixFreeBWT
// (reference include/file_wrappers.hpp:35-222).  The engine keeps its working arrays (gSA, class heads,
// ...) in HBM, so the memory-mapped variants (-m) reduce to "load the file"; all four names are kept so
// that `run_pfbwt<MMapFileSource, MMapFileSink>` and `run_pfbwt<VecFileSource, VecFileSinkPrivate>`
// (src/pfbwt-f.cpp:359-365) compile unchanged.
#ifndef PFBWTF_FILE_WRAPPERS_HPP
#define PFBWTF_FILE_WRAPPERS_HPP
#include <string>
#include <vector>
#include "pfbwtf_common.hpp"

template <typename T, typename... Rest> class VecFileSource : public std::vector<T> {
  public:
    VecFileSource() = default;
    explicit VecFileSource(const std::string &path) : std::vector<T>(pfbwtf::read_vec<T>(path)) {}
    void init_file(const std::string &, size_t n) { this->assign(n, T()); }
};
template <typename T, typename... Rest> using VecFileSinkPrivate = VecFileSource<T, Rest...>;
template <typename T, typename... Rest> using VecFileSinkShared = VecFileSource<T, Rest...>;
template <typename T, typename... Rest> using MMapFileSource = VecFileSource<T, Rest...>;
template <typename T, typename... Rest> using MMapFileSink = VecFileSource<T, Rest...>;
#endif
