// Minimal stand-in for the part of boost::filesystem the `integrate` / `tsdf2mesh` programs use, on top
// of std::filesystem (C++17).  [Boost-recall: v3 semantics of extension()/basename() = path::extension /
// path::stem]
#pragma once

#include <filesystem>
#include <string>

namespace boost {
namespace filesystem {

class path {
 public:
  path() {}
  path(const std::string &s) : p_(s) {}
  path(const char *s) : p_(s) {}
  path(const std::filesystem::path &p) : p_(p) {}
  std::string string() const { return p_.string(); }
  const std::filesystem::path &std_path() const { return p_; }

 private:
  std::filesystem::path p_;
};

class directory_entry {
 public:
  directory_entry() {}
  explicit directory_entry(const std::filesystem::path &p) : p_(p) {}
  const filesystem::path &path() const { return p_; }

 private:
  filesystem::path p_;
};

class directory_iterator {
 public:
  directory_iterator() {}
  explicit directory_iterator(const filesystem::path &dir) : it_(dir.std_path()) { load(); }
  explicit directory_iterator(const std::string &dir) : it_(dir) { load(); }
  directory_iterator &operator++() {
    ++it_;
    load();
    return *this;
  }
  bool operator!=(const directory_iterator &o) const { return it_ != o.it_; }
  bool operator==(const directory_iterator &o) const { return it_ == o.it_; }
  const directory_entry *operator->() const { return &cur_; }
  const directory_entry &operator*() const { return cur_; }

 private:
  void load() {
    if (it_ != std::filesystem::directory_iterator()) cur_ = directory_entry(it_->path());
  }
  std::filesystem::directory_iterator it_;
  directory_entry cur_;
};

inline std::string extension(const path &p) { return p.std_path().extension().string(); }
inline std::string basename(const path &p) { return p.std_path().stem().string(); }
inline bool exists(const path &p) { return std::filesystem::exists(p.std_path()); }
inline bool create_directory(const path &p) {
  std::error_code ec;
  return std::filesystem::create_directory(p.std_path(), ec);
}

}  // namespace filesystem
}  // namespace boost
