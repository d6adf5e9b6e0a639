// Minimal stand-in for the part of boost::program_options that the `integrate` / `tsdf2mesh` programs
// use (no Boost on this machine, no network): long options only (`--name`, `--name value`,
// `--name=value`), typed values parsed on access, `required()`.  [Boost-recall: behaviour of
// parse_command_line / store / notify for this subset]
#pragma once

#include <map>
#include <memory>
#include <ostream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace boost {
namespace program_options {

class error : public std::logic_error {
 public:
  explicit error(const std::string &w) : std::logic_error(w) {}
};
class required_option : public error {
 public:
  explicit required_option(const std::string &n) : error("the option '--" + n + "' is required but missing") {}
};
class unknown_option : public error {
 public:
  explicit unknown_option(const std::string &n) : error("unrecognised option '" + n + "'") {}
};
class invalid_option_value : public error {
 public:
  explicit invalid_option_value(const std::string &v) : error("the argument ('" + v + "') is invalid") {}
};

class value_semantic {
 public:
  bool is_required = false;
  virtual ~value_semantic() {}
};
template <typename T>
class typed_value : public value_semantic {
 public:
  typed_value *required() {
    is_required = true;
    return this;
  }
};
template <typename T>
typed_value<T> *value() {
  return new typed_value<T>();
}

class variable_value {
 public:
  variable_value() {}
  explicit variable_value(const std::string &t) : text_(t) {}
  template <typename T>
  T as() const {
    return convert((T *)nullptr);
  }

 private:
  std::string convert(std::string *) const { return text_; }
  template <typename T>
  T convert(T *) const {
    std::istringstream is(text_);
    T v;
    is >> v;
    if (is.fail() || !(is >> std::ws).eof()) throw invalid_option_value(text_);
    return v;
  }
  std::string text_;
};

struct option_description {
  std::string long_name, description;
  std::shared_ptr<const value_semantic> semantic;  // null: a switch
};

class options_description;
class options_description_easy_init {
 public:
  explicit options_description_easy_init(options_description *o) : owner_(o) {}
  options_description_easy_init &operator()(const char *name, const char *description);
  options_description_easy_init &operator()(const char *name, const value_semantic *s, const char *description);

 private:
  options_description *owner_;
};

class options_description {
 public:
  explicit options_description(const std::string &caption = "") : caption_(caption) {}
  options_description_easy_init add_options() { return options_description_easy_init(this); }
  const option_description *find(const std::string &long_name) const {
    for (const auto &o : options_)
      if (o.long_name == long_name) return &o;
    return nullptr;
  }
  std::string caption_;
  std::vector<option_description> options_;
};

inline std::string long_part(const char *name) {
  const std::string n(name);
  return n.substr(0, n.find(','));
}
inline options_description_easy_init &options_description_easy_init::operator()(const char *name, const char *description) {
  owner_->options_.push_back({long_part(name), description, nullptr});
  return *this;
}
inline options_description_easy_init &options_description_easy_init::operator()(const char *name, const value_semantic *s,
                                                                                const char *description) {
  owner_->options_.push_back({long_part(name), description, std::shared_ptr<const value_semantic>(s)});
  return *this;
}
inline std::ostream &operator<<(std::ostream &os, const options_description &d) {
  os << d.caption_ << ":\n";
  for (const auto &o : d.options_) os << "  --" << o.long_name << (o.semantic ? " arg" : "") << "\t" << o.description << "\n";
  return os;
}

class positional_options_description {};

namespace command_line_style {
enum style_t { allow_long = 1, allow_short = 2, allow_dash_for_short = 4, long_allow_adjacent = 0x40, long_allow_next = 0x80,
               unix_style = 0x3ff };
}

struct parsed_options {
  const options_description *description;
  std::vector<std::pair<std::string, std::string> > items;
};

inline parsed_options parse_command_line(int argc, const char *const *argv, const options_description &desc, int style = 0) {
  (void)style;
  parsed_options out;
  out.description = &desc;
  for (int i = 1; i < argc; ++i) {
    const std::string tok(argv[i]);
    if (tok.size() < 3 || tok[0] != '-' || tok[1] != '-') throw unknown_option(tok);
    std::string name = tok.substr(2), val;
    bool has_val = false;
    const size_t eq = name.find('=');
    if (eq != std::string::npos) {
      val = name.substr(eq + 1);
      name = name.substr(0, eq);
      has_val = true;
    }
    const option_description *o = desc.find(name);
    if (!o) throw unknown_option(tok);
    if (o->semantic) {
      if (!has_val) {
        if (i + 1 >= argc) throw error("the required argument for option '--" + name + "' is missing");
        val = argv[++i];
      }
    } else if (has_val) {
      throw error("option '--" + name + "' does not take any arguments");
    }
    out.items.push_back({name, val});
  }
  return out;
}

class variables_map : public std::map<std::string, variable_value> {
 public:
  size_t count(const std::string &n) const { return std::map<std::string, variable_value>::count(n); }
  const variable_value &operator[](const std::string &n) const {
    static const variable_value empty;
    const auto it = find(n);
    return it == end() ? empty : it->second;
  }
  const options_description *description = nullptr;
};

inline void store(const parsed_options &p, variables_map &vm) {
  vm.description = p.description;
  for (const auto &kv : p.items)
    if (!vm.count(kv.first)) vm.insert({kv.first, variable_value(kv.second)});
}

inline void notify(variables_map &vm) {
  if (!vm.description) return;
  for (const auto &o : vm.description->options_)
    if (o.semantic && o.semantic->is_required && !vm.count(o.long_name)) throw required_option(o.long_name);
}

}  // namespace program_options
}  // namespace boost
