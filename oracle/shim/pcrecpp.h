#pragma once
// oracle build glue: the tiny pcrecpp surface strings.cpp uses, on std::regex
// (only needed so that `import vaex` works in the golden-vector generator).
#include <regex>
#include <string>
namespace pcrecpp {
class RE_Options {
  public:
    bool caseless_ = false;
    void set_caseless(bool v) { caseless_ = v; }
};
class RE {
  public:
    RE(const std::string &p) : re_(p) {}
    RE(const std::string &p, const RE_Options &o) : re_(p, o.caseless_ ? std::regex::ECMAScript | std::regex::icase : std::regex::ECMAScript) {}
    template <class S> bool PartialMatch(const S &s) const { std::string t(s); return std::regex_search(t, re_); }
    template <class S> bool FullMatch(const S &s) const { std::string t(s); return std::regex_match(t, re_); }
    template <class S> int GlobalReplace(const S &rewrite, std::string *str) const { *str = std::regex_replace(*str, re_, std::string(rewrite)); return 1; }
    template <class S> bool Replace(const S &rewrite, std::string *str) const { *str = std::regex_replace(*str, re_, std::string(rewrite), std::regex_constants::format_first_only); return true; }
    std::regex re_;
};
} // namespace pcrecpp
