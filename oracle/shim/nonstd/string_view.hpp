#pragma once
// oracle build glue (test infrastructure): stands in for the un-vendored
// martinmoene/string-view-lite submodule the reference includes.
#include <string_view>
namespace nonstd { using std::string_view; }
