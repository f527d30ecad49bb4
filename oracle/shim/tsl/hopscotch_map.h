#pragma once
// oracle build glue (test infrastructure): API subset of tsl::hopscotch_map on
// top of std::unordered_map, standing in for the un-vendored vaexio/hopscotch-map
// fork. Results are identical (ordinals come from insertion order); timings of the
// hash path are NOT faithful to the reference.
#include <memory>
#include <unordered_map>
namespace tsl {
template <class K, class V, class H = std::hash<K>, class E = std::equal_to<K>, class A = std::allocator<std::pair<K, V>>, unsigned N = 62, bool S = false>
class hopscotch_map : public std::unordered_map<K, V, H, E> {
  public:
    using base = std::unordered_map<K, V, H, E>;
    using base::base;
    struct iterator : base::iterator {
        iterator() {}
        iterator(typename base::iterator it) : base::iterator(it) {}
        V &value() { return (*this)->second; }
    };
    iterator find(const K &k) { return iterator(base::find(k)); }
    template <class K2> iterator find(const K2 &k) { return iterator(base::find(K(k))); }
    iterator end() { return iterator(base::end()); }
    iterator begin() { return iterator(base::begin()); }
    typename base::const_iterator find(const K &k) const { return base::find(k); }
    typename base::const_iterator end() const { return base::end(); }
    typename base::const_iterator begin() const { return base::begin(); }
};
template <class K, class V, class H = std::hash<K>, class E = std::equal_to<K>, class A = std::allocator<std::pair<K, V>>, unsigned N = 62, bool S = false>
using hopscotch_pg_map = hopscotch_map<K, V, H, E, A, N, S>;
} // namespace tsl
