// TEST INFRASTRUCTURE ONLY -- a stand-in for <boost/geometry.hpp>, which is not installed in this environment, so that the
// reference's det3d/ops/nms/nms_cpu.h can be compiled FROM ITS OWN SOURCE (oracle/build.py -> oracle/_ref/ref_nms_cpu*.so) and
// its control flow (greedy rotated NMS :72-168, DI-NMS :173-384) run as the checker of oracle/rotate_nms.c and oracle/di_nms.c.
// It provides exactly what that header uses -- model::point / model::polygon, append, intersection, union_, area, clear -- for the
// only case the header creates: two closed, convex, clockwise quadrilaterals. Geometry: Sutherland-Hodgman clipping and the
// shoelace formula in double precision; union_ returns a polygon that only carries its area |A| + |B| - |A n B|.
// This is NOT boost: degenerate contacts (touching edges, zero-area overlaps) may be classified differently, and boost works in
// the coordinate type while this works in double. What it pins is the reference's CONTROL FLOW around the two area calls.
#pragma once
#include <cmath>
#include <vector>

namespace boost {
namespace geometry {
namespace cs {
struct cartesian {};
}  // namespace cs
namespace model {
template <typename T, int D, typename CS>
struct point {
  T c[2];
  point() : c{0, 0} {}
  point(T x, T y) : c{x, y} {}
};
template <typename P>
struct polygon {
  std::vector<P> ring;
  bool has_area = false;
  double carried_area = 0.0;
  void clear() {
    ring.clear();
    has_area = false;
  }
};
}  // namespace model

namespace shim_detail {
struct V {
  double x, y;
};
inline double shoelace(const std::vector<V>& p) {
  double s = 0.0;
  const size_t n = p.size();
  for (size_t i = 0; i < n; ++i) {
    const V &a = p[i], &b = p[(i + 1) % n];
    s += a.x * b.y - b.x * a.y;
  }
  return 0.5 * std::fabs(s);
}
template <typename Poly>
inline std::vector<V> open_ring(const Poly& p) {
  std::vector<V> r;
  for (const auto& q : p.ring) r.push_back(V{(double)q.c[0], (double)q.c[1]});
  if (r.size() > 1 && r.front().x == r.back().x && r.front().y == r.back().y) r.pop_back();
  return r;
}
// clip `subj` by the half planes of the convex polygon `clip` (either orientation)
inline std::vector<V> clip_convex(std::vector<V> subj, const std::vector<V>& clip) {
  double orient = 0.0;
  for (size_t i = 0; i < clip.size(); ++i) {
    const V &a = clip[i], &b = clip[(i + 1) % clip.size()];
    orient += a.x * b.y - b.x * a.y;
  }
  const double sgn = orient >= 0.0 ? 1.0 : -1.0;
  for (size_t i = 0; i < clip.size() && !subj.empty(); ++i) {
    const V a = clip[i], b = clip[(i + 1) % clip.size()];
    std::vector<V> out;
    auto side = [&](const V& p) { return sgn * ((b.x - a.x) * (p.y - a.y) - (b.y - a.y) * (p.x - a.x)); };
    for (size_t k = 0; k < subj.size(); ++k) {
      const V p = subj[k], q = subj[(k + 1) % subj.size()];
      const double sp = side(p), sq = side(q);
      if (sp >= 0.0) out.push_back(p);
      if ((sp >= 0.0) != (sq >= 0.0)) {
        const double t = sp / (sp - sq);
        out.push_back(V{p.x + t * (q.x - p.x), p.y + t * (q.y - p.y)});
      }
    }
    subj.swap(out);
  }
  return subj;
}
}  // namespace shim_detail

template <typename Poly, typename P>
inline void append(Poly& poly, const P& pt) {
  poly.ring.push_back(pt);
}

template <typename Poly>
inline double area(const Poly& poly) {
  if (poly.has_area) return poly.carried_area;
  return shim_detail::shoelace(shim_detail::open_ring(poly));
}

template <typename Poly>
inline bool intersection(const Poly& a, const Poly& b, std::vector<Poly>& out) {
  const auto r = shim_detail::clip_convex(shim_detail::open_ring(a), shim_detail::open_ring(b));
  if (r.size() >= 3 && shim_detail::shoelace(r) > 0.0) {
    Poly p;
    for (const auto& v : r) p.ring.push_back(typename std::decay<decltype(a.ring[0])>::type((decltype(a.ring[0].c[0]))v.x, (decltype(a.ring[0].c[0]))v.y));
    p.has_area = true;  // keep the double-precision area of the clipped polygon
    p.carried_area = shim_detail::shoelace(r);
    out.push_back(p);
  }
  return true;
}

template <typename Poly>
inline bool union_(const Poly& a, const Poly& b, std::vector<Poly>& out) {
  const auto ra = shim_detail::open_ring(a), rb = shim_detail::open_ring(b);
  const auto ri = shim_detail::clip_convex(ra, rb);
  const double ai = ri.size() >= 3 ? shim_detail::shoelace(ri) : 0.0;
  Poly p;
  p.has_area = true;
  p.carried_area = shim_detail::shoelace(ra) + shim_detail::shoelace(rb) - ai;
  if (p.carried_area > 0.0) out.push_back(p);
  return true;
}

}  // namespace geometry
}  // namespace boost
